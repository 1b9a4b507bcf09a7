#!/bin/bash
# two-level / list forward: one or two waves per sub-tile, by image size and density
mkdir -p gpurun_out/r05y
for sh in "x4_256 64 64 4 0.1 0 20 1 6" "x4_512 128 128 4 0.1 0 20 1 6" "x4_640 160 160 4 0.1 0 20 1 6" "x4_768 192 192 4 0.1 0 20 1 6" "x4_896 224 224 4 0.1 0 20 1 6" "x4_1024 256 256 4 0.1 0 20 1 6" "x4_1280 320 320 4 0.1 0 20 1 6" "x4_1536 384 384 4 0.1 0 20 1 6" \
          "d16_512 128 128 4 0.1 0 20 16 6" "d16_768 192 192 4 0.1 0 20 16 6" "d16_896 224 224 4 0.1 0 10 16 6" "d16_1024 256 256 4 0.1 0 10 16 6" "d16_1280 320 320 4 0.1 0 10 16 6" "c5 192 192 4 0.5 0 10 16 6"; do
  set -- $sh; name=$1; shift
  for p in 1 2; do echo -n "$name parts$p: "; GSASR_SPLAT_DEV=1 GSASR_SPLAT_FWD_PARTS=$p tools/bin/mb "$@" | tail -1 | sed -E 's/N=.*\| plan/plan/; s/\| sum.*//'; done
done | tee gpurun_out/r05y/fwd_parts.txt
