#!/bin/bash
# small images: k_render_fwd_split against the two-level kernel with two waves per sub-tile
mkdir -p gpurun_out/r05w
for sh in "c1_256 64 64 4 0.1 0 20 1 6" "x4_384 96 96 4 0.1 0 20 1 6" "x4_512 128 128 4 0.1 0 20 1 6" "x4_640 160 160 4 0.1 0 20 1 6" "crop192_d16 48 48 4 0.5 0 20 16 6" "d16_256 64 64 4 0.1 0 20 16 6" "d16_384 96 96 4 0.1 0 20 16 6" "x8_512 64 64 8 0.1 0 20 1 6" "x2_512 256 256 2 0.1 0 20 1 6"; do
  set -- $sh; name=$1; shift
  echo -n "$name split: "; tools/bin/mb "$@" | tail -1 | sed -E 's/N=.*\| plan/plan/; s/\| sum.*//'
  echo -n "$name two-level: "; GSASR_SPLAT_DEV=1 GSASR_SPLAT_FWD_SPLIT=0 tools/bin/mb "$@" | tail -1 | sed -E 's/N=.*\| plan/plan/; s/\| sum.*//'
done | tee gpurun_out/r05w/split_vs_twolevel.txt
