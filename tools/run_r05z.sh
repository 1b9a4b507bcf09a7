#!/bin/bash
# Gaussian-stationary backward: plain (one trip ahead, 7 waves) or unrolled (two trips per iteration, 6 waves) sweep, by scale and size distribution
mkdir -p gpurun_out/r05z
for sh in "x4_1024 256 256 4 0.1 0 20 1 6" "x5_1280 256 256 5 0.1 0 20 1 6" "x6_1536 256 256 6 0.1 0 10 1 6" "x8_2048 256 256 8 0.1 0 10 1 6" "x12_1536 128 128 12 0.1 0 10 1 6" "x16_2048 128 128 16 0.1 0 10 1 6" "x4d16_1024 256 256 4 0.1 0 10 16 6" "x3_768 256 256 3 0.1 0 20 1 6"; do
  set -- $sh; name=$1; shift
  for dist in 0 1 2; do for u in 0 1; do echo -n "$name dist$dist unroll$u: "; MB_DIST=$dist GSASR_SPLAT_DEV=1 GSASR_SPLAT_BWD=gaussian GSASR_SPLAT_BWD_UNROLL=$u tools/bin/mb "$@" | tail -1 | sed -E 's/N=.*\| plan/plan/; s/\| sum.*//'; done; done
done | tee gpurun_out/r05z/bwd_unroll.txt
