#!/bin/bash
# px per Gaussian = 4 (tl_dense boundary): Gaussian- or tile-stationary backward (from lists)?  and the policy sweep on the final rules
mkdir -p gpurun_out/r05aa
E1="GSASR_SPLAT_DEV=1 GSASR_SPLAT_BWD=tile GSASR_SPLAT_LISTS=1"
for sh in "x2_1024 512 512 2 0.1 0 10 1 6" "x2_2048 1024 1024 2 0.1 0 5 1 6" "x2_512 256 256 2 0.1 0 10 1 6" "d4x4_1024 256 256 4 0.1 0 10 4 6" "d4x4_512 128 128 4 0.1 0 10 4 6" "d4x8_2048 256 256 8 0.1 0 5 4 6" "d16x8_1024 128 128 8 0.1 0 10 16 6" "x3_1536 512 512 3 0.1 0 10 1 6" "d16x2_512 256 256 2 0.1 0 10 16 6"; do
  set -- $sh; name=$1; shift
  for dist in 0 1 2 3; do
    echo -n "$name dist$dist default: "; MB_DIST=$dist tools/bin/mb "$@" | tail -1 | sed -E 's/N=.*\| plan/plan/; s/\| sum.*reach/reach/; s/ maxcell.*//'
    echo -n "$name dist$dist tile+lists: "; env $E1 MB_DIST=$dist tools/bin/mb "$@" | tail -1 | sed -E 's/N=.*\| plan/plan/; s/\| sum.*reach/reach/; s/ maxcell.*//'
  done
done | tee gpurun_out/r05aa/pxg4.txt
bash tools/policy_sweep.sh > gpurun_out/r05aa/policy_sweep_after.txt 2>&1
python tools/policy_regret.py gpurun_out/r05aa/policy_sweep_after.txt | tee gpurun_out/r05aa/policy_regret_after.txt
