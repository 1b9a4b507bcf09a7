#!/bin/bash
# How good are the library's kernel choices when the Gaussians are NOT "about one LR pixel"?  For seven size / placement distributions
# (MB_DIST, tools/mb.hip) and four shapes, time the default choice and every forced combination of the development switches
# (wide forward x backward kernel (Gaussian-stationary, tile-stationary, home-tile) x tile lists); tools/policy_regret.py turns the output into default-vs-best tables.
#   bash tools/policy_sweep.sh > gpurun_out/policy_sweep.txt       (tools/bin/mb built by tools/build_mb.sh)
cd "$(dirname "$0")/.."
B=tools/bin/mb
shapes=("x4 256 256 4 0.1 0 10 1 6" "x4d16 128 128 4 0.1 0 10 16 6" "x8 384 384 8 0.1 0 5 1 6" "x12 256 256 12 0.1 0 5 1 6" "x2 512 512 2 0.1 0 10 1 6" "c5 192 192 4 0.5 0 10 16 6" "c2x16 256 256 4 0.1 0 6 16 6")
for dist in 0 1 2 3 4 5 6; do
  for sh in "${shapes[@]}"; do
    set -- $sh; name=$1; shift
    echo -n "dist$dist $name default : "; MB_DIST=$dist timeout 120 $B "$@" | tail -1
    for w in 0 1; do for b in gaussian tile home; do for l in 0 1; do
      echo -n "dist$dist $name wide$w-$b-lists$l : "
      MB_DIST=$dist GSASR_SPLAT_DEV=1 GSASR_SPLAT_FWD_WIDE=$w GSASR_SPLAT_BWD=$b GSASR_SPLAT_LISTS=$l timeout 120 $B "$@" | tail -1
    done; done; done
  done
done
