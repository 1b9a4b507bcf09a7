#!/bin/bash
# What the render kernels' time is made of: what-if builds of the library source (guarded macros in gsasr_amd/csrc/splat_forward.hip /
# splat_backward.hip; results of these builds are WRONG by construction) timed by the micro-benchmark on the BASELINE-shaped workloads.
#   BUILD_ONLY=1 bash tools/whatif.sh      (anywhere hipcc is)  then   bash tools/whatif.sh > gpurun_out/whatif.txt   (on the GPU box)
#     FWD_EXP_NOEVAL   records found, fetched and staged, one per chunk evaluated      FWD_EXP_NOSTORE  no image store
#     BWD_EXP_NOLOAD   synthetic gradient values instead of the loads of a trip        BWD_EXP_NOMATH   loads consumed by adds, no trip arithmetic
#     BWD_EXP_PAIRPLANAR the cost of a row-pair planar gradient (one 16-byte + one 8-byte load, 3 + 3 packed FMAs instead of 6 + 5 scalar ones)
#     BWD_EXP_NOSWEEP  no sweep: record fetch + wave reduction + write remain             BWD_EXP_NOREDUCE no wave reduction
#     FWD_PAIR=0 | 1   the pixel-packed forward evaluation of rounds 1-4 | the record-pair packed one, everywhere (default: by density)
cd "$(dirname "$0")/.."
for v in "base" "noeval -DFWD_EXP_NOEVAL" "nostore -DFWD_EXP_NOSTORE" "noeval_nostore -DFWD_EXP_NOEVAL -DFWD_EXP_NOSTORE" "nopair -DFWD_PAIR=0" "pair -DFWD_PAIR=1" \
         "noload -DBWD_EXP_NOLOAD" "nomath -DBWD_EXP_NOMATH" "noload_nomath -DBWD_EXP_NOLOAD -DBWD_EXP_NOMATH" "pairplanar -DBWD_EXP_PAIRPLANAR" \
         "nosweep -DBWD_EXP_NOSWEEP" "noreduce -DBWD_EXP_NOREDUCE" "nosweep_noreduce -DBWD_EXP_NOSWEEP -DBWD_EXP_NOREDUCE"; do
  set -- $v; name=$1; shift
  # (built where hipcc is: tools/bin/ travels with gpurun, so binaries made in the authoring container are reused on the GPU box)
  [ -x tools/bin/mb_$name ] && [ tools/bin/mb_$name -nt gsasr_amd/csrc/splat_forward.hip ] && [ tools/bin/mb_$name -nt gsasr_amd/csrc/splat_backward.hip ] || \
    bash tools/build_mb.sh $name "$@" > /dev/null 2>&1 || { echo "$name: build failed"; continue; }
  [ -n "$BUILD_ONLY" ] || bash tools/mbset.sh tools/bin/mb_$name $name
done
