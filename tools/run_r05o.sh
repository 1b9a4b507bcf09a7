#!/bin/bash
# small dense images: which forward?  (split search = default; narrow from lists; wide from lists)
mkdir -p gpurun_out/r05o
B=tools/bin/mb
for sh in "d16_256 64 64 4 0.1 0 20 16 6" "d16_384 96 96 4 0.1 0 20 16 6" "d16_512 128 128 4 0.1 0 20 16 6" "d16_768 192 192 4 0.1 0 20 16 6" "d16_1024 256 256 4 0.1 0 10 16 6" "d16_x2_512 256 256 2 0.1 0 20 16 6" "d16_x3_768 256 256 3 0.1 0 10 16 6" "d4_512 128 128 4 0.1 0 20 4 6"; do
  set -- $sh; name=$1; shift
  for dist in 0 3; do
  echo -n "dist$dist $name default : "; MB_DIST=$dist $B "$@" | tail -1
  echo -n "dist$dist $name nolists : "; MB_DIST=$dist MB_LIST_CAP=-1 $B "$@" | tail -1
  echo -n "dist$dist $name narrow-lists : "; MB_DIST=$dist MB_LIST_CAP=4096 GSASR_SPLAT_DEV=1 GSASR_SPLAT_FWD_WIDE=0 $B "$@" | tail -1
  echo -n "dist$dist $name wide-lists : "; MB_DIST=$dist MB_LIST_CAP=8192 GSASR_SPLAT_DEV=1 GSASR_SPLAT_FWD_WIDE=1 $B "$@" | tail -1
  echo -n "dist$dist $name wide-nolists : "; MB_DIST=$dist MB_LIST_CAP=-1 GSASR_SPLAT_DEV=1 GSASR_SPLAT_FWD_WIDE=1 $B "$@" | tail -1
  done
done 2>&1 | sed -E 's/N=.*\| plan/plan/; s/\| sum.*reach/reach/; s/tau.*//' | tee gpurun_out/r05o/small_dense.txt
