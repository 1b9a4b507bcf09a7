#!/bin/bash
# small dense images, forward from narrow lists: two or four waves per sub-tile
mkdir -p gpurun_out/r05p
for sh in "d16_192 48 48 4 0.1 0 20 16 6" "d16_256 64 64 4 0.1 0 20 16 6" "d16_384 96 96 4 0.1 0 20 16 6" "d16_512 128 128 4 0.1 0 20 16 6" "d16_640 160 160 4 0.1 0 20 16 6" "d16_768 192 192 4 0.1 0 20 16 6" "d4_512 128 128 4 0.1 0 20 4 6" "d4_256 64 64 4 0.1 0 20 4 6"; do
  set -- $sh; name=$1; shift
  for dist in 0 3; do
  echo -n "dist$dist $name nolists : "; MB_DIST=$dist MB_LIST_CAP=-1 tools/bin/mb_p4_0 "$@" | tail -1
  echo -n "dist$dist $name lists-2waves : "; MB_DIST=$dist MB_LIST_CAP=4096 tools/bin/mb_p4_0 "$@" | tail -1
  echo -n "dist$dist $name lists-4waves : "; MB_DIST=$dist MB_LIST_CAP=4096 tools/bin/mb_p4_4700 "$@" | tail -1
  done
done 2>&1 | sed -E 's/N=.*\| plan/plan/; s/\| sum\(img\)=([^ ]*).*reach/img \1 reach/; s/tau.*//' | tee gpurun_out/r05p/small_dense_parts.txt
