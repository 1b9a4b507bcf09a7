#!/bin/bash
# tile-stationary backward from lists at density: rounds of 256, four waves per tile
mkdir -p gpurun_out/r05t
E1="GSASR_SPLAT_DEV=1 GSASR_SPLAT_BWD=tile GSASR_SPLAT_LISTS=1"
for v in mb_bt_w4 mb_bt_w8; do
  for sh in "c2x16 256 256 4 0.1 0 10 16 6" "c5 192 192 4 0.5 0 10 16 6" "x2 512 512 2 0.1 0 10 1 6"; do
    set -- $sh; name=$1; shift
    for dist in 0 3; do
      echo -n "$v $name dist$dist tile: "; env $E1 MB_DIST=$dist tools/bin/$v "$@" | tail -1
    done
  done
done 2>&1 | sed -E 's/N=.*\| plan/plan/; s/\| sum\(img\)=[^ ]* sum\|gs\|=([^ ]*).*reach/gs \1 reach/; s/tau.*//' | tee gpurun_out/r05t/bt_variants.txt
for sh in "c2x16 256 256 4 0.1 0 10 16 6" "c5 192 192 4 0.5 0 10 16 6"; do set -- $sh; name=$1; shift; for dist in 0 3; do echo -n "mb $name dist$dist default: "; MB_DIST=$dist tools/bin/mb "$@" | tail -1 | sed -E 's/N=.*\| plan/plan/; s/\| sum.*//'; done; done | tee -a gpurun_out/r05t/bt_variants.txt
