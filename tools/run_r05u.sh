#!/bin/bash
mkdir -p gpurun_out/r05u
python -m pytest tests/test_bwd_tile.py tests/test_fwd_lists.py tests/test_tune.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r05u/pytest.txt
E1="GSASR_SPLAT_DEV=1 GSASR_SPLAT_BWD=tile GSASR_SPLAT_LISTS=1"
for sh in "c2x16 256 256 4 0.1 0 10 16 6" "c5 192 192 4 0.5 0 10 16 6" "x2 512 512 2 0.1 0 10 1 6"; do set -- $sh; name=$1; shift; for dist in 0 3; do
  echo -n "$name dist$dist tile+lists: "; env $E1 MB_DIST=$dist tools/bin/mb "$@" | tail -1 | sed -E 's/N=.*\| plan/plan/; s/\| sum.*//'
  echo -n "$name dist$dist default: "; MB_DIST=$dist tools/bin/mb "$@" | tail -1 | sed -E 's/N=.*\| plan/plan/; s/\| sum.*//'
done; done | tee gpurun_out/r05u/tile4.txt
