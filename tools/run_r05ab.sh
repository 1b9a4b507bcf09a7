#!/bin/bash
mkdir -p gpurun_out/r05ab
python -m pytest tests/test_bwd_tile.py tests/test_fwd_lists.py tests/test_tune.py tests/test_hip_parity.py tests/test_rows_vs_oracle.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r05ab/pytest.txt
for sh in "x2_1024 512 512 2 0.1 0 10 1 6" "x2_2048 1024 1024 2 0.1 0 5 1 6" "x2_512 256 256 2 0.1 0 10 1 6" "d4x4_1024 256 256 4 0.1 0 10 4 6" "d16x8_1024 128 128 8 0.1 0 10 16 6" "x3_1536 512 512 3 0.1 0 10 1 6" "c2 256 256 4 0.1 0 20 1 6" "c2x16 256 256 4 0.1 0 10 16 6"; do
  set -- $sh; name=$1; shift
  for dist in 0 3; do echo -n "$name dist$dist default: "; MB_DIST=$dist tools/bin/mb "$@" | tail -1 | sed -E 's/N=.*\| plan/plan/; s/\| sum.*reach/reach/; s/ maxcell.*//'; done
done | tee gpurun_out/r05ab/pxg4_after.txt
timeout 900 python tools/fuzz_cross.py 150 301 2>&1 | grep -v amdgpu | tail -1
timeout 900 python tools/fuzz_choices.py 60 302 2>&1 | grep -v amdgpu | tail -1
