#!/bin/bash
# round 5, run 17: registry + tune tests, list/default-policy tests, tune demo, quick bench legs
mkdir -p gpurun_out/r05q
python -m pytest tests/test_tune.py tests/test_fwd_lists.py tests/test_host_api.py tests/test_fwd_wide.py tests/test_bwd_tile.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05q/pytest.txt
cat gpurun_out/r05q/pytest.txt
python tools/tune_demo.py > gpurun_out/r05q/tune_demo.txt 2>&1; cat gpurun_out/r05q/tune_demo.txt

