#!/bin/bash
mkdir -p gpurun_out/r05n
bash tools/policy_sweep.sh > gpurun_out/r05n/policy_sweep.txt 2>&1
python tools/policy_regret.py gpurun_out/r05n/policy_sweep.txt | tee gpurun_out/r05n/policy_regret.txt
