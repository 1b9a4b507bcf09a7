#!/bin/bash
mkdir -p gpurun_out/r05v
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_c_abi.py tests/test_host_api.py tests/test_host_path.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r05v/pytest.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/tl_graph -o kt -- python $R/tools/graph_replay.py > /dev/null 2>&1
cd $R
(echo "== hipGraph replay of the step (k_zero16 in place of the runtime's fill)"; python $R/tools/timeline.py /tmp/tl_graph/kt_results.db) > gpurun_out/r05v/graph_timeline.txt 2>&1
cat gpurun_out/r05v/graph_timeline.txt
python bench.py --no-cpu-baseline --no-live-pmc --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['value'], b['ms_per_step'], b['config'].get('launch'))"
