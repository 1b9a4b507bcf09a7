#!/bin/bash
# development aid: sampled-pixel tests + per-kernel times (run through gpurun from the repo root)
cd /tmp && export TMPDIR=/tmp
timeout 300 python -m pytest $GRAFT_REPO_ROOT/tests/test_sampled_pixels.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
for B in 16 1; do
SAMPLE_TIME_HOST=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt$B -o kt -- python $GRAFT_REPO_ROOT/tools/sample_time.py $B > /tmp/o.txt 2>&1; grep "C entry\|^B=" /tmp/o.txt
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/kt$B/kt_results.db | grep "k_sample\|k_pts\|k_render\|k_bin\|k_classify\|k_prologue"
done
