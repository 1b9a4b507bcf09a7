set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r05a/pytest.txt
timeout 600 python bench.py > gpurun_out/r05a/bench.json 2> gpurun_out/r05a/bench.err
cd /tmp && export TMPDIR=/tmp
for c in 0 -1; do
  rocprofv3 --kernel-trace --stats -d /tmp/pub$c -o kt -- python $GRAFT_REPO_ROOT/tools/published_run.py --cutoff $c > $GRAFT_REPO_ROOT/gpurun_out/r05a/published_run_$c.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/pub$c/kt_results.db --skip 2 > $GRAFT_REPO_ROOT/gpurun_out/r05a/kernel_stats_published_$c.txt 2>&1
done
