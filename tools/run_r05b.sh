set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
timeout 600 python tools/fuzz_lists.py 40 7 > $O/fuzz_lists.txt 2>&1
tail -3 $O/fuzz_lists.txt
B="python bench.py --no-cpu-baseline --no-extras --no-live-pmc --no-graph"
for cfg in c2 c3 c4 c2x16 c5; do
  $B --config $cfg > $O/bench_${cfg}_lists.json 2>> $O/bench.err
  GSASR_SPLAT_DEV=1 GSASR_SPLAT_LISTS=0 $B --config $cfg > $O/bench_${cfg}_search.json 2>> $O/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05b/bench_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), round(d['ms_per_step']*1e3,1), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-live-pmc --no-graph --steps 20 --warmup 5 > /dev/null 2> $GRAFT_REPO_ROOT/$O/kt2.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/kt2/kt_results.db --skip 2 > $GRAFT_REPO_ROOT/$O/kernel_stats_c2_lists.txt
cat $GRAFT_REPO_ROOT/$O/kernel_stats_c2_lists.txt
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
cat $O/pytest.txt
