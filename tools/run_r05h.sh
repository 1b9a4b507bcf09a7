cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests/test_fwd_lists.py tests/test_bwd_tile.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest_lists.txt; cat $O/pytest_lists.txt
timeout 600 python tools/fuzz_cross.py 60 31 2>&1 | tail -2 > $O/fuzz_cross.txt; cat $O/fuzz_cross.txt
B="python bench.py --no-cpu-baseline --no-extras --no-live-pmc --no-graph"
for cfg in c4 c3 c2; do
  $B --config $cfg > $O/bench_${cfg}_default.json 2>> $O/bench.err
  GSASR_SPLAT_DEV=1 GSASR_SPLAT_LISTS=0 $B --config $cfg > $O/bench_${cfg}_search.json 2>> $O/bench.err
done
GSASR_SPLAT_DEV=1 GSASR_SPLAT_BWD=tile $B --config c2x16 > $O/bench_c2x16_tilebwd_lists.json 2>> $O/bench.err
GSASR_SPLAT_DEV=1 GSASR_SPLAT_BWD=tile GSASR_SPLAT_LISTS=0 $B --config c2x16 > $O/bench_c2x16_tilebwd_search.json 2>> $O/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05h/bench_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), round(d['ms_per_step']*1e3,1), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_full.txt; cat $O/pytest_full.txt
