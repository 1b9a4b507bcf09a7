set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
timeout 600 python tools/fuzz_lists.py 40 11 > $O/fuzz_lists.txt 2>&1; tail -2 $O/fuzz_lists.txt
timeout 600 python tools/fuzz_cross.py 40 3 > $O/fuzz_cross.txt 2>&1; tail -2 $O/fuzz_cross.txt
B="python bench.py --no-cpu-baseline --no-extras --no-live-pmc --no-graph"
for cfg in c2 c2x16 c5; do
  $B --config $cfg > $O/bench_${cfg}_pair_lists.json 2>> $O/bench.err
  GSASR_SPLAT_DEV=1 GSASR_SPLAT_LISTS=0 $B --config $cfg > $O/bench_${cfg}_pair_search.json 2>> $O/bench.err
  GSASR_SPLAT_LIB=$GRAFT_REPO_ROOT/gsasr_amd/lib/libgsasr_splat_nopair.so $B --config $cfg > $O/bench_${cfg}_nopair_lists.json 2>> $O/bench.err
done
for c in 0 -1; do
  python tools/published_run.py --cutoff $c >> $O/published.txt 2>&1
  GSASR_SPLAT_LIB=$GRAFT_REPO_ROOT/gsasr_amd/lib/libgsasr_splat_nopair.so python tools/published_run.py --cutoff $c >> $O/published_nopair.txt 2>&1
done
cat $O/published.txt $O/published_nopair.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05c/bench_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), round(d['ms_per_step']*1e3,1), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
cat $O/pytest.txt
