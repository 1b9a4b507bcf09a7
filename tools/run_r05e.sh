set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 600 python tools/fuzz_cross.py 40 5 > $O/fuzz_cross_default.txt 2>&1; tail -2 $O/fuzz_cross_default.txt
GSASR_SPLAT_DEV=1 GSASR_SPLAT_BWD8=1 timeout 600 python tools/fuzz_cross.py 40 6 > $O/fuzz_cross_bwd8.txt 2>&1; tail -2 $O/fuzz_cross_bwd8.txt
B="python bench.py --no-cpu-baseline --no-extras --no-live-pmc --no-graph"
for cfg in c2 c2x16 c5 c4; do
  $B --config $cfg > $O/bench_${cfg}_bwd8.json 2>> $O/bench.err
  GSASR_SPLAT_DEV=1 GSASR_SPLAT_BWD8=0 $B --config $cfg > $O/bench_${cfg}_bwd1.json 2>> $O/bench.err
done
GSASR_SPLAT_DEV=1 GSASR_SPLAT_BWD8=1 GSASR_SPLAT_BWD=gaussian $B --config c4 > $O/bench_c4_bwd8forced.json 2>> $O/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05e/bench_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), round(d['ms_per_step']*1e3,1), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
cat $O/pytest.txt
