set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_exp.hip -o /tmp/mfma_exp && /tmp/mfma_exp > $O/mfma_exp.txt 2>&1; cat $O/mfma_exp.txt
timeout 600 python tools/fuzz_lists.py 40 13 > $O/fuzz_lists.txt 2>&1; tail -2 $O/fuzz_lists.txt
B="python bench.py --no-cpu-baseline --no-extras --no-live-pmc --no-graph"
for cfg in c2 c3 c4 c2x16 c5; do
  $B --config $cfg > $O/bench_${cfg}_lists.json 2>> $O/bench.err
  GSASR_SPLAT_DEV=1 GSASR_SPLAT_LISTS=0 $B --config $cfg > $O/bench_${cfg}_search.json 2>> $O/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05d/bench_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), round(d['ms_per_step']*1e3,1), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
