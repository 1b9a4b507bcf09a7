cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-extras --no-live-pmc --no-graph"
for cfg in c2 c2x16 c5; do
  $B --config $cfg > $O/bench_${cfg}_lx21.json 2>> $O/bench.err
  GSASR_SPLAT_LIB=$GRAFT_REPO_ROOT/gsasr_amd/lib/libgsasr_splat_nolx21.so $B --config $cfg > $O/bench_${cfg}_nolx21.json 2>> $O/bench.err
  $B --config $cfg > $O/bench_${cfg}_lx21_b.json 2>> $O/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05k/bench_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), round(d['ms_per_step']*1e3,1), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
F=$O/fuzz_final_build.txt
{
echo "# Fuzz of the round-5 final build, one MI355X; each tool prints its own worst errors"
for cmd in "fuzz_cross.py 600 101" "fuzz_lists.py 300 102" "fuzz_step.py 200 103" "fuzz_host.py 200 104" "fuzz_batch.py 200" "fuzz_sample.py 200" "fuzz_bands.py 150 105" "fuzz_wide.py 200"; do
  echo "== python tools/$cmd"; timeout 1500 python tools/$cmd 2>&1 | grep -v amdgpu.ids | tail -2
done
} > $F 2>&1
cat $F
