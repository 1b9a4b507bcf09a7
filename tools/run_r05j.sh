cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j; mkdir -p $O
F=$O/fuzz_final_build.txt
{
echo "# Fuzz of the round-5 final build (HEAD of this commit), one MI355X; each tool prints its own worst errors"
for cmd in "fuzz_cross.py 600 101" "fuzz_lists.py 300 102" "fuzz_step.py 200 103" "fuzz_host.py 200 104" "fuzz_batch.py 200" "fuzz_sample.py 200" "fuzz_bands.py 150 105" "fuzz_wide.py 200"; do
  echo "== python tools/$cmd"; /usr/bin/time -f "   (%e s)" timeout 1500 python tools/$cmd 2>&1 | grep -v amdgpu.ids | tail -2
done
echo "== the same cross fuzz with every development kernel forced (lists everywhere; eight Gaussians per backward wave)"
for env in "GSASR_SPLAT_LISTS=1" "GSASR_SPLAT_BWD8=1" "GSASR_SPLAT_LISTS=1 GSASR_SPLAT_BWD=tile"; do
  echo "== GSASR_SPLAT_DEV=1 $env python tools/fuzz_cross.py 300 106"; env GSASR_SPLAT_DEV=1 $env timeout 1500 python tools/fuzz_cross.py 300 106 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== GSASR_SPLAT_DEV=1 $env python tools/fuzz_step.py 100 107"; env GSASR_SPLAT_DEV=1 $env timeout 1500 python tools/fuzz_step.py 100 107 2>&1 | grep -v amdgpu.ids | tail -1
done
} > $F 2>&1
cat $F
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; cat $O/bench_time.txt
