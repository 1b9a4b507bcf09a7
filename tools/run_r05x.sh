cd $GRAFT_REPO_ROOT
O=gpurun_out/r05x; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_full.txt; cat $O/pytest_full.txt
F=$O/fuzz_final_build.txt
{
echo "# Fuzz of the round-5 final build (tile lists from 512^2, sparse small images through the two-level kernel, four-wave tile backward,"
echo "# kernel-choice registry), one MI355X; each tool prints its own worst errors"
for cmd in "fuzz_choices.py 150 201" "fuzz_cross.py 500 202" "fuzz_lists.py 200 203" "fuzz_step.py 200 204" "fuzz_host.py 150 205" "fuzz_batch.py 150" "fuzz_sample.py 150" "fuzz_bands.py 100 206" "fuzz_wide.py 150"; do
  echo "== python tools/$cmd"; timeout 1500 python tools/$cmd 2>&1 | grep -v amdgpu.ids | tail -2
done
echo "== GSASR_SPLAT_DEV=1 GSASR_SPLAT_BWD=tile GSASR_SPLAT_LISTS=1 python tools/fuzz_cross.py 200 207"
GSASR_SPLAT_DEV=1 GSASR_SPLAT_BWD=tile GSASR_SPLAT_LISTS=1 timeout 1500 python tools/fuzz_cross.py 200 207 2>&1 | grep -v amdgpu.ids | tail -1
echo "== GSASR_SPLAT_DEV=1 GSASR_SPLAT_FWD_SPLIT=0 python tools/fuzz_cross.py 200 208"
GSASR_SPLAT_DEV=1 GSASR_SPLAT_FWD_SPLIT=0 timeout 1500 python tools/fuzz_cross.py 200 208 2>&1 | grep -v amdgpu.ids | tail -1
} > $F 2>&1
cat $F
