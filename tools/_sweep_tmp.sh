cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/persub; export GSASR_SPLAT_DEV=1
for lr in 160 176 184 192 208 224 256; do
  for ps in 0 x; do
    echo -n "lr$lr persub=$ps: "; if [ $ps = 0 ]; then export GSASR_SPLAT_FWD_PERSUB=0; else unset GSASR_SPLAT_FWD_PERSUB; fi; timeout 120 tools/bin/mb $lr $lr 4 0.5 0 12 16 6 2>&1 | grep -o "plan.*bwd [0-9.]* us.*sum(img)=[0-9.e+]*" | head -1; 
  done
done > gpurun_out/persub/rule.txt 2>&1
unset GSASR_SPLAT_FWD_PERSUB
cat gpurun_out/persub/rule.txt
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -5 > gpurun_out/persub/pytest.txt; cat gpurun_out/persub/pytest.txt
python bench.py --config c5 --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})"
