#!/bin/bash
mkdir -p gpurun_out/r05s
python -m pytest tests/test_tune.py tests/test_host_path.py tests/test_bwd_tile.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05s/pytest.txt
cat gpurun_out/r05s/pytest.txt
python - <<'P' 2>&1 | tee gpurun_out/r05s/tune_batch_demo.txt
import sys, torch
sys.path.insert(0, '.')
from gsasr_amd import synthetic, tune
sys.path.insert(0, 'tools')
from tune_demo import reshape, DISTS
dev = torch.device('cuda:0')
B, lr, scale = 16, 48, 4.0
sizes = [(192, 192)] * B
print("config-5 canvas (16 x 192^2, 16 Gaussians per LR pixel, dmax 0.5): gsasr_amd.tune.tune_batch(register=False); sigma scaled through the raw parameters")
import math
for di, dn in enumerate(DISTS):
    p = torch.stack([synthetic.gs_parameters(lr, lr, seed=b, gpp=16) for b in range(B)]).to(dev)
    g = torch.Generator().manual_seed(100 + di)
    r = torch.rand(B, p.shape[1], 3, generator=g).to(dev)
    def logit(x): return torch.log(x / (1 - x))
    sx, sy = torch.sigmoid(p[..., 0]), torch.sigmoid(p[..., 1])
    if di == 1: sx, sy = 0.1 + 0.2 * r[..., 0], 0.1 + 0.2 * r[..., 1]
    elif di == 2: sx, sy = 0.85 + 0.149 * r[..., 0], 0.85 + 0.149 * r[..., 1]
    elif di == 3: sx, sy = 0.05 * torch.pow(torch.tensor(20.0, device=dev), r[..., 0]), 0.05 * torch.pow(torch.tensor(20.0, device=dev), r[..., 1])
    elif di == 4:
        f = r[..., 2] < 0.5
        sx, sy = torch.where(f, 0.9, 0.1), torch.where(f, 0.1, 0.9)
    elif di == 5:
        sx = sy = torch.where(r[..., 2] < 0.9, 0.15, 0.95)
    p[..., 0], p[..., 1] = logit(sx.clamp(1e-4, 1 - 1e-4)), logit(sy.clamp(1e-4, 1 - 1e-4))
    steps = torch.full((B,), 1.2 / scale, device=dev)
    res = tune.tune_batch(p.contiguous(), steps, sizes, 0.5, register=False)
    d = res.ms['default']
    print(f"{dn:18s} default {d:.4f} ms  picks {res.name:16s} {res.ms[res.name]:.4f} ms  gain {100*(1-res.ms[res.name]/d):5.1f}%   " + " ".join(f"{k}={v:.3f}" for k, v in res.ms.items()))
    tune.reset()
P
