#!/usr/bin/env python
"""Randomised check of the batched canvas against single-image calls (GPU, development aid):
   python tools/fuzz_batch.py [cases]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import gaussian_splatting as gsp  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
BIG = int(sys.argv[2]) if len(sys.argv) > 2 else 300      # largest sample side of the "big" cases
rng = np.random.default_rng(7)
worst_img = worst_g = 0.0
for case in range(cases):
    B = int(rng.integers(2, 12))
    n = int(rng.integers(1, 600))
    big = rng.random() < 0.3
    sizes = [(int(rng.integers(2, BIG if big else 90)), int(rng.integers(2, BIG if big else 90))) for _ in range(B)]
    g = torch.Generator().manual_seed(case)
    p = (torch.randn(B, n, 9, generator=g) * float(rng.choice([0.5, 1.5, 3.0]))).to(dev)
    p[:, :, 7:9] = torch.rand(B, n, 2, generator=g).to(dev) * 1.4 - 0.2
    scales = [float(rng.uniform(1.0, 9.0)) for _ in range(B)]
    sms = [torch.tensor([s, s], device=dev) for s in scales]
    kw = [dict(if_dmax=False), dict(if_dmax=True, dmax_mode="fix", dmax=float(10 ** rng.uniform(-2.5, 0.3)))][case % 2]
    hm, wm = max(h for h, _ in sizes), max(w for _, w in sizes)
    pa = p.clone().requires_grad_(True)
    out = gsp.generate_2D_gaussian_splatting_batch(sizes, pa, scales, sms, **kw)
    wgt = torch.rand(B, 3, hm, wm, device=dev)
    (out * wgt).sum().backward()
    for b, (h, w) in enumerate(sizes):
        pb = p[b].clone().requires_grad_(True)
        ref = gsp.generate_2D_gaussian_splatting_step((h, w), pb, scales[b], sms[b], **kw)
        (ref * wgt[b, :, :h, :w]).sum().backward()
        ei = float((out[b, :, :h, :w] - ref).detach().abs().max()) / max(1.0, float(ref.detach().abs().max()))
        pad = out[b].detach().clone()
        pad[:, :h, :w] = 0
        eg = float((pa.grad[b] - pb.grad).abs().max()) / max(1e-30, float(pb.grad.abs().max()))
        worst_img, worst_g = max(worst_img, ei), max(worst_g, eg)
        assert torch.isfinite(out[b]).all() and float(pad.abs().max()) == 0.0, (case, b, "padding / finite")
        assert ei <= 1e-5 and eg <= 2e-4, (case, b, sizes[b], n, kw, ei, eg)
print(f"{cases} cases ok: worst image rel err {worst_img:.2e}, worst gradient rel err {worst_g:.2e}")
