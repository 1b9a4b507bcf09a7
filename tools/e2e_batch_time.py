#!/usr/bin/env python
"""Config-5-shaped rasterizer work end to end through the HOST API: a batch of B samples (48x48 LR crops x4,
16 Gaussians per LR pixel, dmax 0.5) forward + backward to the raw decoder parameters -- the reference's
per-sample loop (basicsr/models/gsasr_model.py:191-233) vs generate_2D_gaussian_splatting_batch. Development aid.

    python tools/e2e_batch_time.py [B lr scale gpp dmax]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import gaussian_splatting as gsp, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lr = int(sys.argv[2]) if len(sys.argv) > 2 else 48
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
gpp = int(sys.argv[4]) if len(sys.argv) > 4 else 16
dmax = float(sys.argv[5]) if len(sys.argv) > 5 else 0.5
dev = torch.device("cuda:0")
p = torch.stack([synthetic.gs_parameters(lr, lr, seed=b, gpp=gpp) for b in range(B)]).to(dev)
H = W = int(lr * scale)
sizes = [(H, W)] * B
sm = [torch.tensor([scale, scale], device=dev)] * B
wgt = torch.rand(B, 3, H, W, device=dev)


def loop():
    pa = p.detach().requires_grad_(True)
    loss = 0
    for b in range(B):
        out = gsp.generate_2D_gaussian_splatting_step((H, W), pa[b], scale, sm[b], dmax=dmax)
        loss = loss + (out * wgt[b]).sum()
    loss.backward()


def batched():
    pa = p.detach().requires_grad_(True)
    out = gsp.generate_2D_gaussian_splatting_batch(sizes, pa, [scale] * B, sm, dmax=dmax)
    (out * wgt).sum().backward()


for name, fn in (("per-sample loop", loop), ("batched", batched)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:16s} B={B} N={p.shape[1]} {H}x{W}: {dt * 1e3:7.3f} ms per fwd+bwd  ({B * H * W / dt / 1e6:.0f} HR Mpx/s)")
