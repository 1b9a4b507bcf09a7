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


def loop(sms, loss_style):
    pa = p.detach().requires_grad_(True)
    loss = 0
    outs = []
    for b in range(B):
        out = gsp.generate_2D_gaussian_splatting_step((H, W), pa[b], scale, sms[b], dmax=dmax)
        if loss_style:
            loss = loss + (out * wgt[b]).sum()
        else:
            outs.append(out)
    if loss_style:
        loss.backward()
    else:
        torch.autograd.backward(outs, [wgt[b] for b in range(B)])


def batched(sms, loss_style):
    pa = p.detach().requires_grad_(True)
    out = gsp.generate_2D_gaussian_splatting_batch(sizes, pa, [scale] * B, sms, dmax=dmax)
    if loss_style:
        (out * wgt).sum().backward()
    else:
        out.backward(wgt)


sm_py = [(scale, scale)] * B
for name, fn in (("per-sample loop", loop), ("batched", batched)):
    for sm_name, sms in (("scale_modify on the GPU", sm), ("scale_modify as numbers", sm_py)):
        for loss_style in (True, False):
            for _ in range(4):
                fn(sms, loss_style)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                n = 20
                for _ in range(n):
                    fn(sms, loss_style)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / n)
            print(f"{name:16s} {sm_name:24s} {'torch loss' if loss_style else 'direct grad':11s} B={B} N={p.shape[1]} {H}x{W}: "
                  f"{best * 1e3:7.3f} ms per fwd+bwd  ({B * H * W / best / 1e6:.0f} HR Mpx/s)")
gsp.deferred_asserts.flush()
