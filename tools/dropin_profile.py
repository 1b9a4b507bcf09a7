#!/usr/bin/env python
"""Host-side cost of the reference's own calling convention at config 2: `GSCUDA.apply(sigmas, coords, colors,
torch.zeros(H,W,3), dmax)` + `.backward(grad)` (wall clock per step, then a cProfile of the enqueue path), and the
fused host API with `scale_modify` on the GPU and a torch loss.  Development aid.

    python tools/dropin_profile.py [profile]
"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import gaussian_splatting as gsp, synthetic  # noqa: E402
from gsasr_amd.gs_cuda_dmax.gswrapper import GSCUDA  # noqa: E402

dev = torch.device("cuda:0")
sig, xy, col, H, W = synthetic.kernel_inputs(256, 256, 4.0, seed=0, device="cpu")
a, b, c = (t.to(dev).requires_grad_(True) for t in (sig, xy, col))
wgt = synthetic.grad_image(H, W, 1).to(dev)
p = synthetic.gs_parameters(256, 256, seed=0).to(dev)
sm = torch.tensor([4.0, 4.0], device=dev)
wchw = wgt.permute(2, 0, 1).contiguous()


def dropin():
    a.grad = b.grad = c.grad = None
    img = GSCUDA.apply(a, b, c, torch.zeros(H, W, 3, device=dev), 0.1)
    img.backward(wgt)


def fused_loss():
    pa = p.detach().requires_grad_(True)
    out = gsp.generate_2D_gaussian_splatting_step((H, W), pa, 4.0, sm, dmax=0.1)
    (out * wchw).sum().backward()


def fused_direct():
    pa = p.detach().requires_grad_(True)
    gsp.generate_2D_gaussian_splatting_step((H, W), pa, 4.0, sm, dmax=0.1).backward(wchw)


def wall(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, fn in (("GSCUDA.apply(zeros) + backward", dropin), ("fused step, scale_modify on the GPU, torch loss", fused_loss),
                 ("fused step, scale_modify on the GPU, direct grad", fused_direct)):
    print(f"{name:52s} {wall(fn):8.1f} us per fwd+bwd")
gsp.deferred_asserts.flush()
if len(sys.argv) > 1:
    for fn in (dropin, fused_loss):
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(300):
            fn()
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)
