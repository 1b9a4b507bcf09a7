#!/usr/bin/env python
"""Time the HOST API end to end (decoder-style raw parameters -> [3,H,W] image -> backward to the raw
parameters): fused step vs the unfused torch prologue, with the gradient fed directly (`out.backward(g)`) and
through a torch loss (`(out * w).sum().backward()`, four more torch kernels), and with `scale_modify` as a CUDA
tensor (what the reference's training loop passes) or as Python numbers.  Development aid.

    python tools/e2e_time.py [h_lr w_lr scale gpp dmax]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import gaussian_splatting as gsp, synthetic  # noqa: E402

h_lr, w_lr = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 256
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
gpp = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dmax = float(sys.argv[5]) if len(sys.argv) > 5 else 0.1
dev = torch.device("cuda:0")
p = synthetic.gs_parameters(h_lr, w_lr, seed=0, gpp=gpp).to(dev)
H, W = int(h_lr * scale), int(w_lr * scale)
sm_gpu = torch.tensor([scale, scale], device=dev)
sm_py = (scale, scale)
wgt = synthetic.grad_image(H, W, 1, device=dev).permute(2, 0, 1).contiguous()


def fused(sm, loss):
    pa = p.detach().requires_grad_(True)
    out = gsp.generate_2D_gaussian_splatting_step((H, W), pa, scale, sm, dmax=dmax)
    if loss:
        (out * wgt).sum().backward()
    else:
        out.backward(wgt)


def unfused(sm, loss):
    pa = p.detach().requires_grad_(True)
    a5 = gsp._activate(pa)
    out = gsp.rendering_cuda_dmax(*a5, (H, W), gsp._step_size(scale, sm, 1.2, "scale_modify"), dev, dmax=dmax)
    if loss:
        (out * wgt).sum().backward()
    else:
        out.backward(wgt)


def bench(fn, n=50):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for kernel in ("gaussian", "tile"):
    gsp.BACKWARD_KERNEL = kernel
    for name, fn in (("fused", fused), ("unfused", unfused)):
        if kernel == "tile" and name == "unfused":
            continue
        for sm_name, sm in (("scale_modify on the GPU", sm_gpu), ("scale_modify as numbers", sm_py)):
            for loss in (False, True):
                dt = bench(lambda: fn(sm, loss))
                print(f"{name:8s} bwd={kernel:8s} {sm_name:24s} {'torch loss' if loss else 'direct grad':11s} N={p.shape[0]} {H}x{W}: "
                      f"{dt * 1e6:8.1f} us per fwd+bwd  ({H * W / dt / 1e6:.0f} HR Mpx/s)")
gsp.deferred_asserts.flush()
