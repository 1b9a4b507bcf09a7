#!/usr/bin/env python
"""Host API forward + backward per shape with the three backward kernels (which one `BACKWARD_KERNEL = "auto"` should pick).
Development aid."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import gaussian_splatting as gsp, synthetic  # noqa: E402

dev = torch.device("cuda:0")


def bench(fn, n=20, repeats=4):
    for _ in range(8):
        fn()
    best = float("inf")
    for _ in range(repeats):       # (best of a few batches: allocator growth / frees of the previous shape land in one of them)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best


for h_lr, w_lr, scale, gpp, dmax, B in ((256, 256, 4.0, 1, 0.1, 1), (128, 128, 4.0, 1, 0.5, 1), (48, 48, 4.0, 16, 0.5, 1), (48, 48, 4.0, 16, 0.5, 16),
                                        (256, 256, 2.0, 1, 0.5, 1), (128, 128, 8.0, 1, 0.1, 1), (512, 512, 8.0, 1, 0.1, 1), (64, 64, 3.0, 16, 0.5, 8),
                                        (96, 96, 2.0, 16, 0.5, 4), (64, 64, 12.0, 1, 0.1, 1), (256, 256, 4.0, 16, 0.1, 1), (320, 320, 4.0, 16, 0.1, 1),
                                        (192, 192, 4.0, 16, 0.5, 1), (384, 384, 3.0, 16, 0.1, 1), (64, 64, 4.0, 16, 0.5, 16)):
    H, W = int(h_lr * scale), int(w_lr * scale)
    res = {}
    for kernel in ("gaussian", "tile", "home"):
        gsp.BACKWARD_KERNEL = kernel
        if B == 1:
            p = synthetic.gs_parameters(h_lr, w_lr, seed=0, gpp=gpp).to(dev)
            wgt = torch.rand(3, H, W, device=dev)

            def fn():
                pa = p.detach().requires_grad_(True)
                gsp.generate_2D_gaussian_splatting_step((H, W), pa, scale, (scale, scale), dmax=dmax).backward(wgt)
        else:
            p = torch.stack([synthetic.gs_parameters(h_lr, w_lr, seed=b, gpp=gpp) for b in range(B)]).to(dev)
            wgt = torch.rand(B, 3, H, W, device=dev)

            def fn():
                pa = p.detach().requires_grad_(True)
                gsp.generate_2D_gaussian_splatting_batch([(H, W)] * B, pa, [scale] * B, [(scale, scale)] * B, dmax=dmax).backward(wgt)
        res[kernel] = bench(fn)
    n = h_lr * w_lr * gpp
    print(f"B={B:2d} {H}x{W} N={n:7d} px/G={H * W / n:6.1f} dmax={dmax}: gaussian {res['gaussian'] * 1e6:8.1f} us  tile {res['tile'] * 1e6:8.1f} us  "
          f"home {res['home'] * 1e6:8.1f} us  -> {min(res, key=res.get)}")
