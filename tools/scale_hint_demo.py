"""The scale hint of gsasr_amd.gaussian_splatting (round 5): fused inference / training step at Fea2GS's 16 Gaussians per LR pixel,
with and without it.   python tools/scale_hint_demo.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import gaussian_splatting as gsp, synthetic  # noqa: E402

dev = torch.device("cuda:0")


def ms(fn, n=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = 0.0
    a.record()
    while t < 60.0:
        fn(); fn()
        b.record(); b.synchronize(); t = a.elapsed_time(b)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(n):
            fn()
        b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best


print(f"{'shape':34s} {'what':10s} {'hint off ms':>11s} {'hint on ms':>11s} {'gain':>6s}")
for name, lr, scale, gpp in (("x8, 256^2 LR -> 2048^2, 16/LR px", 256, 8, 16), ("x6, 256^2 LR -> 1536^2, 16/LR px", 256, 6, 16),
                             ("x12, 128^2 LR -> 1536^2, 16/LR px", 128, 12, 16), ("x8, 256^2 LR -> 2048^2, 4/LR px", 256, 8, 4),
                             ("x8, 384^2 LR -> 3072^2, 1/LR px", 384, 8, 1)):
    H = W = lr * scale
    p = synthetic.gs_parameters(lr, lr, seed=0, gpp=gpp).to(dev)
    sm = torch.tensor([float(scale), float(scale)], device=dev)
    for what in ("inference", "fwd+bwd"):
        res = {}
        for hint in (False, True):
            gsp.SCALE_HINT = hint
            if what == "inference":
                def fn():
                    with torch.no_grad():
                        gsp.generate_2D_gaussian_splatting_step((H, W), p, scale, sm, dmax=0.1)
            else:
                pa = p.clone().requires_grad_(True)

                def fn():
                    pa.grad = None
                    gsp.generate_2D_gaussian_splatting_step((H, W), pa, scale, sm, dmax=0.1).sum().backward()
            res[hint] = ms(fn)
        print(f"{name:34s} {what:10s} {res[False]:11.4f} {res[True]:11.4f} {100 * (1 - res[True] / res[False]):5.1f}%", flush=True)
