"""gsasr_amd.tune.tune_step on the fused single-image entry point: does gaussian_splatting._tile_backward's rule still pick the faster
backward?   python tools/tune_step_demo.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import synthetic, tune  # noqa: E402
from gsasr_amd.gaussian_splatting import _tile_backward  # noqa: E402

dev = torch.device("cuda:0")
print(f"{'shape':26s} {'rule':9s} {'default ms':>10s}  " + " ".join(f"{n:>16s}" for n in ("gaussian-search", "gaussian-lists", "tile-search", "tile-lists")) + "   picks")
for name, lr, scale, gpp, dmax in (("x4 1024^2 (config 2)", 256, 4, 1, 0.1), ("x2 1024^2", 512, 2, 1, 0.1), ("x3 768^2", 256, 3, 1, 0.1), ("x8 2048^2", 256, 8, 1, 0.1),
                                   ("x4 192^2 crop, 16/LR px", 48, 4, 16, 0.5), ("x4 512^2, 16/LR px", 128, 4, 16, 0.1), ("x4 1024^2, 16/LR px", 256, 4, 16, 0.1),
                                   ("x12 1536^2", 128, 12, 1, 0.1)):
    H = W = lr * scale
    p = synthetic.gs_parameters(lr, lr, seed=0, gpp=gpp).to(dev)
    step = torch.tensor([1.2 / scale], device=dev)
    res = tune.tune_step(p, step, H, W, dmax, register=False)
    rule = "tile" if _tile_backward(H * W, p.shape[0]) else "gaussian"
    print(f"{name:26s} {rule:9s} {res.ms['default']:10.4f}  " + " ".join(f"{res.ms.get(n, float('nan')):16.4f}" for n in ("gaussian-search", "gaussian-lists", "tile-search", "tile-lists")) + f"   {res.name}", flush=True)
    tune.reset()
