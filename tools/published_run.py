"""The reference's one published workload (utils/gs_cuda/profile.py:104-113) as a stand-alone command for rocprofv3:
    python tools/published_run.py [--cutoff 0|-1] [--calls 10]
plan + forward of the unbounded op, 512 x 512 image, 262 144 Gaussians with sigma in [0, 1) of the grid."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from benchlib import published_inputs  # noqa: E402
from gsasr_amd import _cabi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cutoff", type=float, default=0.0)
ap.add_argument("--calls", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
sig, xy, col, H, W = published_inputs()
s, c, k = sig.to(dev), xy.to(dev), col.to(dev)
img = torch.empty(H, W, 3, device=dev)
for i in range(a.calls + 2):
    if i == 2:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    plan = _cabi.plan(s, c, k, H, W, None, cutoff=a.cutoff, flags=_cabi.FLAG_FORWARD_ONLY)
    _cabi.forward(plan, img, overwrite=True)
torch.cuda.synchronize()
print(f"cutoff {a.cutoff}: {(time.perf_counter() - t0) / a.calls * 1e3:.2f} ms per plan + forward; image max {float(img.max()):.1f}")
