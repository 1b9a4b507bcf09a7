#!/usr/bin/env python
"""The wide forward (16x16 sub-tiles, GSASR_FLAG_FWD_WIDE) against the 8x16 kernels (GSASR_FLAG_FWD_NARROW) on random
shapes, scales, row bands and boxes (GPU, development aid):   python tools/fuzz_wide.py [cases]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsasr_amd import _cabi, synthetic  # noqa: E402

CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
rng = np.random.default_rng(11)
worst = 0.0
for case in range(CASES):
    h_lr, w_lr = int(rng.integers(8, 90)), int(rng.integers(8, 90))
    scale = float(rng.choice([3.0, 4.0, 5.5, 8.0, 12.0, 24.0, 32.0]))
    gpp = int(rng.choice([1, 1, 1, 4, 16])) if h_lr * w_lr * scale * scale < 2e6 else 1
    sig, xy, col, H, W = synthetic.kernel_inputs(h_lr, w_lr, scale, seed=100 + case, gpp=gpp)
    dmax = [0.1, None, 0.02, 0.5][case % 4]
    r0 = int(rng.integers(0, H // 3)) if case % 2 else 0
    r1 = int(rng.integers(2 * H // 3, H)) if case % 2 else H
    chw = bool(case % 3 == 0)
    a, b, c = sig.to(dev), xy.to(dev), col.to(dev)
    plan = _cabi.plan(a, b, c, H, W, dmax, rows=(r0, r1), flags=_cabi.FLAG_FORWARD_ONLY)
    imgs = []
    for flag in (_cabi.FLAG_FWD_WIDE, _cabi.FLAG_FWD_NARROW):
        img = torch.full((3, r1 - r0, W) if chw else (r1 - r0, W, 3), float("nan"), device=dev)
        _cabi.forward(plan, img, overwrite=True, chw=chw, flags=flag)
        assert bool(torch.isfinite(img).all()), (case, flag)
        imgs.append(img)
    e = float((imgs[0] - imgs[1]).abs().max()) / max(1.0, float(imgs[1].abs().max()))
    worst = max(worst, e)
    assert e <= 5e-6, (case, e)
    if case % 10 == 9:
        print(f"{case + 1} cases, worst relative difference wide vs narrow {worst:.2e}", flush=True)
print(f"{CASES} cases ok: worst relative difference wide vs narrow {worst:.2e}")
