#!/usr/bin/env python
"""Randomised check of the kernel-choice registry (round 5; gsasr_set_kernel_choice / gsasr_amd.tune) on the GPU:
    python tools/fuzz_choices.py [cases] [seed]
Per case: a random image and Gaussian set (as tools/fuzz_lists.py: GSASR-shaped mixed with hairlines, large and off-image ones;
1 / 4 / 16 per LR pixel; bounded / unbounded), rendered and differentiated by the library's own rule, then again under every one of
the eight registered combinations {8 x 16 | 16 x 16 forward} x {Gaussian- | tile-stationary backward} x {lists | search} -- through
plain dims that carry no choice of their own.  All nine must agree (same sums in another order); a few rows go to the oracle."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import _cabi, tune  # noqa: E402
from oracle import gs_oracle  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst = {"image": 0.0, "gradient": 0.0, "oracle": 0.0}
t0 = time.time()
for case in range(cases):
    shape = int(rng.integers(0, 3))
    if shape == 0:
        H, W = int(rng.integers(17, 300)), int(rng.integers(17, 300))
    elif shape == 1:
        H, W = int(rng.integers(300, 1400)), int(rng.integers(300, 1400))
    else:
        H, W = int(rng.integers(1000, 2600)), int(rng.integers(1000, 2600))
    scale = float(rng.choice([2.0, 3.0, 4.0, 6.0, 8.0, 12.0]))
    gpp = int(rng.choice([1, 1, 4, 16])) if H * W < 500000 else 1
    h_lr, w_lr = max(1, int(H / scale)), max(1, int(W / scale))
    n = min(h_lr * w_lr * gpp, 300000)
    size = float(rng.choice([0.3, 1.0, 1.0, 1.8]))      # small / GSASR-sized / saturated Gaussians
    sx = rng.uniform(0.25, 0.85, n) * size * scale / 1.2 * 2.0 / max(W - 1, 1)
    sy = rng.uniform(0.25, 0.85, n) * size * scale / 1.2 * 2.0 / max(H - 1, 1)
    odd = rng.random(n) < 0.02
    sx[odd] = 10 ** rng.uniform(-3.5, 0.0, int(odd.sum()))
    sy[odd] = 10 ** rng.uniform(-3.5, 0.0, int(odd.sum()))
    sig = np.stack([sx, sy, np.clip(rng.normal(0, 0.5, n), -0.995, 0.995)], 1).astype(np.float32)
    xy = rng.uniform(-1.02, 1.02, (n, 2)).astype(np.float32)
    col = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    dmax = [None, float(10 ** rng.uniform(-1.5, 0.0))][int(rng.integers(0, 2))]
    wgt = rng.uniform(0, 1, (H, W, 3)).astype(np.float32)
    s, c, k, g = (torch.from_numpy(a).to(dev) for a in (sig, xy, col, wgt))

    def run():
        p = _cabi.plan(s, c, k, H, W, dmax)
        img = torch.empty(H, W, 3, device=dev)
        _cabi.forward(p, img, overwrite=True)
        return (img, *_cabi.backward_new(p, s, c, k, g))

    tune.reset()
    ref = run()
    top_i = max(1.0, float(ref[0].abs().max()))
    shp = _cabi.make_dims(n, H, W, dmax)
    for name, flags, cap in tune.candidates(n, W, H, True)[1:]:
        _cabi.set_kernel_choice(shp, flags, cap)
        out = run()
        ei = float((out[0] - ref[0]).abs().max()) / top_i
        eg = max(float((a - b).abs().max()) / max(1.0, float(b.abs().max())) for a, b in zip(out[1:], ref[1:]))
        worst["image"], worst["gradient"] = max(worst["image"], ei), max(worst["gradient"], eg)
        if not (ei <= 1e-5 and eg <= 5e-5):
            print(f"case {case} {name}: H={H} W={W} n={n} scale={scale} gpp={gpp} size={size} dmax={dmax}: image {ei:.3e} gradient {eg:.3e}")
            sys.exit(1)
    tune.reset()
    if n * W <= 1.5e8:       # (a row of the oracle costs n x W exponentials on the host)
        for r in sorted(set(int(r) for r in rng.integers(0, H, 2))):
            o = gs_oracle.forward_f64(sig, xy, col, H, W, dmax, rows=(r, r + 1))
            e = float(np.abs(ref[0][r].cpu().numpy() - o[0]).max()) / top_i
            worst["oracle"] = max(worst["oracle"], e)
            if not e <= 2e-5:
                print(f"case {case}: row {r} against the oracle {e:.3e}")
                sys.exit(1)
print(f"{cases} cases, {time.time() - t0:.0f} s, worst {worst}")
print(f"{cases} cases ok: worst differences between the nine combinations (relative to max(1, largest value)) {worst}")
