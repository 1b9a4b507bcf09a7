#!/usr/bin/env python
"""Randomised check of the tile-list forward (round 5) against the search kernels and the oracle (GPU):
    python tools/fuzz_lists.py [cases] [seed]
Per case: a random image (small, GSASR-sized or thin), a random row band, Gaussians of GSASR's shape (about one LR pixel)
mixed with hairlines, large and off-image ones; bounded / unbounded; default / exact / no cutoff.  Rendered four ways --
search kernels (list_cap = -1), lists with the library's capacity, lists with a tiny capacity (most tiles overflow and fall
back to the search), each through the 8 x 16 and the wide 16 x 16 kernels -- and compared with each other (same sums in
another order) and with the oracle on a few rows."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import _cabi  # noqa: E402
from oracle import gs_oracle  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst = {"lists_vs_search": 0.0, "oracle": 0.0}
t0 = time.time()
done = 0
for case in range(cases):
    shape = int(rng.integers(0, 4))
    if shape == 0:
        H, W = int(rng.integers(17, 300)), int(rng.integers(17, 300))
    elif shape == 1:
        H, W = int(rng.integers(300, 1400)), int(rng.integers(300, 1400))
    elif shape == 2:
        H, W = int(rng.integers(2, 40)), int(rng.integers(500, 6000))
        if rng.random() < 0.5:
            H, W = W, H
    else:
        H, W = int(rng.integers(1000, 3000)), int(rng.integers(1000, 3000))
    # GSASR-shaped: one Gaussian per LR pixel of a grid `scale` times coarser, sigma about 0.3..0.8 LR pixels
    scale = float(rng.choice([2.0, 3.0, 4.0, 6.0, 8.0, 12.0, 24.0]))
    gpp = int(rng.choice([1, 1, 4, 16])) if H * W < 600000 else 1
    h_lr, w_lr = max(1, int(H / scale)), max(1, int(W / scale))
    n = min(h_lr * w_lr * gpp, 400000)
    cx = rng.uniform(-1.02, 1.02, n)
    cy = rng.uniform(-1.02, 1.02, n)
    sx = rng.uniform(0.25, 0.85, n) * scale / 1.2 * 2.0 / max(W - 1, 1)
    sy = rng.uniform(0.25, 0.85, n) * scale / 1.2 * 2.0 / max(H - 1, 1)
    rho = np.clip(rng.normal(0, 0.5, n), -0.995, 0.995)
    odd = rng.random(n) < 0.03            # hairlines, large and image-spanning ones among them
    sx[odd] = 10 ** rng.uniform(-3.5, 0.3, int(odd.sum()))
    sy[odd] = 10 ** rng.uniform(-3.5, 0.3, int(odd.sum()))
    sig = np.stack([sx, sy, rho], 1).astype(np.float32)
    xy = np.stack([cx, cy], 1).astype(np.float32)
    col = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    dmax = [None, float(10 ** rng.uniform(-2.0, 0.3))][int(rng.integers(0, 2))]
    cutoff = [0.0, 0.0, 104.0, -1.0][int(rng.integers(0, 4))]
    if cutoff != 0.0 and n * H * W > 3e10:
        cutoff = 0.0
    if rng.random() < 0.5:
        nb = int(min(H, rng.integers(1, 200)))
        r0 = int(rng.integers(0, H - nb + 1))
        rows = (r0, r0 + nb)
    else:
        rows = (0, H)
    nb = rows[1] - rows[0]
    a, b, c = (torch.from_numpy(x).to(dev) for x in (sig, xy, col))
    what = (case, H, W, rows, n, scale, gpp, dmax, cutoff)
    imgs = {}
    for wide in (0, _cabi.FLAG_FWD_NARROW, _cabi.FLAG_FWD_WIDE):
        for cap in (-1, 0, 64):
            if wide == 0 and cap == 64:
                continue
            plan = _cabi.plan(a, b, c, H, W, dmax, rows=rows, cutoff=cutoff, flags=_cabi.FLAG_FORWARD_ONLY | wide, list_cap=cap)
            img = torch.full((nb, W, 3), float("nan"), device=dev)
            _cabi.forward(plan, img, overwrite=True, flags=wide)
            imgs[(wide, cap)] = img
    ref = imgs[(0, -1)]
    top = max(1.0, float(ref.abs().max()))
    assert bool(torch.isfinite(ref).all()), what
    for k, v in imgs.items():
        e = float((v - ref).abs().max()) / top
        assert np.isfinite(e) and e <= 1e-5, (what, k, e)       # (same terms in another order: fp32 summation noise of ~100 terms per pixel)
        worst["lists_vs_search"] = max(worst["lists_vs_search"], e)
    # ... and the oracle on up to three rows of the band
    pick = sorted(set(int(x) for x in rng.integers(rows[0], rows[1], 3)))
    for r in pick:
        if n * W > 3e8:
            break
        want = gs_oracle.forward_f64(sig, xy, col, H, W, dmax, rows=(r, r + 1))
        for k in ((0, 0), (_cabi.FLAG_FWD_WIDE, 0), (_cabi.FLAG_FWD_NARROW, 64)):
            e = float(np.abs(imgs[k][r - rows[0]].cpu().numpy() - want[0]).max()) / top
            worst["oracle"] = max(worst["oracle"], e)
            assert e <= 1e-4, (what, k, r, e)
    done += 1
    if case % 10 == 9:
        print(f"{case + 1} cases, {time.time() - t0:.0f} s, worst {worst}", flush=True)
print(f"{done} cases ok: worst errors relative to max(1, image max) {worst}")
