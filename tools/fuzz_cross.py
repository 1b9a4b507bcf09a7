#!/usr/bin/env python
"""Randomised cross-check of the kernels against each other and against the oracle at sizes the small fuzzers never
reach (GPU, development aid):   python tools/fuzz_cross.py [cases] [seed]

Per case: a random image (up to ~5000 px a side, sometimes very thin), a random row band, Gaussians from hairlines to
several image widths, dmax or none, default / exact / no cutoff.  Checked:
  forward    band render  ==  oracle (f64) on a few rows of the band
  backward   Gaussian-stationary, tile-stationary (slots), tile-stationary (atomics), home-tile: each against the oracle's
             gradient of the same band (upstream gradient dense or sparse)
The sampled-pixel fuzzer found the one bug of round 3 this way (a window wider than 2048 px in the tile backward)."""
import ctypes
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
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
FLAGS = {"gaussian": _cabi.FLAG_BWD_GAUSSIAN, "tile": _cabi.FLAG_BWD_TILE, "atomic": _cabi.FLAG_BWD_TILE | _cabi.FLAG_BWD_ATOMIC,
         "home": _cabi.FLAG_BWD_HOME}
worst = {"img": 0.0, "gaussian": 0.0, "tile": 0.0, "atomic": 0.0, "home": 0.0}
t0 = time.time()
for case in range(cases):
    shape = rng.integers(0, 5)
    big = [int(rng.integers(600, 5000)), int(rng.integers(600, 5000))]
    if shape == 0:
        H, W = int(rng.integers(2, 64)), big[1]
    elif shape == 1:
        H, W = big[0], int(rng.integers(2, 64))
    elif shape == 2:
        H, W = int(rng.integers(100, 1200)), int(rng.integers(100, 1200))
    elif shape == 3:
        H, W = big
    else:                                              # up to the largest side the ABI takes (32767), the other side thin
        H, W = int(rng.integers(20000, 32768)), int(rng.integers(2, 40))
        if rng.random() < 0.5:
            H, W = W, H
    n = int(rng.integers(1, 1500))
    nb = int(min(H, rng.integers(1, 40)))             # rows of the band (the oracle's cost)
    r0 = int(rng.integers(0, H - nb + 1))
    rows = (r0, r0 + nb)
    lo = float(rng.choice([-4.0, -3.0, -2.0]))
    sig = np.stack([10 ** rng.uniform(lo, 0.5, n), 10 ** rng.uniform(lo, 0.5, n), np.clip(rng.normal(0, 0.6, n), -0.999, 0.999)], 1).astype(np.float32)
    xy = rng.uniform(-1.3, 1.3, (n, 2)).astype(np.float32)
    # most centres near the band, so that the band sees work
    yc = (2.0 * (r0 + nb / 2) / max(H - 1, 1) - 1.0)
    near = rng.random(n) < 0.7
    xy[near, 1] = np.clip(yc + rng.normal(0, 3.0 * nb / H + 0.02, int(near.sum())), -1.3, 1.3).astype(np.float32)
    col = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    dmax = [None, float(10 ** rng.uniform(-2.5, 0.5))][int(rng.integers(0, 2))]
    cutoff = [0.0, 104.0, -1.0][int(rng.integers(0, 3))]
    wgt = rng.normal(0, 1, (nb, W, 3)).astype(np.float32)
    sparse = rng.random() < 0.5
    if sparse and cutoff != 0.0:                       # sparse upstream gradient: only where no term is culled -- under the
        wgt *= (rng.random((nb, W, 1)) < 0.01)         # default cutoff a gradient made of tail terms alone is legitimately dropped
    what = (case, H, W, rows, n, dmax, cutoff)
    a, b, c = (torch.from_numpy(x).to(dev) for x in (sig, xy, col))
    gw = torch.from_numpy(wgt).to(dev)
    ref_img = gs_oracle.forward_f64(sig, xy, col, H, W, dmax, rows=rows)
    want = gs_oracle.backward_f64(sig, xy, col, wgt, dmax, h=H, rows=rows)
    scale = max(1.0, float(np.abs(ref_img).max()))
    if cutoff == 0.0 and float(np.abs(ref_img).max()) < 1e-3:
        continue                                       # nothing of these Gaussians is visible in the band: tails only
    for name, flag in FLAGS.items():
        plan = _cabi.plan(a, b, c, H, W, dmax, rows=rows, cutoff=cutoff, flags=flag)
        if name == "gaussian":
            # forward variants: stored HWC, stored planar CHW, accumulated into a canvas, forward-only plan
            fv = int(rng.integers(0, 4))
            fplan = _cabi.plan(a, b, c, H, W, dmax, rows=rows, cutoff=cutoff, flags=_cabi.FLAG_FORWARD_ONLY) if fv == 3 else plan
            if fv == 1:
                img = torch.empty(3, nb, W, device=dev)
                _cabi.forward(fplan, img, overwrite=True, chw=True)
                got_img = img.permute(1, 2, 0).cpu().numpy()
            elif fv == 2:
                base = torch.from_numpy(rng.normal(0, 1, (nb, W, 3)).astype(np.float32)).to(dev)
                img = base.clone()
                _cabi.forward(fplan, img, overwrite=False)
                got_img = (img - base).cpu().numpy()
            else:
                img = torch.empty(nb, W, 3, device=dev)
                _cabi.forward(fplan, img, overwrite=True)
                got_img = img.cpu().numpy()
            ei = float(np.abs(got_img - ref_img).max()) / (scale if fv != 2 else scale + 4.0)   # (accumulated: rounding of the sum)
            worst["img"] = max(worst["img"], ei)
            assert np.isfinite(ei) and ei <= 1e-4, (what, "forward variant", fv, ei)
        # backward variants: stored / accumulated into the caller's gradients; the tile kernels also on a planar gradient
        acc = rng.random() < 0.3
        g = [torch.full_like(t, 0.25 if acc else float("nan")) for t in (a, b, c)]
        if name in ("tile", "atomic") and rng.random() < 0.4:
            gchw = gw.permute(2, 0, 1).contiguous()
            d = type(plan.dims).from_buffer_copy(plan.dims)
            d.flags |= _cabi.FLAG_CHW_GRAD | (0 if acc else _cabi.FLAG_OVERWRITE_GRADS)
            _cabi.check(_cabi.lib().gsasr_splat_backward(a.data_ptr(), b.data_ptr(), c.data_ptr(), gchw.data_ptr(), g[0].data_ptr(),
                                                         g[1].data_ptr(), g[2].data_ptr(), ctypes.byref(d), plan.workspace.data_ptr(),
                                                         plan.workspace.numel(), _cabi._stream(dev)), "gsasr_splat_backward")
        else:
            _cabi.backward(plan, a, b, c, gw, *g, overwrite=not acc)
        if acc:
            g = [t - 0.25 for t in g]
        for got, w_, tn in zip(g, want, ("sigmas", "coords", "colors")):
            got = got.cpu().numpy()
            assert np.isfinite(got).all(), (what, name, tn, "non-finite")
            e = max(0.0, float(np.abs(got - w_).max()) - (6e-8 if acc else 0.0)) / max(1e-30, float(np.abs(w_).max()))
            worst[name] = max(worst[name], e)
            assert e <= 2e-4, (what, name, tn, e)
    if case % 10 == 9:
        print(f"{case + 1} cases, {time.time() - t0:.0f} s, worst {worst}", flush=True)
print(f"{cases} cases ok: worst relative errors {worst}")
