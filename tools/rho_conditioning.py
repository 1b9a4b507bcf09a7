#!/usr/bin/env python
"""Per-Gaussian gradient error of the HIP backward against the oracle's f64 truth as |rho| -> 1 (the host prologue emits
0.999999 * tanh: saturated correlations are reachable in training).  Prints, per band of kappa = 1 - rho^2, the worst
row error relative to the row's own max-abs, for both backward kernels.  Development aid behind the tolerance of
tests/test_hip_parity.py::test_saturated_rho_per_gaussian_gradients."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import _cabi, synthetic  # noqa: E402
from oracle import gs_oracle  # noqa: E402

dev = torch.device("cuda:0")
sig, xy, col, H, W = synthetic.kernel_inputs(24, 24, 4.0, seed=3)
n = sig.shape[0]
g = torch.Generator().manual_seed(7)
kap = 10.0 ** (-6.0 * torch.rand(n, generator=g))            # kappa in [1e-6, 1]
sign = torch.where(torch.rand(n, generator=g) < 0.5, -1.0, 1.0)
rho = (sign * torch.sqrt(1.0 - kap)).float().clamp(-0.999999, 0.999999)
sig[:, 2] = rho
wgt = synthetic.grad_image(H, W, 4)
s, c, k, w_ = sig.numpy(), xy.numpy(), col.numpy(), wgt.numpy()
want = gs_oracle.backward_f64(s, c, k, w_, 0.3)
kappa = 1.0 - s[:, 2].astype(np.float64) ** 2
for flags, name in ((_cabi.FLAG_BWD_GAUSSIAN, "gaussian"), (_cabi.FLAG_BWD_TILE, "tile")):
    a, b, cc = (t.to(dev) for t in (sig, xy, col))
    plan = _cabi.plan(a, b, cc, H, W, 0.3, flags=flags)
    gs = [torch.empty_like(t) for t in (a, b, cc)]
    _cabi.backward(plan, a, b, cc, wgt.to(dev), *gs, overwrite=True)
    torch.cuda.synchronize()
    for got, ref, tn in zip(gs, want, ("sigmas", "coords", "colors")):
        got = got.cpu().numpy()
        rel = np.abs(got - ref).max(axis=1) / (np.abs(ref).max(axis=1) + 1e-5 * np.abs(ref).max() + 1e-30)
        line = []
        for lo, hi in ((1e-1, 1.1), (2e-2, 1e-1), (1e-3, 2e-2), (1e-4, 1e-3), (1e-5, 1e-4), (1e-7, 1e-5)):
            m = (kappa >= lo) & (kappa < hi)
            line.append(f"k<{hi:.0e}: {rel[m].max() if m.any() else 0:.1e} (x kappa {np.max(rel[m] * kappa[m]) if m.any() else 0:.1e})")
        print(f"{name:8s} {tn:7s} " + "  ".join(line))
