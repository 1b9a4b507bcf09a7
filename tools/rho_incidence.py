#!/usr/bin/env python
"""How often does the host prologue (rho = 0.999999 tanh(p), utils/gaussian_splatting.py:176) land where this package and the
reference's fp32 arithmetic part ways, and by how much?  (VERDICT r5 weak #1.)

The HIP kernels evaluate the completed square of the exponent; the reference (gs.cu:33-56, restated by oracle.forward_f32)
evaluates the monomial form, which cancels terms of size u^2 / (1 - rho^2) in fp32.  For decoder outputs p ~ N(0, s_p),
s_p in {0.5 (bench.py's synthetic Gaussians), 1, 2, 4}, on BASELINE config 2's shape (256^2 LR x4, a band of its rows):
  incidence  : the share of Gaussians with 1 - rho^2 below 1e-2 / 1e-3 / 1e-4 / 1e-5,
  image      : max |HIP - f64 truth|, max |reference fp32 - f64 truth|, max |HIP - reference fp32| over the band.
Run on the GPU box: python tools/rho_incidence.py  (the oracle runs on the host cores)."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import _cabi, synthetic  # noqa: E402
from gsasr_amd import gaussian_splatting as gsp  # noqa: E402
from oracle import gs_oracle  # noqa: E402

dev = torch.device("cuda:0")
h_lr = w_lr = 256
scale = 4.0
H = W = 1024
rows = (448, 576)
print(f"config-2 shape: {h_lr}x{w_lr} LR x{scale:g}, {h_lr * w_lr} Gaussians, dmax 0.1, image rows {rows}")
print("analytic: 1 - rho^2 < k  <=>  |p| > atanh(sqrt(1 - k) / 0.999999):",
      {k: round(math.atanh(min(math.sqrt(1 - k) / 0.999999, 1 - 1e-16)), 2) for k in (1e-2, 1e-3, 1e-4, 1e-5)})
for s_p in (0.5, 1.0, 2.0, 4.0):
    p = synthetic.gs_parameters(h_lr, w_lr, seed=0)
    g = torch.Generator().manual_seed(17)
    p[:, 2] = s_p * torch.randn(p.shape[0], generator=g)
    sx, sy, rho, xy, col = gsp._activate(p)
    sig, xk, ck, _, _ = gsp._to_kernel_frame(sx, sy, rho, xy, col, (H, W), 1.2 / scale)
    kappa = 1.0 - sig[:, 2].double() ** 2
    inc = {k: float((kappa < k).double().mean()) for k in (1e-2, 1e-3, 1e-4, 1e-5)}
    a, b, c = (t.contiguous().to(dev) for t in (sig, xk, ck))
    plan = _cabi.plan(a, b, c, H, W, 0.1, rows=rows, flags=_cabi.FLAG_FORWARD_ONLY)
    img = torch.empty(rows[1] - rows[0], W, 3, device=dev)
    _cabi.forward(plan, img, overwrite=True)
    got = img.cpu().numpy()
    s_, x_, c_ = sig.numpy(), xk.numpy(), ck.numpy()
    t64 = gs_oracle.forward_f64(s_, x_, c_, H, W, 0.1, rows=rows)
    r32 = gs_oracle.forward_f32(s_, x_, c_, H, W, 0.1, rows=rows, use_fma=True)
    print(f"s_p = {s_p:3.1f}: share of Gaussians with 1-rho^2 < 1e-2/1e-3/1e-4/1e-5 = "
          f"{inc[1e-2]:.2e} / {inc[1e-3]:.2e} / {inc[1e-4]:.2e} / {inc[1e-5]:.2e};  max|HIP - truth| = {np.abs(got - t64).max():.2e}, "
          f"max|reference fp32 - truth| = {np.abs(r32 - t64).max():.2e}, max|HIP - reference fp32| = {np.abs(got - r32).max():.2e} "
          f"(image max {np.abs(t64).max():.2f})")
