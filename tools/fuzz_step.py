#!/usr/bin/env python
"""Randomised check of the host API (fused prologue + plan + splat, autograd to the raw decoder parameters) against
oracle(prologue(.)) and the oracle's analytic backward chained through the reference prologue in double, at image sizes
the unit tests do not reach (GPU, development aid):   python tools/fuzz_step.py [cases] [seed]

The loss weights only a random row band of the image (the oracle's cost), the rest of the image is checked to be finite."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import _cabi, gaussian_splatting as gsp  # noqa: E402
from oracle import gs_oracle, host_ref  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(seed)
worst_img = worst_g = 0.0
t0 = time.time()
for case in range(cases):
    shape = int(rng.integers(0, 3))
    H, W = [(int(rng.integers(2, 80)), int(rng.integers(400, 3000))), (int(rng.integers(400, 3000)), int(rng.integers(2, 80))),
            (int(rng.integers(60, 1500)), int(rng.integers(60, 1500)))][shape]
    n = int(rng.integers(1, 1200))
    nb = int(min(H, rng.integers(1, 24)))
    r0 = int(rng.integers(0, H - nb + 1))
    g = torch.Generator().manual_seed(1000 * seed + case)
    p = torch.randn(n, 9, generator=g) * float(rng.choice([0.5, 1.5, 3.0]))
    p[:, 7:9] = torch.rand(n, 2, generator=g) * 1.4 - 0.2
    # (|rho| -> 1 makes the fp32 exponent of the FORWARD ill-conditioned -- three terms of size u^2 / (1 - rho^2) cancel, in the
    # reference's kernels as here: at rho = 0.999999 an image value moves by 0.3% against the double-precision truth, which is
    # not what this check is after; tools/fuzz_sample.py clamps the same way, the gradient tests have their own
    # conditioning-aware bar)
    p[:, 2].clamp_(-3.0, 3.0)
    near = torch.rand(n, generator=g) < 0.7
    p[near, 8] = ((r0 + nb / 2) / H + torch.randn(int(near.sum()), generator=g) * (2.0 * nb / H + 0.02)).clamp(-0.2, 1.2)
    s = float(rng.uniform(1.0, 12.0))
    sm_gpu = rng.random() < 0.5
    mode = int(rng.integers(0, 3))
    kw = [dict(if_dmax=True, dmax_mode="fix", dmax=float(10 ** rng.uniform(-2.0, 0.3))), dict(if_dmax=True, dmax_mode="dynamic", dmax=float(rng.uniform(2, 60))),
          dict(if_dmax=False)][mode]
    gsp.BACKWARD_KERNEL = ["auto", "gaussian", "tile"][int(rng.integers(0, 3))]
    what = (case, H, W, (r0, r0 + nb), n, round(s, 3), kw, gsp.BACKWARD_KERNEL, "sm on gpu" if sm_gpu else "sm numbers")
    sm = torch.tensor([s, s])
    pg = p.clone().to(dev).requires_grad_(True)
    out = gsp.generate_2D_gaussian_splatting_step((H, W), pg, s, sm.to(dev) if sm_gpu else (s, s), **kw)
    assert out.shape == (3, H, W) and torch.isfinite(out).all(), what
    sig_r, xy_r, col_r, dmax = host_ref.prologue(p, (H, W), sm, dmax=kw.get("dmax", 25), dmax_mode=kw.get("dmax_mode", "fix"))
    if not kw["if_dmax"]:
        dmax = None
    # The kernel-frame tensors as the GPU prologue rounds them (the stand-alone gsasr_prologue_forward: same arithmetic as
    # the fused one).  They may differ from the torch expression by an ulp, and a sub-pixel Gaussian (sigma < 0.1 px, which
    # randn x 3 parameters produce) turns one ulp of its centre (2e-4 px on a 3000-px image) into > 1e-4 of its value: the
    # rasterizer is held to the oracle on ITS inputs, the prologue to the reference expression on its own.
    sig, xy, col = (t.cpu() for t in _cabi.prologue_forward(pg.detach(), torch.tensor([1.2 / s], device=dev), H, W))
    for a_, b_, tn in ((sig, sig_r, "sigmas"), (xy, xy_r, "coords"), (col, col_r, "colors")):
        ep = float(((a_ - b_).abs() / b_.abs().clamp_min(1.0)).max())     # (CPU torch divides truly, the GPU multiplies by reciprocals: an ulp)
        assert ep <= 5e-7, (what, "prologue", tn, ep)
    rows = (r0, r0 + nb)
    ref = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, dmax, rows=rows)
    ei = float(np.abs(out.detach()[:, r0:r0 + nb].permute(1, 2, 0).cpu().numpy() - ref).max()) / max(1.0, float(np.abs(ref).max()))
    worst_img = max(worst_img, ei)
    assert ei <= 1e-4, (what, "image", ei)
    wgt = torch.randn(nb, W, 3, generator=g)
    full = torch.zeros(3, H, W, device=dev)
    full[:, r0:r0 + nb] = wgt.permute(2, 0, 1).to(dev)
    (out * full).sum().backward()
    pr = p.clone().double().requires_grad_(True)
    s2, x2, c2, _ = host_ref.prologue(pr, (H, W), sm.double(), dmax=kw.get("dmax", 25), dmax_mode=kw.get("dmax_mode", "fix"))
    gk = gs_oracle.backward_f64(sig.numpy(), xy.numpy(), col.numpy(), wgt.numpy(), dmax, h=H, rows=rows)
    torch.autograd.backward([s2, x2, c2], [torch.from_numpy(a) for a in gk])
    want = pr.grad.numpy()
    if float(np.abs(ref).max()) < 1e-3:
        continue                                     # nothing visible in the band: the gradient is made of culled tails
    eg = float(np.abs(pg.grad.cpu().numpy() - want).max()) / max(1e-30, float(np.abs(want).max()))
    worst_g = max(worst_g, eg)
    assert np.isfinite(pg.grad.cpu().numpy()).all() and eg <= 2e-4, (what, "gradient", eg)
    if case % 20 == 19:
        print(f"{case + 1} cases, {time.time() - t0:.0f} s, worst image err (of max(1, max|image|)) {worst_img:.2e}, worst gradient rel err {worst_g:.2e}", flush=True)
gsp.deferred_asserts.flush()
print(f"{cases} cases ok: worst image err (of max(1, max|image|)) {worst_img:.2e}, worst gradient rel err {worst_g:.2e}")
