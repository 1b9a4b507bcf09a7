#!/usr/bin/env python
"""Randomised self-consistency of the host entry points around the fused step (GPU, development aid):
    python tools/fuzz_host.py [cases] [seed]
  step  ==  step_buffer with a random chunk size (image and gradient; the buffer path is the unfused torch prologue +
            chunks accumulated into one image, every chunk with the cutoff of the whole set)
  step  ==  step under bf16 autocast (the fp32 boundary)
  step(sample_coords) == step()[:, rows, cols]       (value and gradient)
  module gscuda (both arities) == GSCUDA.apply of the matching package, forward and backward contracts (+= / =)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import gaussian_splatting as gsp, gscuda  # noqa: E402
from gsasr_amd.gs_cuda.gswrapper import GSCUDA as GS_U  # noqa: E402
from gsasr_amd.gs_cuda_dmax.gswrapper import GSCUDA as GS_D  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)


def rel(a, b):
    return float((a - b).abs().max()) / max(1e-30, float(b.abs().max()))


worst = {}


def note(k, v, lim, what):
    worst[k] = max(worst.get(k, 0.0), v)
    assert np.isfinite(v) and v <= lim, (what, k, v)


for case in range(cases):
    g = torch.Generator().manual_seed(case)
    H, W = int(rng.integers(2, 1400)), int(rng.integers(2, 1400))
    if rng.random() < 0.5:
        H, W = int(rng.integers(2, 120)), int(rng.integers(2, 120))
    n = int(rng.integers(1, 2500))
    p = torch.randn(n, 9, generator=g) * float(rng.choice([0.5, 1.5]))
    p[:, 7:9] = torch.rand(n, 2, generator=g) * 1.3 - 0.15
    p[:, 2].clamp_(-2.5, 2.5)
    p = p.to(dev)
    s = float(rng.uniform(1.0, 10.0))
    sm = torch.tensor([s, s], device=dev)
    kw = [dict(if_dmax=True, dmax_mode="fix", dmax=float(10 ** rng.uniform(-2.0, 0.3))), dict(if_dmax=True, dmax_mode="dynamic", dmax=float(rng.uniform(2, 60))),
          dict(if_dmax=False)][int(rng.integers(0, 3))]
    gsp.BACKWARD_KERNEL = ["auto", "gaussian", "tile"][int(rng.integers(0, 3))]
    what = (case, H, W, n, round(s, 3), kw, gsp.BACKWARD_KERNEL)
    wgt = torch.randn(3, H, W, generator=g).to(dev)
    pa = p.clone().requires_grad_(True)
    out = gsp.generate_2D_gaussian_splatting_step((H, W), pa, s, sm, **kw)
    (out * wgt).sum().backward()
    scale_i = max(1.0, float(out.detach().abs().max()))
    # chunked
    # The chunked path runs the UNFUSED prologue (torch expressions); the fused step's prologue kernel rounds exactly as
    # torch does (an ulp of a centre would be worth 1e-3 of a sub-pixel Gaussian's value), so all three agree tightly.
    pb = p.clone().requires_grad_(True)
    buf = gsp.generate_2D_gaussian_splatting_step_buffer((H, W), pb, s, sm, buffer_size=int(rng.integers(1, 2 * n + 2)), **kw)
    (buf * wgt).sum().backward()
    pu = p.clone().requires_grad_(True)
    act = gsp._activate(pu)
    dm_eff = gsp._resolve_dmax(kw.get("dmax", 25), kw.get("dmax_mode", "fix"), (H, W)) if kw["if_dmax"] else None
    unf = (gsp.rendering_cuda_dmax(*act, (H, W), 1.2 / s, dev, dmax=dm_eff) if kw["if_dmax"] else gsp.rendering_cuda(*act, (H, W), 1.2 / s, dev))
    (unf * wgt).sum().backward()
    note("buffer image (vs unchunked)", float((buf - unf).detach().abs().max()) / scale_i, 2e-5, what)
    note("buffer gradient (vs unchunked)", rel(pb.grad, pu.grad), 2e-4, what)
    note("fused vs unfused image", float((out - unf).detach().abs().max()) / scale_i, 2e-5, what)
    note("fused vs unfused gradient", rel(pa.grad, pu.grad), 2e-4, what)
    # autocast
    with torch.autocast("cuda", dtype=torch.bfloat16):
        oc = gsp.generate_2D_gaussian_splatting_step((H, W), p, s, sm, **kw)
    assert oc.dtype == torch.float32
    note("autocast image", float((oc - out.detach()).abs().max()) / scale_i, 1e-6, what)
    # sampled
    S = int(rng.integers(1, 3000))
    pts = torch.stack([torch.randint(0, H, (S,), generator=g), torch.randint(0, W, (S,), generator=g)], 1).to(dev)
    ws = torch.randn(3, S, generator=g).to(dev)
    pc = p.clone().requires_grad_(True)
    o1 = gsp.generate_2D_gaussian_splatting_step((H, W), pc, s, sm, sample_coords=pts, **kw)
    (o1 * ws).sum().backward()
    pd = p.clone().requires_grad_(True)
    o2 = gsp.generate_2D_gaussian_splatting_step((H, W), pd, s, sm, **kw)[:, pts[:, 0], pts[:, 1]]
    (o2 * ws).sum().backward()
    note("sampled value", float((o1 - o2).detach().abs().max()) / scale_i, 2e-5, what)
    # (floor: when no sampled point lies in any window the rendered path's gradient is exactly 0 and the sampled one holds
    # the tails below exp(-tau) it does not test for -- 1e-7 against nothing)
    note("sampled gradient", float((pc.grad - pd.grad).abs().max()) / max(1e-3, float(pd.grad.abs().max())), 5e-4, what)
    # module gscuda against GSCUDA.apply, on the kernel-frame tensors of this case
    sx, sy, rho, cxy, cwa = gsp._activate(p)
    sig, xy, col, _, _ = gsp._to_kernel_frame(sx, sy, rho, cxy, cwa, (H, W), 1.2 / s)
    sig, xy, col = sig.contiguous(), xy.contiguous(), col.contiguous()
    for dm in (None, float(10 ** rng.uniform(-2.0, 0.3))):
        base = torch.rand(H, W, 3, generator=g).to(dev)
        a_, b_, c_ = (t.clone().requires_grad_(True) for t in (sig, xy, col))
        ref = (GS_U.apply(a_, b_, c_, base.clone()) if dm is None else GS_D.apply(a_, b_, c_, base.clone(), dm))
        gi = wgt.permute(1, 2, 0).contiguous()
        ref.backward(gi)
        img = base.clone()
        if dm is None:
            gscuda.gs_render(sig, xy, col, img, n, H, W, 3)
        else:
            gscuda.gs_render(sig, xy, col, img, n, H, W, 3, dm)
        note("gscuda image", float((img - ref.detach()).abs().max()) / max(1.0, float(ref.detach().abs().max())), 1e-6, what)
        gs_ = [torch.full_like(t, 0.5) for t in (sig, xy, col)]      # dmax backward adds into these, the unbounded one overwrites
        if dm is None:
            gscuda.gs_render_backward(sig, xy, col, gi, *gs_, n, H, W, 3)
        else:
            gscuda.gs_render_backward(sig, xy, col, gi, *gs_, n, H, W, 3, dm)
        for t, r in zip(gs_, (a_.grad, b_.grad, c_.grad)):
            got = t if dm is None else t - 0.5
            e = max(0.0, float((got - r).abs().max()) - (0.0 if dm is None else 6e-8)) / max(1e-30, float(r.abs().max()))
            note("gscuda gradient", e, 2e-5, (what, dm))
gsp.deferred_asserts.flush()
print(f"{cases} cases ok: worst {worst}")
