#!/usr/bin/env python
"""Size-independent properties at sizes no oracle run reaches (GPU):   python tools/big_n_check.py LR GAUSSIANS_PER_LR_PX SCALE
   e.g. 1024 16 4 = 16.7 M Gaussians on 4096^2;  2048 1 12 = 4.2 M Gaussians on 24576^2 (a 7 GB image)
  - a Gaussian's gradient does not depend on the other Gaussians: rows of every 64th Gaussian == a run with only those
  - both backward kernels, and the images of their plans, agree
  - the forward is additive over a split of the Gaussians (second half accumulated into the first half's image)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import _cabi, synthetic
dev = torch.device("cuda:0")
lr, gpp, scale = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
sig, xy, col, H, W = synthetic.kernel_inputs(lr, lr, scale, seed=0, gpp=gpp)
n = sig.shape[0]
print("N", n, "image", H, W, flush=True)
a, b, c = sig.to(dev), xy.to(dev), col.to(dev)
wgt = torch.randn(H, W, 3, device=dev)
sel = torch.arange(0, n, 64, device=dev)
sub = [t[sel].contiguous() for t in (a, b, c)]
res = {}
for name, flag in (("gaussian", _cabi.FLAG_BWD_GAUSSIAN), ("tile", _cabi.FLAG_BWD_TILE)):
    plan = _cabi.plan(a, b, c, H, W, 0.1, flags=flag)
    img = torch.empty(H, W, 3, device=dev); _cabi.forward(plan, img, overwrite=True)
    g = [torch.empty_like(t) for t in (a, b, c)]
    torch.cuda.synchronize(); t0 = time.time()
    _cabi.backward(plan, a, b, c, wgt, *g, overwrite=True)
    torch.cuda.synchronize(); print(name, "backward %.1f ms" % ((time.time() - t0) * 1e3), "image finite", bool(torch.isfinite(img).all()), "mean", float(img.mean()), flush=True)
    # the subset run under the adaptive default as well: the backward's window is that of min(tau', GSASR_SPLAT_GRAD_TAU) in
    # both runs (an explicit cutoff would be used as given by the subset run alone, and the two would differ by the 1e-5 of
    # a Gaussian's mass that lies between the two ellipses)
    tau = _cabi.resolve_cutoff(0.0, n)
    ps = _cabi.plan(*sub, H, W, 0.1, flags=flag)
    gs = [torch.empty_like(t) for t in sub]
    _cabi.backward(ps, *sub, wgt, *gs, overwrite=True)
    for t, u, tn in zip(g, gs, ("sigmas", "coords", "colors")):
        e = float((t[sel] - u).abs().max()) / float(u.abs().max())
        print("  ", tn, "gradient rows of every 64th Gaussian vs a run with only those: rel", e)
        assert e <= 1e-5
    res[name] = img
    del plan, ps
print("images gaussian-plan vs tile-plan:", float((res["gaussian"] - res["tile"]).abs().max()))
# additivity of the forward over a split of the Gaussians
half = n // 2
pa = _cabi.plan(a[:half].contiguous(), b[:half].contiguous(), c[:half].contiguous(), H, W, 0.1, cutoff=tau, flags=_cabi.FLAG_FORWARD_ONLY)
i1 = torch.empty(H, W, 3, device=dev); _cabi.forward(pa, i1, overwrite=True)
pb = _cabi.plan(a[half:].contiguous(), b[half:].contiguous(), c[half:].contiguous(), H, W, 0.1, cutoff=tau, flags=_cabi.FLAG_FORWARD_ONLY)
_cabi.forward(pb, i1, overwrite=False)
e = float((i1 - res["gaussian"]).abs().max()) / float(res["gaussian"].abs().max())
print("forward additivity over two halves: rel", e); assert e <= 1e-5
