#!/usr/bin/env python
"""Compare the backward kernels of libgsasr_splat against each other per Gaussian (development aid).

    python tools/bwd_compare.py [h_lr w_lr scale gpp dmax]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import _cabi, synthetic  # noqa: E402

h_lr, w_lr = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 64
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
gpp = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dmax = float(sys.argv[5]) if len(sys.argv) > 5 else 0.1
dev = torch.device("cuda:0")
sig, xy, col, H, W = synthetic.kernel_inputs(h_lr, w_lr, scale, seed=0, gpp=gpp)
sig, xy, col = sig.to(dev), xy.to(dev), col.to(dev)
grad = synthetic.grad_image(H, W, 1).to(dev)
res = {}
nrun = int(os.environ.get("NRUN", "1"))
for name, flag in (("gaussian", _cabi.FLAG_BWD_GAUSSIAN), ("tile", _cabi.FLAG_BWD_TILE), ("atomic", _cabi.FLAG_BWD_ATOMIC)):
    plan = _cabi.plan(sig, xy, col, H, W, None if dmax < 0 else dmax, flags=flag)     # a fresh workspace per mode
    g = [torch.full_like(t, float("nan")) for t in (sig, xy, col)]
    for _ in range(nrun):
        _cabi.backward(plan, sig, xy, col, grad, *g, overwrite=True)
    torch.cuda.synchronize()
    res[name] = torch.cat([g[1], g[0], g[2]], 1).cpu()   # x y | sx sy rho | r g b
ref = res["gaussian"]
if H * W * sig.shape[0] <= 4e9:
    from oracle import gs_oracle
    o = gs_oracle.backward_f64(sig.cpu().numpy(), xy.cpu().numpy(), col.cpu().numpy(), grad.cpu().numpy(), None if dmax < 0 else dmax)
    ref = torch.cat([torch.from_numpy(o[1]), torch.from_numpy(o[0]), torch.from_numpy(o[2])], 1).float()
    res["oracle"] = ref
    d = (res["gaussian"] - ref).abs()
    print("gaussian vs oracle: max abs diff / column max:", (d.amax(0) / ref.abs().amax(0).clamp_min(1e-20)).tolist())
hx, hy = 0.5 * (W - 1), 0.5 * (H - 1)
for name in ("tile", "atomic"):
    d = (res[name] - ref).abs()
    scale_c = ref.abs().amax(0).clamp_min(1e-20)
    print(f"{name}: max abs diff / column max:", (d.amax(0) / scale_c).tolist())
    rel = (d / (ref.abs() + 1e-3 * scale_c)).amax(1)
    bad = torch.nonzero(rel > 1e-3).flatten()
    print(f"  {bad.numel()} of {ref.shape[0]} Gaussians differ by > 1e-3")
    for i in bad[:12].tolist():
        cx, cy = (xy[i, 0].item() + 1) * hx, (xy[i, 1].item() + 1) * hy
        print(f"   i={i} centre=({cx:.1f},{cy:.1f}) sig_px=({sig[i, 0].item() * hx:.2f},{sig[i, 1].item() * hy:.2f}) rho={sig[i, 2].item():.3f}"
              f" ratio={(res[name][i] / ref[i]).tolist()}")
