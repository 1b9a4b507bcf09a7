#!/usr/bin/env python
"""`sample_coords` training step through the HOST API: B samples of config-5 shape, S points each,
forward + backward to the raw decoder parameters -- full render + gather (what the reference does) vs the sampled
kernels, per sample and as one batched canvas; plus the device time of the sampled C entry points alone.

    python tools/sample_time.py [B lr scale gpp dmax S]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import _cabi, gaussian_splatting as gsp, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lr = int(sys.argv[2]) if len(sys.argv) > 2 else 48
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
gpp = int(sys.argv[4]) if len(sys.argv) > 4 else 16
dmax = float(sys.argv[5]) if len(sys.argv) > 5 else 0.5
S = int(sys.argv[6]) if len(sys.argv) > 6 else 2304
dev = torch.device("cuda:0")
p = torch.stack([synthetic.gs_parameters(lr, lr, seed=b, gpp=gpp) for b in range(B)]).to(dev)
H = W = int(lr * scale)
sizes = [(H, W)] * B
sm = [torch.tensor([scale, scale], device=dev)] * B
g = torch.Generator().manual_seed(0)
pts = torch.stack([torch.stack([torch.randint(0, H, (S,), generator=g), torch.randint(0, W, (S,), generator=g)], 1)
                   for _ in range(B)]).to(dev)
wgt = torch.rand(B, 3, S, device=dev)


def run(frac, batched):
    old = gsp.SAMPLED_MAX_FRACTION
    gsp.SAMPLED_MAX_FRACTION = frac
    try:
        pa = p.detach().requires_grad_(True)
        if batched:
            out = gsp.generate_2D_gaussian_splatting_batch(sizes, pa, [scale] * B, sm, dmax=dmax, sample_coords=pts)
            if out.dim() == 4:      # full render: ONE vectorised gather of the whole batch (the reference indexes per point)
                out = out[torch.arange(B, device=dev)[:, None], :, pts[:, :, 0], pts[:, :, 1]].permute(0, 2, 1)
            (out * wgt).sum().backward()
        else:
            loss = 0
            for b in range(B):
                out = gsp.generate_2D_gaussian_splatting_step((H, W), pa[b], scale, sm[b], dmax=dmax, sample_coords=pts[b])
                loss = loss + (out * wgt[b]).sum()
            loss.backward()
    finally:
        gsp.SAMPLED_MAX_FRACTION = old


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


print(f"B={B} N={p.shape[1]} {H}x{W} S={S} ({S / (H * W):.3f} of the pixels) dmax={dmax}")
if B == 1 or os.environ.get("SAMPLE_TIME_HOST", "1") == "0":
    HOST = ()
else:
    HOST = (("per-sample, full render + gather", 0.0, False), ("per-sample, sampled", 1.0, False),
            ("batched canvas, full render + gather", 0.0, True), ("batched canvas, sampled", 1.0, True))
for name, frac, batched in HOST:
    print(f"  {name:38s} {timed(lambda: run(frac, batched)) * 1e3:8.3f} ms per fwd+bwd")

# device time of the C entry points alone (events on the launch stream)
steps = torch.full((B,), 1.2 / scale, device=dev)
if B == 1:
    p, pts, wgt = p[0].contiguous(), pts[0].contiguous(), wgt[0].contiguous()
    _cabi.batch_sample_forward = lambda p_, s_, z_, d_, q_: _cabi.step_sample_forward(p_, s_, H, W, d_, q_)
    _cabi.batch_forward = lambda p_, s_, z_, d_: _cabi.step_forward(p_, s_, H, W, d_)
    _cabi.batch_backward = _cabi.step_backward
out, plan, state = _cabi.batch_sample_forward(p, steps, sizes, dmax, pts)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
n = 50
tf = tb = 0.0
for it in range(n + 5):
    ev[0].record()
    out, plan, state = _cabi.batch_sample_forward(p, steps, sizes, dmax, pts)
    ev[1].record()
    _cabi.step_sample_backward(plan, state, p, steps, wgt)
    ev[2].record()
    torch.cuda.synchronize()
    if it >= 5:
        tf += ev[0].elapsed_time(ev[1])
        tb += ev[1].elapsed_time(ev[2])
print(f"  C entry points, batched sampled: forward {tf / n * 1e3:.1f} us, backward {tb / n * 1e3:.1f} us")
img, plan2 = _cabi.batch_forward(p, steps, sizes, dmax)
gimg = torch.rand(B, plan2.dims.slot, W, 3, device=dev) if B > 1 else torch.rand(H, W, 3, device=dev)
tf = tb = 0.0
for it in range(n + 5):
    ev[0].record()
    img, plan2 = _cabi.batch_forward(p, steps, sizes, dmax)
    ev[1].record()
    _cabi.batch_backward(plan2, p, steps, gimg)
    ev[2].record()
    torch.cuda.synchronize()
    if it >= 5:
        tf += ev[0].elapsed_time(ev[1])
        tb += ev[1].elapsed_time(ev[2])
print(f"  C entry points, batched full image: forward {tf / n * 1e3:.1f} us, backward {tb / n * 1e3:.1f} us")
