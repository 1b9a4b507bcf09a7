#!/usr/bin/env python
"""What PyTorch's own machinery costs around a custom autograd Function on this box (host time per call, GPU idle):
the floor under `GSCUDA.apply(...)` + `.backward(grad)`.  Development aid."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
N, H, W = 65536, 1024, 1024
a = torch.rand(N, 3, device=dev, requires_grad=True)
b = torch.rand(N, 2, device=dev, requires_grad=True)
c = torch.rand(N, 3, device=dev, requires_grad=True)
wgt = torch.rand(H, W, 3, device=dev)


class Null(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c, img):
        ctx.save_for_backward(a, b, c)
        return img

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        a, b, c = ctx.saved_tensors
        return torch.empty_like(a), torch.empty_like(b), torch.empty_like(c), None


def t(fn, n=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt


img0 = torch.zeros(H, W, 3, device=dev)


def null_step():
    a.grad = b.grad = c.grad = None
    Null.apply(a, b, c, img0).backward(wgt)


def null_fwd():
    Null.apply(a, b, c, img0)


sys.stdout.reconfigure(line_buffering=True)
print(f"torch.zeros(H,W,3)                      {t(lambda: torch.zeros(H, W, 3, device=dev)):7.1f} us")
print(f"3 x torch.empty_like                    {t(lambda: (torch.empty_like(a), torch.empty_like(b), torch.empty_like(c))):7.1f} us")
print(f"null Function.apply                     {t(null_fwd):7.1f} us")
print(f"null Function.apply + .backward(grad)   {t(null_step):7.1f} us")
from gsasr_amd import _cabi  # noqa: E402
from gsasr_amd.gs_cuda_dmax.gswrapper import GSCUDA  # noqa: E402
from gsasr_amd import synthetic  # noqa: E402
sig, xy, col, H, W = synthetic.kernel_inputs(256, 256, 4.0, seed=0, device="cpu")
sa, sb, sc = (x.to(dev) for x in (sig, xy, col))
img = torch.zeros(H, W, 3, device=dev)
g = [torch.empty_like(x) for x in (sa, sb, sc)]
plan = [None]


def cabi_fwd():
    plan[0] = _cabi.plan_forward(sa, sb, sc, img, 0.1)


def cabi_bwd():
    _cabi.backward(plan[0], sa, sb, sc, wgt, *g, overwrite=True)


# (host enqueue time only: the GPU falls behind and is drained outside the timed loop; keep n small enough for the queue)
print(f"_cabi.plan_forward (2 C calls, 3 launches) {t(cabi_fwd, 100):7.1f} us host")
print(f"_cabi.backward (1 C call, 1 launch)        {t(cabi_bwd, 100):7.1f} us host")
L = _cabi.lib()
import ctypes  # noqa: E402
d = _cabi.make_dims(sa.shape[0], H, W, 0.1)        # (flags 0: the plan zeroes its own counters)
ws = torch.empty(L.gsasr_splat_workspace_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
st = _cabi._stream(dev)
print(f"C gsasr_splat_plan alone                   {t(lambda: L.gsasr_splat_plan(sa.data_ptr(), sb.data_ptr(), sc.data_ptr(), ctypes.byref(d), ws.data_ptr(), ws.numel(), st), 100):7.1f} us host")
print(f"C gsasr_splat_forward alone                {t(lambda: L.gsasr_splat_forward(ctypes.byref(d), ws.data_ptr(), ws.numel(), img.data_ptr(), st), 100):7.1f} us host")

# ---- phases of the drop-in step, host time each (perf_counter, no synchronisation inside the loop) ----
a2, b2, c2 = (x.clone().requires_grad_(True) for x in (sa, sb, sc))
acc = [0.0, 0.0, 0.0]


def phases():
    a2.grad = b2.grad = c2.grad = None
    t0 = time.perf_counter()
    z = torch.zeros(H, W, 3, device=dev)
    t1 = time.perf_counter()
    out = GSCUDA.apply(a2, b2, c2, z, 0.1)
    t2 = time.perf_counter()
    out.backward(wgt)
    t3 = time.perf_counter()
    acc[0] += t1 - t0
    acc[1] += t2 - t1
    acc[2] += t3 - t2


for _ in range(30):
    phases()
torch.cuda.synchronize()
acc[:] = [0.0, 0.0, 0.0]
n = 300
t0 = time.perf_counter()
for _ in range(n):
    phases()
wall = (time.perf_counter() - t0) / n * 1e6
torch.cuda.synchronize()
print(f"drop-in step: zeros {acc[0] / n * 1e6:.1f} us, GSCUDA.apply {acc[1] / n * 1e6:.1f} us, .backward {acc[2] / n * 1e6:.1f} us, loop wall {wall:.1f} us")
# the same with the GPU drained every step (no queue back-pressure on the host calls)
acc[:] = [0.0, 0.0, 0.0]
for _ in range(n):
    phases()
    torch.cuda.synchronize()
print(f"drained each step: zeros {acc[0] / n * 1e6:.1f} us, GSCUDA.apply {acc[1] / n * 1e6:.1f} us, .backward {acc[2] / n * 1e6:.1f} us")
