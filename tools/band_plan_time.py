"""Plan / forward / backward time of a row band that is handed EVERY Gaussian of the image (the broadcast flow of
gsasr_amd.shard.splat_band), config 4 (8192^2, 1 M Gaussians), for both backward kernels.  One GPU is enough: a
rank's band costs the same whichever rank renders it."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gsasr_amd import _cabi, synthetic

dev = torch.device("cuda:0")
sig, xy, col, H, W = synthetic.kernel_inputs(1024, 1024, 8.0, seed=0)
a, b, c = sig.to(dev), xy.to(dev), col.to(dev)


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows in ((0, H), (0, H // 8), (3 * H // 8, 4 * H // 8)):
    line = f"rows {rows}:"
    for name, flag in (("gaussian", _cabi.FLAG_BWD_GAUSSIAN), ("tile", _cabi.FLAG_BWD_TILE)):
        keep = []
        t_plan = timed(lambda: keep.append(_cabi.plan(a, b, c, H, W, 0.1, rows=rows, flags=flag)), 8)
        p = keep[-1]
        slab = torch.empty(rows[1] - rows[0], W, 3, device=dev)
        t_fwd = timed(lambda: _cabi.forward(p, slab, overwrite=True), 5)
        g = [torch.empty_like(t) for t in (a, b, c)]
        wgt = torch.rand(rows[1] - rows[0], W, 3, device=dev)
        t_bwd = timed(lambda: _cabi.backward(p, a, b, c, wgt, *g, overwrite=True), 5)
        line += f"  [{name}] plan {t_plan:.1f} fwd {t_fwd:.1f} bwd {t_bwd:.1f} us"
        del keep, p
    print(line)
