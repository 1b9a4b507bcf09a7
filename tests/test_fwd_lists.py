"""Tile lists (round 5; gsasr_dims.list_cap, k_bin's tl_emit, k_render_fwd_list / k_render_fwd16_list): the forward rendered from
the plan's per-tile hit lists against the search kernels of rounds 1-4 (same sums in another order) and against the oracle.
Reference semantics: utils/gs_cuda_dmax/gs.cu:24-60 (the only cull of the reference is its box test, :38-50)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run with -m gpu on the MI355X box)"
    return torch.device("cuda:0")


@pytest.mark.parametrize("gpp,scale", [(1, 4.0), (16, 4.0), (1, 8.0)], ids=["x4", "x4-16-per-LR-px", "x8"])
@pytest.mark.parametrize("dmax", [0.1, None], ids=["bounded", "unbounded"])
def test_lists_equal_search_and_oracle(gpp, scale, dmax, dev):
    """the same image from: search kernels (list_cap -1), lists of the library's capacity, lists so small that most tiles
    overflow (rendered by the search instead) -- through the 8 x 16 and the wide kernels; rows against the oracle"""
    from gsasr_amd import _cabi, synthetic
    from oracle import gs_oracle
    h_lr, w_lr = (40, 52) if gpp > 1 else (96, 130)
    sig, xy, col, H, W = synthetic.kernel_inputs(h_lr, w_lr, scale, seed=31, gpp=gpp)
    a, b, c = sig.to(dev), xy.to(dev), col.to(dev)
    imgs = {}
    for wide in (_cabi.FLAG_FWD_NARROW, _cabi.FLAG_FWD_WIDE):
        for cap in (-1, 512 if gpp == 1 else 4096, 64):
            plan = _cabi.plan(a, b, c, H, W, dmax, flags=_cabi.FLAG_FORWARD_ONLY | wide, list_cap=cap)
            img = torch.full((H, W, 3), float("nan"), device=dev)
            _cabi.forward(plan, img, overwrite=True, flags=wide)
            imgs[(wide, cap)] = img
    ref = imgs[(_cabi.FLAG_FWD_NARROW, -1)]
    assert bool(torch.isfinite(ref).all())
    for k, v in imgs.items():
        assert float((v - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), k
    rows = (H // 2 - 3, H // 2 + 5)
    want = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, dmax, rows=rows)
    for k, v in imgs.items():
        assert np.abs(v[rows[0]:rows[1]].cpu().numpy() - want).max() <= 1e-4, k


def test_lists_on_a_row_band_and_accumulate(dev):
    """a row band (the shard's form) with lists: accumulate-into contract, band offsets of the tiles"""
    from gsasr_amd import _cabi, synthetic
    from oracle import gs_oracle
    sig, xy, col, H, W = synthetic.kernel_inputs(64, 48, 4.0, seed=33, gpp=4)
    a, b, c = sig.to(dev), xy.to(dev), col.to(dev)
    rows = (37, 203)
    plan = _cabi.plan(a, b, c, H, W, 0.2, rows=rows, list_cap=1024)
    base = torch.rand(rows[1] - rows[0], W, 3, device=dev)
    img = base.clone()
    _cabi.forward(plan, img, overwrite=False)
    want = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, 0.2, rows=rows)
    assert np.abs((img - base).cpu().numpy() - want).max() <= 1e-4


def test_default_lists_only_for_dense_plans(dev):
    """0 = the library decides: lists for >= 1 Gaussian per 4 pixels (workspace grows by the list region), none below"""
    import ctypes
    from gsasr_amd import _cabi
    L = _cabi.lib()
    size = lambda s, cap: L.gsasr_splat_workspace_bytes(ctypes.byref(_cabi.make_dims(s, 1024, 1024, 0.1, list_cap=cap)))
    assert size(65536, 0) == size(65536, -1) and size(1048576, 0) > size(1048576, -1)


def test_fuzz_lists():
    """tools/fuzz_lists.py, 40 cases: random images (small, GSASR-sized, thin), row bands, GSASR-shaped Gaussians mixed with
    hairlines, large and off-image ones; search = lists = overflowing lists, narrow and wide, + rows against the oracle"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_lists.py"), "40", "17"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "cases ok" in r.stdout


@pytest.mark.parametrize("scale,wide", [(4.0, 0), (8.0, 1)], ids=["x4-32x16-tiles", "x8-32x32-tiles"])
@pytest.mark.parametrize("cap", [2048, 64], ids=["lists", "overflowing-lists"])
def test_tile_backward_from_lists_against_oracle(scale, wide, cap, dev):
    """the tile-stationary backward reading the plan's tile lists instead of walking the cells around each tile (entries carry the
    quadrant mask and the Gaussian's slot; k_bin zeroes the slots of tiles its ellipse misses): gradients against the oracle and
    against the same kernel without lists; slots and atomics; a capacity so small that most tiles fall back to the walk"""
    from gsasr_amd import _cabi, synthetic
    from oracle import gs_oracle
    from test_bwd_tile import per_gaussian_ok
    sig, xy, col, H, W = synthetic.kernel_inputs(48, 40, scale, seed=41)
    a, b, c = sig.to(dev), xy.to(dev), col.to(dev)
    wgt = synthetic.grad_image(H, W, 42)
    want = gs_oracle.backward_f64(sig.numpy(), xy.numpy(), col.numpy(), wgt.numpy(), 0.2)
    fw = _cabi.FLAG_FWD_WIDE if wide else _cabi.FLAG_FWD_NARROW
    got = {}
    for name, flags, lc in (("lists", _cabi.FLAG_BWD_TILE, cap), ("search", _cabi.FLAG_BWD_TILE, -1),
                            ("lists-atomic", _cabi.FLAG_BWD_TILE | _cabi.FLAG_BWD_ATOMIC, cap)):
        plan = _cabi.plan(a, b, c, H, W, 0.2, flags=flags | fw, list_cap=lc)
        g = [torch.full_like(t, float("nan")) for t in (a, b, c)]
        _cabi.backward(plan, a, b, c, wgt.to(dev), *g, overwrite=True)
        got[name] = [t.cpu().numpy() for t in g]
        for arr, w_, tn in zip(got[name], want, ("sigmas", "coords", "colors")):
            per_gaussian_ok(arr, w_, tn, rho=sig.numpy()[:, 2])
    for x, y in zip(got["lists"], got["search"]):
        assert np.abs(x - y).max() <= 2e-5 * max(1e-30, np.abs(y).max())
