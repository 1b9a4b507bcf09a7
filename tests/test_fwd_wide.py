"""The WIDE forward (k_render_fwd16: 16 x 16 sub-tiles, four pixels per lane, the library's choice from x5 up on single
images of 2 Mpx and more) against the oracle and against the 8 x 16 kernels, forced by GSASR_FLAG_FWD_WIDE /
GSASR_FLAG_FWD_NARROW on shapes of every raggedness: the same sums in a different order.

Reference semantics: utils/gs_cuda_dmax/gs.cu:24-60 (bounded), utils/gs_cuda/gs.cu:24-50 (unbounded)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
IMG_ATOL = 1e-4       # the parity bar of every forward test (sums of O(1) terms; north_star's 1e-4)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RASTER = sorted(glob.glob(os.path.join(GOLDEN, "raster_*.npz")))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run with -m gpu on the MI355X box)"
    return torch.device("cuda:0")


def _both(sig, xy, col, H, W, dmax, dev, rows=None, chw=False, cutoff=0.0):
    """the image of one plan through the wide and through the narrow kernel"""
    from gsasr_amd import _cabi
    a, b, c = (torch.as_tensor(t).to(dev) for t in (sig, xy, col))
    plan = _cabi.plan(a, b, c, H, W, dmax, rows=rows, cutoff=cutoff, flags=_cabi.FLAG_FORWARD_ONLY)
    nrows = H if rows is None else rows[1] - rows[0]
    out = []
    for flag in (_cabi.FLAG_FWD_WIDE, _cabi.FLAG_FWD_NARROW):
        img = torch.full((3, nrows, W) if chw else (nrows, W, 3), float("nan"), device=dev)
        _cabi.forward(plan, img, overwrite=True, chw=chw, flags=flag)
        out.append(img.permute(1, 2, 0).contiguous() if chw else img)
    return out


@pytest.mark.parametrize("path", RASTER, ids=[os.path.basename(p)[7:-4] for p in RASTER])
def test_golden_vectors_through_the_wide_forward(path, dev):
    """every reference-captured raster fixture (tests/golden/make_golden.py: check.py::torch_version)"""
    z = np.load(path)
    dmax = None if float(z["dmax"]) < 0 else float(z["dmax"])
    wide, narrow = _both(z["sigmas"], z["coords"], z["colors"], int(z["h"]), int(z["w"]), dmax, dev)
    assert np.abs(wide.cpu().numpy() - z["img_f64"]).max() <= IMG_ATOL
    assert float((wide - narrow).abs().max()) <= 2e-6 * max(1.0, float(narrow.abs().max()))


@pytest.mark.parametrize("case", [(37, 29, 3.0, 1), (24, 40, 2.5, 1), (12, 12, 4.0, 16), (20, 16, 12.0, 1), (31, 17, 6.5, 2),
                                  (9, 50, 8.0, 1), (50, 9, 8.0, 1)], ids=lambda c: "lr%dx%d_s%g_gpp%d" % c)
@pytest.mark.parametrize("dmax", [None, 0.25, 0.05], ids=["unbounded", "dmax0.25", "dmax0.05"])
def test_ragged_sizes_against_oracle(case, dmax, dev):
    """H / W not multiples of the 16-px sub-tile or the 32-px tile, fractional scales, 16 Gaussians per LR pixel, dmax boxes
    that cut most windows (the exact in-kernel test): wide forward vs the f64 oracle, and vs the narrow kernels"""
    from gsasr_amd import synthetic
    from oracle import gs_oracle
    h_lr, w_lr, scale, gpp = case
    sig, xy, col, H, W = synthetic.kernel_inputs(h_lr, w_lr, scale, seed=20, gpp=gpp)
    ref = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, dmax)
    wide, narrow = _both(sig, xy, col, H, W, dmax, dev)
    tol = IMG_ATOL * max(1.0, float(np.abs(ref).max()))
    assert np.abs(wide.cpu().numpy() - ref).max() <= tol
    assert float((wide - narrow).abs().max()) <= 2e-6 * max(1.0, float(narrow.abs().max()))


@pytest.mark.parametrize("rows", [(0, 40), (23, 151), (100, 117), (130, 200)])
def test_row_bands_and_planar_output(rows, dev):
    """row bands whose first row is no multiple of anything (the tile grid starts at row0), planar CHW output, and the
    accumulate contract (img += splat, gs.cu:52-58)"""
    from gsasr_amd import _cabi, synthetic
    from oracle import gs_oracle
    sig, xy, col, H, W = synthetic.kernel_inputs(25, 31, 8.0, seed=3)
    assert (H, W) == (200, 248)
    ref = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, 0.2, rows=rows)
    for chw in (False, True):
        wide, narrow = _both(sig, xy, col, H, W, 0.2, dev, rows=rows, chw=chw)
        assert np.abs(wide.cpu().numpy() - ref).max() <= IMG_ATOL * max(1.0, float(np.abs(ref).max()))
        assert float((wide - narrow).abs().max()) <= 2e-6 * max(1.0, float(narrow.abs().max()))
    a, b, c = sig.to(dev), xy.to(dev), col.to(dev)
    plan = _cabi.plan(a, b, c, H, W, 0.2, rows=rows, flags=_cabi.FLAG_FORWARD_ONLY)
    base = torch.rand(rows[1] - rows[0], W, 3, device=dev)
    img = base.clone()
    _cabi.forward(plan, img, flags=_cabi.FLAG_FWD_WIDE)                     # accumulates
    assert np.abs((img - base).cpu().numpy() - ref).max() <= IMG_ATOL * max(1.0, float(np.abs(ref).max()))


def test_library_picks_the_wide_forward_from_x5_and_2mpx(dev):
    """x8 at 2048^2 (64 HR pixels per Gaussian, 32 768 sub-tiles of 8 x 16): the default is the wide kernel
    (gsasr_forward_subtile_width) and a band of it agrees with the oracle; x4 at the same size keeps the narrow kernel"""
    from gsasr_amd import _cabi, synthetic
    from oracle import gs_oracle
    for scale, lr, is_wide in ((8.0, 256, True), (4.0, 512, False)):
        sig, xy, col, H, W = synthetic.kernel_inputs(lr, lr, scale, seed=1)
        assert (H, W) == (2048, 2048)
        a, b, c = sig.to(dev), xy.to(dev), col.to(dev)
        plan = _cabi.plan(a, b, c, H, W, 0.1, flags=_cabi.FLAG_FORWARD_ONLY)
        assert _cabi.forward_subtile_width(plan) == (16 if is_wide else 8)
        assert _cabi.forward_subtile_width(plan, _cabi.FLAG_FWD_WIDE) == 16 and _cabi.forward_subtile_width(plan, _cabi.FLAG_FWD_NARROW) == 8
        imgs = {}
        for name, flag in (("default", 0), ("wide", _cabi.FLAG_FWD_WIDE), ("narrow", _cabi.FLAG_FWD_NARROW)):
            imgs[name] = torch.empty(H, W, 3, device=dev)
            _cabi.forward(plan, imgs[name], overwrite=True, flags=flag)
        assert float((imgs["wide"] - imgs["narrow"]).abs().max()) <= 2e-6 * max(1.0, float(imgs["narrow"].abs().max()))
        rows = (1000, 1040)
        ref = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, 0.1, rows=rows)
        assert np.abs(imgs["default"][rows[0]:rows[1]].cpu().numpy() - ref).max() <= IMG_ATOL * max(1.0, float(np.abs(ref).max()))


def test_explicit_cutoffs_and_a_window_wider_than_255_columns(dev):
    """tau = 104 (every non-zero fp32 term) and no cull at all through the wide kernel; and the window k_bin computes no
    column spans for (wider than 255 units of 8 px): its far sub-tiles must not lose it"""
    from gsasr_amd import synthetic
    from oracle import gs_oracle
    sig, xy, col, H, W = synthetic.kernel_inputs(16, 20, 6.0, seed=8)
    ref = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, 0.3)
    for cutoff in (104.0, -1.0):
        wide, narrow = _both(sig, xy, col, H, W, 0.3, dev, cutoff=cutoff)
        assert np.abs(wide.cpu().numpy() - ref).max() <= IMG_ATOL * max(1.0, float(np.abs(ref).max()))
    H, W = 24, 2600
    sig = torch.tensor([[0.9, 0.05, 0.3], [0.02, 0.4, -0.5]], dtype=torch.float32)
    xy = torch.tensor([[0.0, 0.0], [0.5, 0.2]], dtype=torch.float32)
    col = torch.tensor([[0.5, 0.25, 1.0], [1.0, 0.5, 0.1]], dtype=torch.float32)
    ref = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, None)
    wide, narrow = _both(sig, xy, col, H, W, None, dev)
    assert np.abs(wide.cpu().numpy() - ref).max() <= IMG_ATOL
    assert np.abs(narrow.cpu().numpy() - ref).max() <= IMG_ATOL


def test_host_api_at_x8_runs_the_wide_forward_and_matches_the_oracle(dev):
    """`generate_2D_gaussian_splatting_step` (utils/gaussian_splatting.py:158-217) at x8 on 2048^2: the fused step path (planar
    CHW image, prologue fused into the plan) takes the wide forward; a band of the result against oracle(prologue(raw
    parameters)), and the gradient of that band to the raw parameters against the oracle's analytic backward"""
    from gsasr_amd import _cabi, gaussian_splatting as gsp, synthetic
    from oracle import gs_oracle, host_ref
    p = synthetic.gs_parameters(256, 256, seed=2)
    H = W = 2048
    sm = torch.tensor([8.0, 8.0])
    d = _cabi.make_dims(p.shape[0], H, W, 0.1)
    assert int(_cabi.lib().gsasr_forward_subtile_width(__import__("ctypes").byref(d))) == 16
    pg = p.clone().to(dev).requires_grad_(True)
    out = gsp.generate_2D_gaussian_splatting_step((H, W), pg, 8.0, sm.to(dev), if_dmax=True, dmax_mode="fix", dmax=0.1)
    assert out.shape == (3, H, W)
    sig, xy, col, dmax = host_ref.prologue(p, (H, W), sm, dmax=0.1, dmax_mode="fix")
    rows = (1016, 1048)
    ref = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, dmax, rows=rows)
    got = out.detach()[:, rows[0]:rows[1]].permute(1, 2, 0).cpu().numpy()
    assert np.abs(got - ref).max() <= IMG_ATOL * max(1.0, float(np.abs(ref).max()))
    wgt = synthetic.grad_image(rows[1] - rows[0], W, 4)
    (out[:, rows[0]:rows[1]] * wgt.permute(2, 0, 1).to(dev)).sum().backward()
    pr = p.clone().double().requires_grad_(True)
    s2, x2, c2, _ = host_ref.prologue(pr, (H, W), sm.double(), dmax=0.1, dmax_mode="fix")
    g = gs_oracle.backward_f64(sig.numpy(), xy.numpy(), col.numpy(), wgt.numpy(), dmax, h=H, rows=rows)
    torch.autograd.backward([s2, x2, c2], [torch.from_numpy(a) for a in g])
    want, have = pr.grad.numpy(), pg.grad.cpu().numpy()
    assert np.abs(have - want).max() <= 2e-4 * np.abs(want).max()
