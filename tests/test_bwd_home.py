"""GPU parity of the HOME-TILE backward (k_render_bwd_home, GSASR_FLAG_BWD_HOME; round 6) against the CPU oracle, on the
inputs the other two backward kernels are held to (tests/test_bwd_tile.py, tests/test_hip_parity.py).  Reference semantics:
utils/gs_cuda_dmax/gs.cu:85-165, utils/gs_cuda/gs.cu:112-176.

The kernel has two paths per Gaussian -- items evaluated from the staged region, and a whole-wave sweep for Gaussians whose
window does not fit it -- and three tile shapes (GSASR_SPLAT_HOME_VARIANT is a development switch; the shapes are also chosen
by density: 1, 4 and 16 Gaussians per LR pixel reach all three).  Scales from x2 to x24 move the share of swept Gaussians
from none to all.
"""
import os

import numpy as np
import pytest
import torch

from test_bwd_tile import GRAD_RTOL, RASTER, _backward, _check, _synth, _t, per_gaussian_ok

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run with -m gpu on the MI355X box)"
    return torch.device("cuda:0")


def _home():
    from gsasr_amd import _cabi
    return _cabi.FLAG_BWD_HOME


@pytest.mark.parametrize("path", RASTER, ids=[os.path.basename(p)[7:-4] for p in RASTER])
def test_golden_backward_home(path, dev):
    z = np.load(path)
    dmax = None if float(z["dmax"]) < 0 else float(z["dmax"])
    got = _check(z["sigmas"], z["coords"], z["colors"], z["weight"], int(z["h"]), int(z["w"]), dmax, dev, _home())
    for g, key in zip(got, ("g_sigmas", "g_coords", "g_colors")):   # the reference's own fp64-input run
        assert np.abs(g - z[key + "_f64"]).max() <= GRAD_RTOL * np.abs(z[key + "_f64"]).max(), key


@pytest.mark.parametrize("cutoff", [0.0, 32.0, 104.0, -1.0], ids=["adaptive", "tau32", "tau104", "nocut"])
@pytest.mark.parametrize("dmax", [None, 0.5, 0.1], ids=["unbounded", "dmax0.5", "dmax0.1"])
def test_synthetic_x4_256_home(dmax, cutoff, dev):
    sig, xy, col, H, W, wgt = _synth(64, 64, 4.0, seed=10)
    _check(sig, xy, col, wgt, H, W, dmax, dev, _home(), cutoff=cutoff)


@pytest.mark.parametrize("case", [(37, 29, 3.0, 1), (24, 40, 2.5, 1), (12, 12, 4.0, 16), (20, 16, 12.0, 1), (31, 17, 6.5, 2),
                                  (48, 48, 4.0, 16), (12, 15, 16.0, 1), (9, 11, 24.0, 1), (40, 44, 2.0, 4), (33, 35, 4.0, 4)],
                         ids=lambda c: "lr%dx%d_s%g_gpp%d" % c)
def test_ragged_sizes_scales_and_densities_home(case, dev):
    """non-square, H/W not multiples of the tile, fractional scales; x6.5 and up: most windows do not fit the region (the sweep
    path); 1 / 4 / 16 Gaussians per LR pixel: the three tile shapes; (48, 48, 4, 16) is BASELINE config 5's sample"""
    h_lr, w_lr, scale, gpp = case
    sig, xy, col, H, W, wgt = _synth(h_lr, w_lr, scale, seed=20, gpp=gpp)
    for dmax in (None, 0.25):
        _check(sig, xy, col, wgt, H, W, dmax, dev, _home())


def test_deterministic_and_accumulate_contract_home(dev):
    """one write per Gaussian from sums added in a fixed order: run to run bit-identical; without OVERWRITE_GRADS the
    gradient is ADDED to what the caller left in the outputs (the reference's dmax backward, gs.cu:139-146)"""
    sig, xy, col, H, W, wgt = _synth(56, 56, 4.0, seed=50)
    a = _backward(sig, xy, col, wgt, H, W, 0.1, dev, _home())
    b = _backward(sig, xy, col, wgt, H, W, 0.1, dev, _home())
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    c = _backward(sig, xy, col, wgt, H, W, 0.1, dev, _home(), accumulate=True)
    for x, y in zip(a, c):
        assert np.abs(x - y).max() <= 1e-6 * max(1.0, np.abs(x).max())


def test_row_band_home(dev):
    """row bands that start and end inside cells and tiles: the staged region is clipped to the band"""
    sig, xy, col, H, W, wgt = _synth(48, 40, 4.0, seed=60)
    for rows in ((0, 64), (50, 131), (H - 37, H), (77, 78)):
        _check(sig, xy, col, wgt, H, W, 0.2, dev, _home(), rows=rows)


def test_large_class_and_degenerate_gaussians_home(dev):
    """check.py-style sigma ~ U(0,1): every Gaussian spans the image (the large class: row chunks over all waves), plus
    off-image, NaN and needle Gaussians"""
    from oracle import gs_oracle
    rng = np.random.default_rng(7)
    s, h, w = 60, 70, 90
    sig = np.concatenate([0.999 * rng.random((s, 2)), 1.8 * rng.random((s, 1)) - 0.9], 1).astype(np.float32)
    xy = (2 * rng.random((s, 2)) - 1).astype(np.float32)
    col = rng.random((s, 3)).astype(np.float32)
    wgt = rng.random((h, w, 3)).astype(np.float32)
    sig[5, :2] = (1e-5, 0.5)
    sig[6, :2] = (0.4, 2e-6)
    xy[7] = (3.0, 0.2)
    xy[8] = (np.nan, 0.0)
    sig[9, 2] = 0.9995
    for dmax in (None, 0.6):
        got = _backward(sig, xy, col, wgt, h, w, dmax, dev, _home())
        keep = np.ones(s, bool)
        keep[8] = False   # a non-finite Gaussian is dropped (documented deviation: the reference propagates NaN)
        want = gs_oracle.backward_f64(sig[keep], xy[keep], col[keep], wgt, dmax)
        for g, r, name in zip(got, want, ("sigmas", "coords", "colors")):
            assert np.all(g[8] == 0.0)
            per_gaussian_ok(g[keep], r, name, rho=sig[keep, 2])


def test_mixed_sizes_in_one_tile_home(dev):
    """Gaussians of every size binned in the same cells: sub-pixel needles, x4-sized ones (items), windows of 40-120 px
    (swept) and image-spanning ones (large class), saturated correlations among them"""
    from oracle import gs_oracle
    rng = np.random.default_rng(11)
    s, h, w = 900, 150, 170
    sig = np.stack([10 ** rng.uniform(-3.0, -0.3, s), 10 ** rng.uniform(-3.0, -0.3, s),
                    np.clip(rng.normal(0, 0.7, s), -0.99999, 0.99999)], 1).astype(np.float32)
    xy = rng.uniform(-1.05, 1.05, (s, 2)).astype(np.float32)
    col = rng.uniform(0, 1, (s, 3)).astype(np.float32)
    wgt = rng.uniform(-1, 1, (h, w, 3)).astype(np.float32)
    for dmax in (None, 0.35):
        for cutoff in (0.0, -1.0):
            got = _backward(sig, xy, col, wgt, h, w, dmax, dev, _home(), cutoff=cutoff)
            want = gs_oracle.backward_f64(sig, xy, col, wgt, dmax)
            for g, r, name in zip(got, want, ("sigmas", "coords", "colors")):
                per_gaussian_ok(g, r, name, rho=sig[:, 2])


def test_config2_full_size_home_backward_against_oracle(dev):
    """BASELINE config 2 in full (1024^2, 65 536 Gaussians, dmax 0.1): gradient of a 96-row band against the oracle (all
    Gaussians, band rows only), the whole image against the Gaussian-stationary kernel; the same at 16 per LR pixel on 512^2"""
    from gsasr_amd import _cabi, synthetic
    from oracle import gs_oracle
    sig, xy, col, H, W = synthetic.kernel_inputs(256, 256, 4.0, seed=0)
    wgt = synthetic.grad_image(H, W, 1)
    s, c, k, g = sig.numpy(), xy.numpy(), col.numpy(), wgt.numpy()
    rows = (448, 544)
    got = _backward(s, c, k, g, H, W, 0.1, dev, _home(), rows=rows)
    want = gs_oracle.backward_f64(s, c, k, g[rows[0]:rows[1]], 0.1, h=H, rows=rows)
    for a, b, name in zip(got, want, ("sigmas", "coords", "colors")):
        per_gaussian_ok(a, b, name, rho=s[:, 2])
    full = _backward(s, c, k, g, H, W, 0.1, dev, _home())
    ref = _backward(s, c, k, g, H, W, 0.1, dev, _cabi.FLAG_BWD_GAUSSIAN)
    for a, b, name in zip(full, ref, ("sigmas", "coords", "colors")):
        per_gaussian_ok(a, b, name, rho=s[:, 2])
    sig, xy, col, H, W = synthetic.kernel_inputs(128, 128, 4.0, seed=3, gpp=16)
    wgt = synthetic.grad_image(H, W, 4)
    s, c, k, g = sig.numpy(), xy.numpy(), col.numpy(), wgt.numpy()
    full = _backward(s, c, k, g, H, W, 0.1, dev, _home())
    ref = _backward(s, c, k, g, H, W, 0.1, dev, _cabi.FLAG_BWD_GAUSSIAN)
    for a, b, name in zip(full, ref, ("sigmas", "coords", "colors")):
        per_gaussian_ok(a, b, name, rho=s[:, 2])


@pytest.mark.parametrize("seed", range(int(os.environ.get("GSASR_FUZZ_SEEDS", "12"))))   # more seeds: set the variable
def test_fuzz_extreme_parameters_home_backward(seed, dev):
    """the fuzz of test_hip_parity.py::test_fuzz_extreme_parameters for the home-tile backward: random sizes and parameter
    ranges far outside what the decoder emits -- sigma over five decades, |rho| up to 0.9995, centres far off the image, dmax
    from sub-pixel to larger than the image, all three cutoff modes"""
    from oracle import gs_oracle
    rng = np.random.default_rng(1000 + seed)
    h, w = int(rng.integers(2, 90)), int(rng.integers(2, 90))
    s = int(rng.integers(1, 400))
    sig = np.stack([10 ** rng.uniform(-4, 0.7, s), 10 ** rng.uniform(-4, 0.7, s),
                    np.clip(rng.normal(0, 0.6, s), -0.9995, 0.9995)], 1).astype(np.float32)
    xy = rng.uniform(-1.6, 1.6, (s, 2)).astype(np.float32)
    col = rng.uniform(0, 1, (s, 3)).astype(np.float32)
    wgt = rng.uniform(-1, 1, (h, w, 3)).astype(np.float32)
    dmax = [None, float(10 ** rng.uniform(-2.5, 0.5))][seed % 2]
    cutoff = [0.0, 104.0, -1.0][seed % 3]
    grads = _backward(sig, xy, col, wgt, h, w, dmax, dev, _home(), cutoff=cutoff)
    gref = gs_oracle.backward_f64(sig, xy, col, wgt, dmax)
    g32 = gs_oracle.backward_f32(sig, xy, col, wgt, dmax, use_fma=True)   # the reference's own fp32 arithmetic
    for got, want, r32, name in zip(grads, gref, g32, ("sigmas", "coords", "colors")):
        assert np.isfinite(got).all(), name
        tol = (5e-4 * np.abs(want).max(axis=1, keepdims=True) + 2.0 * np.abs(r32 - want) + 1e-5 * np.abs(want).max() + 1e-6)
        err = np.abs(got - want)
        assert err.max() <= GRAD_RTOL * np.abs(want).max() + 1e-30, name
        well = (1.0 - sig[:, 2].astype(np.float64) ** 2) >= 0.02
        bad = (err > tol) & well[:, None]
        assert not bad.any(), (name, int(np.argwhere(bad)[0][0]), float(err[bad].max()), float(np.abs(want).max()),
                               sig[np.argwhere(bad)[0][0]].tolist())


def test_packed_records_home_backward(dev):
    """GSASR_FLAG_STRIDE8 (the wire format of the multi-GPU exchange) through the home-tile backward: inputs and gradients
    as columns of one [N,8] array, row band, dead (NaN) padding records whose zero gradient the kernel's tail writes"""
    import ctypes

    from gsasr_amd import _cabi, shard, synthetic
    from oracle import gs_oracle
    sig, xy, col, H, W = synthetic.kernel_inputs(40, 36, 4.0, seed=95)
    wgt = synthetic.grad_image(H, W, 96)
    rec = shard.pack(sig, xy, col)
    pad = torch.full((37, 8), float("nan"))
    rec_dev = torch.cat([rec, pad]).to(dev)
    rows = (40, 120)
    d = _cabi.make_dims(rec_dev.shape[0], H, W, 0.25, rows, 0.0, _cabi.FLAG_STRIDE8 | _cabi.FLAG_BWD_HOME | _cabi.FLAG_OVERWRITE_GRADS)
    L = _cabi.lib()
    nbytes = L.gsasr_splat_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    base = rec_dev.data_ptr()
    _cabi.check(L.gsasr_splat_plan(base, base + 12, base + 20, ctypes.byref(d), ws.data_ptr(), nbytes, _cabi._stream(dev)), "plan")
    g = torch.full_like(rec_dev, float("nan"))
    gw = wgt[rows[0]:rows[1]].contiguous().to(dev)
    gb = g.data_ptr()
    _cabi.check(L.gsasr_splat_backward(base, base + 12, base + 20, gw.data_ptr(), gb, gb + 12, gb + 20, ctypes.byref(d), ws.data_ptr(),
                                       nbytes, _cabi._stream(dev)), "backward")
    torch.cuda.synchronize()
    got = g.cpu().numpy()
    assert np.all(got[rec.shape[0]:] == 0.0)                 # dead padding: zero gradient, not NaN
    want = gs_oracle.backward_f64(sig.numpy(), xy.numpy(), col.numpy(), wgt[rows[0]:rows[1]].numpy(), 0.25, h=H, rows=rows)
    n = rec.shape[0]
    for cols, w_, name in ((slice(0, 3), want[0], "sigmas"), (slice(3, 5), want[1], "coords"), (slice(5, 8), want[2], "colors")):
        per_gaussian_ok(got[:n, cols], w_, name, rho=sig.numpy()[:, 2])


def test_fused_step_home_backward_through_host_api(dev):
    """generate_2D_gaussian_splatting_step with BACKWARD_KERNEL = 'home': the planar gradient autograd hands back is interleaved
    inside the C call and the home-tile kernel sweeps it; against the oracle composed with the torch prologue"""
    from gsasr_amd import gaussian_splatting as gsp, synthetic
    from oracle import gs_oracle
    h_lr, w_lr, scale = 30, 26, 4.0
    H, W = int(h_lr * scale), int(w_lr * scale)
    p = synthetic.gs_parameters(h_lr, w_lr, seed=70)
    wgt = synthetic.grad_image(H, W, 71)
    old = gsp.BACKWARD_KERNEL
    res = {}
    try:
        for mode in ("gaussian", "home"):
            gsp.BACKWARD_KERNEL = mode
            pa = p.to(dev).requires_grad_(True)
            out = gsp.generate_2D_gaussian_splatting_step((H, W), pa, scale, torch.tensor([scale, scale]), dmax=0.3)
            (out * wgt.to(dev).permute(2, 0, 1)).sum().backward()
            res[mode] = (out.detach().cpu().numpy(), pa.grad.cpu().numpy())
    finally:
        gsp.BACKWARD_KERNEL = old
    pc = p.clone().requires_grad_(True)
    sx, sy, rho, xy, col = gsp._activate(pc)
    sig_k, xy_k, col_k, _, _ = gsp._to_kernel_frame(sx, sy, rho, xy, col, (H, W), 1.2 / scale)
    g = gs_oracle.backward_f64(sig_k.detach().numpy(), xy_k.detach().numpy(), col_k.detach().numpy(), wgt.numpy(), 0.3)
    torch.autograd.backward([sig_k, xy_k, col_k], [torch.from_numpy(x).float() for x in g])
    want = pc.grad.numpy()
    for mode in ("gaussian", "home"):
        per_gaussian_ok(res[mode][1], want, "gs_parameters/" + mode)


@pytest.mark.parametrize("gpp", [2, 16], ids=["gpp2", "gpp16"])
def test_batched_canvas_home_backward(gpp, dev):
    """a ragged batch through generate_2D_gaussian_splatting_batch with the home-tile backward: tiles are counted per slot (a
    32-row tile never straddles two samples; slots here are 64 rows with sample heights 40 / 64 / 50), every sample against the
    per-sample loop of single-image steps (the reference's structure) and against the Gaussian-stationary kernel"""
    from gsasr_amd import gaussian_splatting as gsp, synthetic
    sizes = [(40, 56), (64, 37), (50, 50)]
    h_lr, w_lr = 10, 12
    p = torch.stack([synthetic.gs_parameters(h_lr, w_lr, seed=80 + b, gpp=gpp) for b in range(3)]).to(dev)
    scales = [4.0, 5.3, 4.4]
    sms = [torch.tensor([s, s]) for s in scales]
    hm, wm = max(h for h, _ in sizes), max(w for _, w in sizes)
    wgt = torch.rand(3, 3, hm, wm, generator=torch.Generator().manual_seed(5)).to(dev)
    old = gsp.BACKWARD_KERNEL
    res = {}
    try:
        for mode in ("gaussian", "home"):
            gsp.BACKWARD_KERNEL = mode
            pa = p.clone().requires_grad_(True)
            out = gsp.generate_2D_gaussian_splatting_batch(sizes, pa, scales, sms, dmax=0.4)
            (out * wgt).sum().backward()
            res[mode] = pa.grad.cpu().numpy()
    finally:
        gsp.BACKWARD_KERNEL = old
    for b in range(3):
        per_gaussian_ok(res["home"][b], res["gaussian"][b], f"sample {b}")
    pa = p.clone().requires_grad_(True)
    tot = 0
    for b in range(3):
        o = gsp.generate_2D_gaussian_splatting_step(sizes[b], pa[b], scales[b], sms[b], dmax=0.4)
        tot = tot + (o * wgt[b, :, :sizes[b][0], :sizes[b][1]]).sum()
    tot.backward()
    for b in range(3):
        per_gaussian_ok(res["home"][b], pa.grad[b].cpu().numpy(), f"sample {b} vs loop")


def test_dense_whole_images_take_the_home_tile_backward_by_default(dev):
    """The library's own rule (splat_common.h:bwd_wants_home): denser than one Gaussian per two pixels on at least 1024 tiles of
    32 x 16 px.  Both kernels are deterministic, so "the default ran the home-tile kernel" is bit-equality with the flagged run
    (and a last-bits difference from the Gaussian-stationary one, which sums in another order); one tile row fewer, a row band
    or too few tiles keep the Gaussian-stationary kernel."""
    from gsasr_amd import _cabi
    sig, xy, col, H, W, wgt = _synth(192, 256, 4.0, seed=31, gpp=2)       # 768 x 1024: 48 x 32 = 1536 tiles; 98 304 Gaussians
    rep = 5                                                              # the rule reads the density: five jittered copies, 491 520 > 768 * 1024 / 2
    sig, xy, col = (np.tile(a, (rep, 1)) for a in (sig, xy, col))
    xy = xy + np.random.default_rng(5).normal(0.0, 2.0 / W, xy.shape).astype(np.float32)
    dflt = _backward(sig, xy, col, wgt, H, W, 0.1, dev, 0)
    home = _backward(sig, xy, col, wgt, H, W, 0.1, dev, _home())
    gaus = _backward(sig, xy, col, wgt, H, W, 0.1, dev, _cabi.FLAG_BWD_GAUSSIAN)
    for d, h_, g in zip(dflt, home, gaus):
        assert np.array_equal(d, h_)
        assert not np.array_equal(d, g)
        assert np.abs(d - g).max() <= 2e-5 * np.abs(g).max()
    # 768 x 768: 1152 tiles = 2.25 sets of the 512 workgroups the chip runs at a time -- the one-cell tile shape (launch_bwd_home's
    # variant 3; the image above runs whole sets of 32 x 16-px tiles)
    s2, x2, c2, H2, W2, w2 = _synth(192, 192, 4.0, seed=33, gpp=2)
    s2, x2, c2 = (np.tile(a, (rep, 1)) for a in (s2, x2, c2))
    x2 = x2 + np.random.default_rng(6).normal(0.0, 2.0 / W2, x2.shape).astype(np.float32)
    dflt = _backward(s2, x2, c2, w2, H2, W2, 0.25, dev, 0)
    home = _backward(s2, x2, c2, w2, H2, W2, 0.25, dev, _home())
    gaus = _backward(s2, x2, c2, w2, H2, W2, 0.25, dev, _cabi.FLAG_BWD_GAUSSIAN)
    for d, h_, g in zip(dflt, home, gaus):
        assert np.array_equal(d, h_)
        assert not np.array_equal(d, g)
        assert np.abs(d - g).max() <= 2e-5 * np.abs(g).max()
    # 31 tile rows of 32 (992 tiles): the Gaussian-stationary kernel, bit for bit
    Hs = 496
    dflt = _backward(sig, xy, col, wgt[:Hs], Hs, W, 0.1, dev, 0)
    gaus = _backward(sig, xy, col, wgt[:Hs], Hs, W, 0.1, dev, _cabi.FLAG_BWD_GAUSSIAN)
    for d, g in zip(dflt, gaus):
        assert np.array_equal(d, g)
    # a row band of the dense image: the Gaussian-stationary kernel as well
    dflt = _backward(sig, xy, col, wgt, H, W, 0.1, dev, 0, rows=(64, 256))
    gaus = _backward(sig, xy, col, wgt, H, W, 0.1, dev, _cabi.FLAG_BWD_GAUSSIAN, rows=(64, 256))
    for d, g in zip(dflt, gaus):
        assert np.array_equal(d, g)
