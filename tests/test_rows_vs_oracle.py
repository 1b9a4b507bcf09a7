"""Oracle-compared GPU tests for the SURVEY.md 8 rows that round 1 only compared with this library's own output
(a4/a5 the `*_buffer` renderers, a9 both `gaussiansplatting_render` helpers, f3 the tiled driver), the full-size
config-2 checks for dmax 0.5 and the unbounded op, and raw-op inputs the host API never produces (negative sigmas).

References: utils/gaussian_splatting.py:100-117,133-155,219-265; utils/gs_cuda/gswrapper.py:41-48;
utils/gs_cuda_dmax/gswrapper.py:46-53; utils/split_and_joint_image.py:160-225.
"""
import glob
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tiled_models  # noqa: E402

pytestmark = pytest.mark.gpu
IMG_ATOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run with -m gpu on the MI355X box)"
    return torch.device("cuda:0")


def _oracle_image(p, H, W, scale_modify, dmax, dmax_mode="fix"):
    """[3,H,W] float64: oracle(prologue(gs_parameters)) -- the reference's host prologue restated on the CPU
    (oracle/host_ref.py, pinned by tests/golden/prologue_*.npz) in front of the oracle's exact-semantics splat"""
    from oracle import gs_oracle, host_ref
    sig, xy, col, dm = host_ref.prologue(p.cpu(), (H, W), scale_modify.cpu(), dmax=25 if dmax is None else dmax, dmax_mode=dmax_mode)
    ref = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, None if dmax is None else dm)
    return np.transpose(ref, (2, 0, 1))


@pytest.mark.parametrize("kw", [dict(if_dmax=True, dmax_mode="fix", dmax=0.3), dict(if_dmax=True, dmax_mode="dynamic", dmax=25),
                                dict(if_dmax=False)], ids=["fix0.3", "dynamic25", "unbounded"])
@pytest.mark.parametrize("buffer_size", [300, 1600, 5000], ids=["6chunks", "exact-multiple", "one-chunk"])
def test_buffer_renderers_against_oracle(kw, buffer_size, dev):
    """a4/a5: generate_2D_gaussian_splatting_step_buffer chains GSCUDA.apply over `buffer_size` slices on the +=
    contract (reference :146-151 runs len//buffer_size + 1 slices, the last possibly empty)"""
    from gsasr_amd import gaussian_splatting as gsp, synthetic
    h_lr, w_lr, scale = 40, 40, 4.0
    H, W = 160, 160
    p = synthetic.gs_parameters(h_lr, w_lr, seed=50).to(dev)
    sm = torch.tensor([scale, scale], device=dev)
    out = gsp.generate_2D_gaussian_splatting_step_buffer((H, W), p, scale, sm, buffer_size=buffer_size, **kw)
    dm = None if not kw["if_dmax"] else kw["dmax"]
    ref = _oracle_image(p, H, W, sm, dm, kw.get("dmax_mode", "fix"))
    assert out.shape == (3, H, W)
    assert np.abs(out.cpu().numpy() - ref).max() <= IMG_ATOL


def test_rendering_cuda_buffer_functions_against_oracle(dev):
    """a4: rendering_cuda_buffer / rendering_cuda_dmax_buffer called directly with activated properties"""
    from gsasr_amd import gaussian_splatting as gsp, synthetic
    from oracle import gs_oracle, host_ref
    H, W, scale = 90, 124, 3.0
    p = synthetic.gs_parameters(30, 41, seed=52)
    sx, sy, rho, xy, col = (t.to(dev) for t in host_ref.activations(p))
    step = 1.2 / scale
    sig_k, xy_k, col_k, _ = host_ref.prologue(p, (H, W), torch.tensor([scale, scale]))
    for dmax in (None, 0.15):
        if dmax is None:
            out = gsp.rendering_cuda_buffer(sx, sy, rho, xy, col, (H, W), step, dev, buffer_size=500)
        else:
            out = gsp.rendering_cuda_dmax_buffer(sx, sy, rho, xy, col, (H, W), step, dev, dmax=dmax, buffer_size=500)
        ref = gs_oracle.forward_f64(sig_k.numpy(), xy_k.numpy(), col_k.numpy(), H, W, dmax)
        assert np.abs(out.permute(1, 2, 0).cpu().numpy() - ref).max() <= IMG_ATOL


def test_gaussiansplatting_render_helpers_against_oracle(dev):
    """a9: the two convenience wrappers (contiguous-ify, zero image of image_size[:2], apply; dmax defaults to 100)"""
    from gsasr_amd import synthetic
    from gsasr_amd.gs_cuda.gswrapper import gaussiansplatting_render as render_unbounded
    from gsasr_amd.gs_cuda_dmax.gswrapper import gaussiansplatting_render as render_dmax
    from oracle import gs_oracle
    sig, xy, col, H, W = synthetic.kernel_inputs(20, 27, 4.0, seed=53)
    s, c, k = sig.numpy(), xy.numpy(), col.numpy()
    a, b, d = sig.to(dev), xy.to(dev), col.to(dev)
    # non-contiguous inputs are accepted (the wrappers call .contiguous(), gswrapper.py:42-44)
    a_nc = torch.stack([a, a], 1)[:, 0]
    img = render_unbounded(a_nc, b, d, (H, W, 3))
    assert img.shape == (H, W, 3) and np.abs(img.cpu().numpy() - gs_oracle.forward_f64(s, c, k, H, W, None)).max() <= IMG_ATOL
    img = render_dmax(a_nc, b, d, (H, W), 0.2)
    assert np.abs(img.cpu().numpy() - gs_oracle.forward_f64(s, c, k, H, W, 0.2)).max() <= IMG_ATOL
    img = render_dmax(a, b, d, (H, W))                      # dmax=100: the box never binds
    assert np.abs(img.cpu().numpy() - gs_oracle.forward_f64(s, c, k, H, W, 100.0)).max() <= IMG_ATOL
    # and they are differentiable like GSCUDA.apply
    ag = a.clone().requires_grad_(True)
    wgt = synthetic.grad_image(H, W, 54)
    (render_dmax(ag, b, d, (H, W), 0.2) * wgt.to(dev)).sum().backward()
    want = gs_oracle.backward_f64(s, c, k, wgt.numpy(), 0.2)[0]
    assert np.abs(ag.grad.cpu().numpy() - want).max() <= 2e-4 * np.abs(want).max()


_TILED = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiled_*.npz")))


@pytest.mark.parametrize("path", _TILED, ids=[os.path.basename(p)[6:-4] for p in _TILED])
def test_tiled_driver_on_gpu_golden_shapes_against_oracle_canvas(path, dev):
    """f3 on the GPU at the eight shapes of the reference-captured goldens (tile grids incl. one row / one column, crop 0,
    scales 1.2 ... 4): batched canvases of tiles against a canvas of ORACLE tiles pasted with the restated case tree.
    (The goldens' own `out` is the reference's CPU approximation `rendering_python`; here the kernels' exact maths.)"""
    from gsasr_amd.split_and_joint_image import split_and_joint_image
    z = np.load(path)
    lq = torch.from_numpy(z["lq"])
    scale, split, overlap, crop = float(z["scale"]), int(z["split_size"]), int(z["overlap_size"]), int(z["crop_size"])
    hl, wl = lq.shape[-2:]
    sm = torch.tensor([scale, scale])
    out = split_and_joint_image(lq.to(dev), scale, split, overlap, tiled_models.model_g, tiled_models.model_fea2gs, sm.to(dev),
                                crop_size=crop, if_dmax=True, dmax_mode="fix", dmax=0.4)
    stride = split - overlap
    nh, nw = math.ceil((hl - overlap) / stride), math.ceil((wl - overlap) / stride)
    lq_pad = torch.nn.functional.pad(lq, (0, nw * stride + overlap - wl, 0, nh * stride + overlap - hl), mode="reflect")
    size_sr, overlap_sr = math.ceil(split * scale), math.ceil(overlap * scale)
    st = size_sr - overlap_sr
    want = np.zeros((1, 3, (nh - 1) * st + size_sr, (nw - 1) * st + size_sr), np.float64)
    frac = scale != int(scale)
    for i in range(nh):
        for j in range(nw):
            tile = lq_pad[:, :, i * stride: i * stride + split, j * stride: j * stride + split]
            p = tiled_models.model_fea2gs(tiled_models.model_g(tile), sm[0].unsqueeze(0))[0]
            t = _oracle_image(p, size_sr, size_sr, sm, 0.4)
            top = 0 if i == 0 else crop
            left = 0 if j == 0 else crop
            if frac and i > 0 and j > 0 and j == nw - 1 and i != nh - 1:
                top = 0
            if frac and i > 0 and j > 0 and i == nh - 1 and j != nw - 1:
                left = 0
            want[0, :, i * st + top: i * st + size_sr, j * st + left: j * st + size_sr] = t[:, top:, left:]
    assert out.shape == want.shape
    assert np.abs(out.cpu().numpy() - want).max() <= IMG_ATOL


@pytest.mark.parametrize("scale", [2.0, 2.5])
def test_tiled_driver_on_gpu_against_oracle_canvas(scale, dev):
    """f3: the GPU tiled run (cuda_rendering=True, batched canvases) against a canvas assembled on the CPU from ORACLE
    renders of every tile, pasted with the reference's rule (restated here from utils/split_and_joint_image.py:160-222
    as the explicit case tree, independently of the driver's `_paste_rule`)"""
    from gsasr_amd.split_and_joint_image import split_and_joint_image
    torch.manual_seed(5)
    lq = torch.rand(1, 3, 40, 52)
    split, overlap, crop = 12, 3, 2
    kw = dict(if_dmax=True, dmax_mode="fix", dmax=0.4)
    sm = torch.tensor([scale, scale])
    out = split_and_joint_image(lq.to(dev), scale, split, overlap, tiled_models.model_g, tiled_models.model_fea2gs, sm.to(dev),
                                crop_size=crop, **kw)
    stride = split - overlap
    nh, nw = math.ceil((40 - overlap) / stride), math.ceil((52 - overlap) / stride)
    lq_pad = torch.nn.functional.pad(lq, (0, nw * stride + overlap - 52, 0, nh * stride + overlap - 40), mode="reflect")
    size_sr, overlap_sr = math.ceil(split * scale), math.ceil(overlap * scale)
    st = size_sr - overlap_sr
    want = np.zeros((1, 3, (nh - 1) * st + size_sr, (nw - 1) * st + size_sr), np.float64)
    frac = scale != int(scale)
    for i in range(nh):
        for j in range(nw):
            tile = lq_pad[:, :, i * stride: i * stride + split, j * stride: j * stride + split]
            p = tiled_models.model_fea2gs(tiled_models.model_g(tile), sm[0].unsqueeze(0))[0]
            t = _oracle_image(p, size_sr, size_sr, sm, 0.4)
            # reference case tree: first row/column keep everything on that side; otherwise `crop` rows/columns go,
            # except (fractional scale only) the top of a last-column tile and the left of a last-row tile that are
            # not the corner
            top = 0 if i == 0 else crop
            left = 0 if j == 0 else crop
            if frac and i > 0 and j > 0 and j == nw - 1 and i != nh - 1:
                top = 0
            if frac and i > 0 and j > 0 and i == nh - 1 and j != nw - 1:
                left = 0
            want[0, :, i * st + top: i * st + size_sr, j * st + left: j * st + size_sr] = t[:, top:, left:]
    assert out.shape == want.shape
    assert np.abs(out.cpu().numpy() - want).max() <= IMG_ATOL


@pytest.mark.parametrize("dmax", [0.5, None], ids=["dmax0.5", "unbounded"])
def test_config2_full_size_other_variants_against_oracle(dmax, dev):
    """BASELINE config 2 in full (1024^2, 65 536 Gaussians) for the training box and the unbounded op: a 64-row band
    of the image and the band's gradient against the oracle (every Gaussian, exact box semantics; seconds on the CPU)"""
    from gsasr_amd import _cabi, synthetic
    from oracle import gs_oracle
    sig, xy, col, H, W = synthetic.kernel_inputs(256, 256, 4.0, seed=0)
    wgt = synthetic.grad_image(H, W, 1)
    s, c, k = sig.numpy(), xy.numpy(), col.numpy()
    rows = (480, 544)
    a, b, d = sig.to(dev), xy.to(dev), col.to(dev)
    plan = _cabi.plan(a, b, d, H, W, dmax)
    img = torch.empty(H, W, 3, device=dev)
    _cabi.forward(plan, img, overwrite=True)
    ref = gs_oracle.forward_f64(s, c, k, H, W, dmax, rows=rows)
    assert np.abs(img[rows[0]:rows[1]].cpu().numpy() - ref).max() <= IMG_ATOL
    band = _cabi.plan(a, b, d, H, W, dmax, rows=rows)
    g = [torch.empty_like(t) for t in (a, b, d)]
    _cabi.backward(band, a, b, d, wgt[rows[0]:rows[1]].contiguous().to(dev), *g, overwrite=True)
    want = gs_oracle.backward_f64(s, c, k, wgt[rows[0]:rows[1]].numpy(), dmax, h=H, rows=rows)
    from test_bwd_tile import per_gaussian_ok
    for got, w_, name in zip(g, want, ("sigmas", "coords", "colors")):
        per_gaussian_ok(got.cpu().numpy(), w_, name, rho=s[:, 2])


def test_raw_op_negative_sigmas_match_reference_semantics(dev):
    """the reference kernels only see sigma^2 and 1/(sx sy) (gs.cu:33-56): a negative sigma is a Gaussian with the
    sign of rho's cross term flipped, not a dead one.  check.py's own __main__ feeds randn sigmas."""
    from test_hip_parity import _check
    rng = np.random.default_rng(11)
    s, h, w = 50, 60, 75
    sig = np.stack([rng.uniform(0.03, 0.3, s) * rng.choice([-1, 1], s), rng.uniform(0.03, 0.3, s) * rng.choice([-1, 1], s),
                    rng.uniform(-0.8, 0.8, s)], 1).astype(np.float32)
    assert (sig[:, 0] < 0).any() and (sig[:, 1] < 0).any()
    xy = rng.uniform(-1, 1, (s, 2)).astype(np.float32)
    col = rng.uniform(0, 1, (s, 3)).astype(np.float32)
    wgt = rng.uniform(0, 1, (h, w, 3)).astype(np.float32)
    for dmax in (None, 0.4):
        _check(sig, xy, col, h, w, dmax, dev, wgt)
    from test_bwd_tile import _check as check_tile, _flags
    check_tile(sig, xy, col, wgt, h, w, 0.4, dev, _flags()["tile"])


def test_batch_larger_than_one_canvas_is_split(dev):
    """20 samples of 2000 x 24 px: 20 slots of 2000 rows exceed the canvas limit of 32 767 rows, so the batch runs as
    two canvases (16 + 4); the result equals the per-sample loop"""
    from gsasr_amd import gaussian_splatting as gsp, synthetic
    assert gsp.max_canvas_batch(1920) == 17 and gsp.max_canvas_batch(4096) == 7 and gsp.max_canvas_batch(512) == 63
    assert gsp.max_canvas_batch(2000) == 16 and gsp.max_canvas_batch(192) == 64 and gsp.max_canvas_batch(20000) == 1
    B, H, W = 20, 2000, 24
    p = torch.stack([synthetic.gs_parameters(50, 3, seed=300 + b) for b in range(B)]).to(dev)
    sm = [torch.tensor([8.0, 8.0]) for _ in range(B)]
    pa = p.clone().requires_grad_(True)
    out = gsp.generate_2D_gaussian_splatting_batch([(H, W)] * B, pa, [8.0] * B, sm, dmax=0.2)
    assert out.shape == (B, 3, H, W)
    out.square().sum().backward()
    pb = p.clone().requires_grad_(True)
    ref = torch.stack([gsp.generate_2D_gaussian_splatting_step((H, W), pb[b], 8.0, sm[b], dmax=0.2) for b in range(B)])
    ref.square().sum().backward()
    assert float((out - ref).detach().abs().max()) <= 2e-5
    assert float((pa.grad - pb.grad).abs().max()) <= 2e-4 * float(pb.grad.abs().max())


def test_config3_full_size_band_against_oracle(dev):
    """BASELINE config 3 in full (512x512 LR -> x12 = 6144^2, 262 144 Gaussians, dmax 0.1, forward): a 16-row band of
    the image against the oracle with every Gaussian (exact box semantics)"""
    from gsasr_amd import _cabi, synthetic
    from oracle import gs_oracle
    sig, xy, col, H, W = synthetic.kernel_inputs(512, 512, 12.0, seed=0)
    assert (H, W, sig.shape[0]) == (6144, 6144, 262144)
    a, b, c = sig.to(dev), xy.to(dev), col.to(dev)
    plan = _cabi.plan(a, b, c, H, W, 0.1, flags=_cabi.FLAG_FORWARD_ONLY)
    img = torch.empty(H, W, 3, device=dev)
    _cabi.forward(plan, img, overwrite=True)
    for rows in ((3056, 3072), (0, 16)):
        ref = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, 0.1, rows=rows)
        assert np.abs(img[rows[0]:rows[1]].cpu().numpy() - ref).max() <= IMG_ATOL


def test_config4_full_size_band_against_oracle(dev):
    """BASELINE config 4 on one GPU (1024x1024 LR -> x8 = 8192^2, 1 048 576 Gaussians, dmax 0.1): a 16-row band of the
    image, and the gradient of that band through BOTH backward kernels (the tile-stationary one is the library's
    default at this scale), against the oracle with every Gaussian"""
    from gsasr_amd import _cabi, synthetic
    from oracle import gs_oracle
    from test_bwd_tile import per_gaussian_ok
    sig, xy, col, H, W = synthetic.kernel_inputs(1024, 1024, 8.0, seed=0)
    assert (H, W, sig.shape[0]) == (8192, 8192, 1048576)
    s, c, k = sig.numpy(), xy.numpy(), col.numpy()
    a, b, d = sig.to(dev), xy.to(dev), col.to(dev)
    rows = (4088, 4104)
    wgt = synthetic.grad_image(rows[1] - rows[0], W, 3)
    want = gs_oracle.backward_f64(s, c, k, wgt.numpy(), 0.1, h=H, rows=rows)
    ref = gs_oracle.forward_f64(s, c, k, H, W, 0.1, rows=rows)
    for flag in (_cabi.FLAG_BWD_TILE, _cabi.FLAG_BWD_GAUSSIAN):
        band = _cabi.plan(a, b, d, H, W, 0.1, rows=rows, flags=flag)
        img = torch.empty(rows[1] - rows[0], W, 3, device=dev)
        _cabi.forward(band, img, overwrite=True)
        assert np.abs(img.cpu().numpy() - ref).max() <= IMG_ATOL
        g = [torch.empty_like(t) for t in (a, b, d)]
        _cabi.backward(band, a, b, d, wgt.to(dev), *g, overwrite=True)
        for got, w_, name in zip(g, want, ("sigmas", "coords", "colors")):
            per_gaussian_ok(got.cpu().numpy(), w_, name, rho=s[:, 2])
    # the whole image at this size plans for the tile-stationary backward by default (slots in the workspace)
    import ctypes
    whole, forced = _cabi.make_dims(sig.shape[0], H, W, 0.1), _cabi.make_dims(sig.shape[0], H, W, 0.1, flags=_cabi.FLAG_BWD_GAUSSIAN)
    L = _cabi.lib()
    assert L.gsasr_splat_workspace_bytes(ctypes.byref(whole)) > L.gsasr_splat_workspace_bytes(ctypes.byref(forced))


def test_large_scale_forward_against_oracle(dev):
    """x32 inference shape (57x72 LR -> 1824x2304, 1024 HR pixels per Gaussian, windows of ~170 px): the library renders
    it with the wide forward (16x16 sub-tiles; gsasr_splat_forward).  1824 rows = 57 tiles of 32, 2304 columns = 72.
    Bands at the top, across a tile boundary and at the end, bounded and unbounded, against the oracle with every
    Gaussian -- through the default choice and through the 8x16 kernels"""
    from gsasr_amd import _cabi, synthetic
    from oracle import gs_oracle
    sig, xy, col, H, W = synthetic.kernel_inputs(57, 72, 32.0, seed=5)
    assert (H, W) == (1824, 2304) and (W // 8) * (H // 16) >= 16384 and H * W >= 25 * sig.shape[0]
    a, b, c = sig.to(dev), xy.to(dev), col.to(dev)
    for dmax in (0.1, None):
        plan = _cabi.plan(a, b, c, H, W, dmax, flags=_cabi.FLAG_FORWARD_ONLY)
        for flag in (0, _cabi.FLAG_FWD_NARROW):
            img = torch.full((H, W, 3), float("nan"), device=dev)
            _cabi.forward(plan, img, overwrite=True, flags=flag)
            assert bool(torch.isfinite(img).all())
            for rows in ((0, 16), (56, 72), (1784, 1824)):
                ref = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, dmax, rows=rows)
                assert np.abs(img[rows[0]:rows[1]].cpu().numpy() - ref).max() <= IMG_ATOL * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("dmax", [0.1, 0.5])
def test_real_density_inference_size_band_against_oracle(dmax, dev):
    """GSASR's real Gaussian density at inference size (Fea2GS emits 16 Gaussians per LR pixel,
    utils/fea2gs.py:546-551): 256x256 LR -> x4 = 1024^2, 1 048 576 Gaussians, inference box 0.1 and training box 0.5.
    A 16-row band of the whole-image forward, and the gradient of that band through both backward kernels, against the
    oracle with every Gaussian (image sums of ~500 terms: tolerance relative to the band's peak, as at x32)."""
    from gsasr_amd import _cabi, synthetic
    from oracle import gs_oracle
    from test_bwd_tile import per_gaussian_ok
    sig, xy, col, H, W = synthetic.kernel_inputs(256, 256, 4.0, seed=0, gpp=16)
    assert (H, W, sig.shape[0]) == (1024, 1024, 1048576)
    s, c, k = sig.numpy(), xy.numpy(), col.numpy()
    a, b, d = sig.to(dev), xy.to(dev), col.to(dev)
    rows = (504, 520)
    ref = gs_oracle.forward_f64(s, c, k, H, W, dmax, rows=rows)
    plan = _cabi.plan(a, b, d, H, W, dmax, flags=_cabi.FLAG_FORWARD_ONLY)
    img = torch.empty(H, W, 3, device=dev)
    _cabi.forward(plan, img, overwrite=True)
    atol = IMG_ATOL * max(1.0, float(np.abs(ref).max()))
    assert np.abs(img[rows[0]:rows[1]].cpu().numpy() - ref).max() <= atol
    wgt = synthetic.grad_image(rows[1] - rows[0], W, 3)
    want = gs_oracle.backward_f64(s, c, k, wgt.numpy(), dmax, h=H, rows=rows)
    for flag in (_cabi.FLAG_BWD_GAUSSIAN, _cabi.FLAG_BWD_TILE):
        band = _cabi.plan(a, b, d, H, W, dmax, rows=rows, flags=flag)
        slab = torch.empty(rows[1] - rows[0], W, 3, device=dev)
        _cabi.forward(band, slab, overwrite=True)
        assert np.abs(slab.cpu().numpy() - ref).max() <= atol
        g = [torch.empty_like(t) for t in (a, b, d)]
        _cabi.backward(band, a, b, d, wgt.to(dev), *g, overwrite=True)
        for got, w_, name in zip(g, want, ("sigmas", "coords", "colors")):
            per_gaussian_ok(got.cpu().numpy(), w_, name, rho=s[:, 2])


def test_cross_fuzz_large_and_thin_images(dev):
    """tools/fuzz_cross.py, 60 cases: random images up to ~5000 px a side (some only a few pixels thin), a random row
    band, Gaussians from hairlines to several image widths -- forward and all three backward kernels against the oracle on
    that band.  (The sizes the small fuzzers never reach: round 3's one parity bug lived there.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_cross.py"), "60", "11"], capture_output=True, text=True,
                       timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "60 cases ok" in r.stdout


@pytest.mark.parametrize("bwd8", ["0", "1"], ids=["wave-per-gaussian-everywhere", "eight-per-wave-everywhere"])
def test_cross_fuzz_both_gaussian_stationary_backwards(bwd8, dev):
    """the Gaussian-stationary backward has two kernels since round 5 -- one wave per Gaussian (k_render_bwd) and eight
    Gaussians per wave (k_render_bwd8, the default below 32 HR pixels per Gaussian): the same fuzz with each of them forced
    on EVERY case (development switch GSASR_SPLAT_BWD8 under GSASR_SPLAT_DEV=1), windows of hundreds of pixels included"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GSASR_SPLAT_DEV="1", GSASR_SPLAT_BWD8=bwd8)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_cross.py"), "40", "23"], capture_output=True, text=True,
                       timeout=900, cwd=root, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "40 cases ok" in r.stdout


def test_sixteen_million_gaussians_properties(dev):
    """tools/big_n_check.py at 1024^2 LR x 16 Gaussians per LR pixel x4 (16 777 216 Gaussians on 4096^2): per-Gaussian
    gradients independent of the other Gaussians (both backward kernels), forward additive over a split"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "big_n_check.py"), "1024", "16", "4"], capture_output=True, text=True,
                       timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "forward additivity over two halves" in r.stdout


def test_published_workload_bands_against_oracle(dev):
    """The reference's one published workload (utils/gs_cuda/profile.py:104-113: unbounded op, 512 x 512 image, 262 144
    Gaussians with sigma in [0, 1) of the grid -- 99% of them in the plan's "large" class): row bands of the image against
    the oracle with every Gaussian, at the default cutoff and with the cull off.  A pixel sums ~1e5 terms of size <= 1, so
    the criterion is relative to the image's largest value (fp32 accumulation), not the 1e-4 absolute of GSASR's colours."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from benchlib import published_inputs
    from gsasr_amd import _cabi
    from oracle import gs_oracle
    sig, xy, col, H, W = published_inputs()
    assert (H, W, sig.shape[0]) == (512, 512, 262144)
    a, b, c = sig.to(dev), xy.to(dev), col.to(dev)
    bands = ((0, 2), (255, 257), (510, 512))
    refs = [gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, None, rows=r) for r in bands]
    top = max(float(np.abs(r).max()) for r in refs)
    assert top > 1e3
    for tau in (0.0, -1.0):
        plan = _cabi.plan(a, b, c, H, W, None, cutoff=tau, flags=_cabi.FLAG_FORWARD_ONLY)
        img = torch.empty(H, W, 3, device=dev)
        _cabi.forward(plan, img, overwrite=True)
        for r, ref in zip(bands, refs):
            # (a pixel is a sequential fp32 sum of 2.6e5 terms: sqrt(n) * 2^-24 ~ 3e-5 of the sum is its rounding noise)
            err = float(np.abs(img[r[0]:r[1]].cpu().numpy() - ref).max())
            assert err <= 2e-4 * top, (tau, r, err, top)


@pytest.mark.parametrize("config", ["c3", "c4"])
def test_full_size_scattered_rows_against_oracle(config, dev):
    """the full-size comparisons of configs 3 and 4 were 16-row bands at two places.  Here single rows
    scattered over the WHOLE image (tile boundaries, the first and last row, rows that straddle the XCD bands of the launch
    order) of the default render -- x12 wide forward; x8 wide forward from the plan's tile lists -- against the oracle with
    every Gaussian, and at config 4 the gradient of those rows alone through the library's default backward"""
    from gsasr_amd import _cabi, synthetic
    from oracle import gs_oracle
    from test_bwd_tile import per_gaussian_ok
    h_lr, scale = (512, 12.0) if config == "c3" else (1024, 8.0)
    sig, xy, col, H, W = synthetic.kernel_inputs(h_lr, h_lr, scale, seed=0)
    s, c, k = sig.numpy(), xy.numpy(), col.numpy()
    a, b, d = sig.to(dev), xy.to(dev), col.to(dev)
    plan = _cabi.plan(a, b, d, H, W, 0.1, flags=_cabi.FLAG_FORWARD_ONLY if config == "c3" else 0)
    img = torch.empty(H, W, 3, device=dev)
    _cabi.forward(plan, img, overwrite=True)
    rng = np.random.default_rng(7)
    rows = sorted(set([0, H - 1, H // 8 - 1, H // 8, 5 * H // 8 + 31, 5 * H // 8 + 32] + [int(r) for r in rng.integers(0, H, 4)]))
    for r in rows:
        ref = gs_oracle.forward_f64(s, c, k, H, W, 0.1, rows=(r, r + 1))
        assert np.abs(img[r:r + 1].cpu().numpy() - ref).max() <= IMG_ATOL, (config, r)
    if config == "c4":
        # the gradient of three scattered rows: upstream gradient zero everywhere else, whole-image default backward
        pick = rows[2:5]
        wgt = torch.zeros(H, W, 3)
        full = synthetic.grad_image(H, W, 11)
        want = None
        for r in pick:
            wgt[r] = full[r]
            g1 = gs_oracle.backward_f64(s, c, k, full[r:r + 1].contiguous().numpy(), 0.1, h=H, rows=(r, r + 1))
            want = g1 if want is None else tuple(x + y for x, y in zip(want, g1))
        g = [torch.empty_like(t) for t in (a, b, d)]
        _cabi.backward(plan, a, b, d, wgt.to(dev), *g, overwrite=True)
        for got, w_, name in zip(g, want, ("sigmas", "coords", "colors")):
            per_gaussian_ok(got.cpu().numpy(), w_, name, rho=s[:, 2])
