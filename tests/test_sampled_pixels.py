"""GPU parity of the sampled-pixel path: values and gradients at `sample_coords` only.

The reference renders the whole image and then gathers the points (utils/gaussian_splatting.py:214-216), so the
expected result is `gather(full render)` / the backward of a gradient image that is zero except at the points
(repeated points accumulate).  Checked against the CPU oracle at oracle-friendly sizes and against this
library's own full render elsewhere; tolerances as in test_hip_parity.py (1e-4 abs per value; gradients relative).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

IMG_ATOL = 1e-4
GRAD_RTOL = 2e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run with -m gpu on the MI355X box)"
    return torch.device("cuda:0")


def _relmax(got, want):
    return float(np.abs(got - want).max() / max(1e-12, np.abs(want).max()))


def _synth(h_lr, w_lr, scale, seed, gpp=1):
    from gsasr_amd import synthetic
    sig, xy, col, H, W = synthetic.kernel_inputs(h_lr, w_lr, scale, seed=seed, gpp=gpp)
    return sig, xy, col, H, W


def _points(H, W, n, seed, dev, dup=True):
    g = torch.Generator().manual_seed(seed)
    pts = torch.stack([torch.randint(0, H, (n,), generator=g), torch.randint(0, W, (n,), generator=g)], dim=1)
    if dup and n > 8:
        pts[5] = pts[3]          # repeated points are independent outputs whose gradients add
        pts[7] = pts[3]
    return pts.to(dev)


def _sampled(sig, xy, col, H, W, dmax, pts, gout, dev, cutoff=0.0):
    """sampled forward + backward through the plan API -> (out [3,S], (g_sigmas, g_coords, g_colors)) as numpy"""
    from gsasr_amd import _cabi
    a, b, c = (t.to(dev).contiguous() for t in (sig, xy, col))
    plan = _cabi.plan(a, b, c, H, W, dmax, cutoff=cutoff)
    out, state = _cabi.sample_forward(plan, pts)
    g = (torch.empty_like(a), torch.empty_like(b), torch.empty_like(c))
    _cabi.sample_backward(plan, state, a, b, c, gout, *g, overwrite=True)
    torch.cuda.synchronize()
    return out.cpu().numpy(), tuple(t.cpu().numpy() for t in g)


def _scatter(gout, pts, H, W):
    """the gradient image the reference's gather produces: zero except at the points, repeats accumulated"""
    wgt = torch.zeros(H, W, 3, dtype=torch.float32)
    p = pts.cpu().long()
    wgt.index_put_((p[:, 0], p[:, 1]), gout.cpu().t().contiguous(), accumulate=True)
    return wgt


@pytest.mark.parametrize("cutoff", [0.0, 104.0, -1.0], ids=["adaptive", "tau104", "nocut"])
@pytest.mark.parametrize("dmax", [None, 0.5, 0.1], ids=["unbounded", "dmax0.5", "dmax0.1"])
def test_sampled_against_oracle(dmax, cutoff, dev):
    from oracle import gs_oracle
    sig, xy, col, H, W = _synth(48, 40, 4.0, seed=21)
    pts = _points(H, W, 700, 5, dev)
    gout = torch.rand(3, pts.shape[0], generator=torch.Generator().manual_seed(6)).to(dev)
    out, grads = _sampled(sig, xy, col, H, W, dmax, pts, gout, dev, cutoff)
    s, x, c = sig.numpy(), xy.numpy(), col.numpy()
    ref = gs_oracle.forward_f64(s, x, c, H, W, dmax)
    p = pts.cpu().numpy()
    want = ref[p[:, 0], p[:, 1], :].T
    assert np.abs(out - want).max() <= IMG_ATOL
    gref = gs_oracle.backward_f64(s, x, c, _scatter(gout, pts, H, W).numpy(), dmax)
    for got, wnt, name in zip(grads, gref, ("sigmas", "coords", "colors")):
        assert np.isfinite(got).all(), name
        assert _relmax(got, wnt) <= GRAD_RTOL, name


@pytest.mark.parametrize("case", [(64, 64, 4.0, 1, 0.1, 2304), (24, 40, 2.5, 1, 0.5, 300), (12, 12, 4.0, 16, 0.5, 144),
                                  (20, 16, 12.0, 1, None, 1000), (128, 96, 8.0, 1, 0.1, 5000)])
def test_sampled_equals_gather_of_full_render(case, dev):
    """the same plan, the full forward/backward of this library as the reference (sizes the oracle would take minutes for)"""
    from gsasr_amd import _cabi
    h_lr, w_lr, scale, gpp, dmax, n = case
    sig, xy, col, H, W = _synth(h_lr, w_lr, scale, seed=3, gpp=gpp)
    pts = _points(H, W, n, 9, dev)
    gout = torch.randn(3, n, generator=torch.Generator().manual_seed(2)).to(dev)
    out, grads = _sampled(sig, xy, col, H, W, dmax, pts, gout, dev)
    a, b, c = (t.to(dev).contiguous() for t in (sig, xy, col))
    plan = _cabi.plan(a, b, c, H, W, dmax)
    img = torch.empty(H, W, 3, device=dev)
    _cabi.forward(plan, img, overwrite=True)
    p = pts.long()
    want = img[p[:, 0], p[:, 1], :].t()
    assert float((torch.from_numpy(out).to(dev) - want).abs().max()) <= 2e-5
    g = (torch.empty_like(a), torch.empty_like(b), torch.empty_like(c))
    _cabi.backward(plan, a, b, c, _scatter(gout, pts, H, W).to(dev), *g, overwrite=True)
    for got, wnt, name in zip(grads, g, ("sigmas", "coords", "colors")):
        assert _relmax(got, wnt.cpu().numpy()) <= GRAD_RTOL, name


def test_sampled_large_class_dead_and_out_of_range_points(dev):
    """large Gaussians (window > 128 px), dead ones (NaN / off-image), negative (wrapping) and out-of-range points"""
    from gsasr_amd import _cabi
    g = torch.Generator().manual_seed(4)
    n, H, W = 300, 300, 420
    sig = torch.cat([0.02 + 0.9 * torch.rand(n, 2, generator=g), 1.8 * torch.rand(n, 1, generator=g) - 0.9], dim=1)
    xy = 2.4 * torch.rand(n, 2, generator=g) - 1.2
    col = torch.rand(n, 3, generator=g)
    sig[7, 0] = float("nan")
    xy[9] = torch.tensor([30.0, -30.0])
    pts = _points(H, W, 400, 1, dev)
    pts[0] = torch.tensor([-1, -1])            # wraps to (H-1, W-1)
    pts[1] = torch.tensor([-H, 0])             # wraps to (0, 0)
    pts[2] = torch.tensor([H, 3])              # out of range: 0, no gradient
    pts[3] = torch.tensor([5, -W - 1])         # out of range
    gout = torch.rand(3, 400, generator=g).to(dev)
    for dmax in (None, 0.7):
        out, grads = _sampled(sig, xy, col, H, W, dmax, pts, gout, dev)
        a, b, c = (t.to(dev).contiguous() for t in (sig, xy, col))
        plan = _cabi.plan(a, b, c, H, W, dmax)
        img = torch.empty(H, W, 3, device=dev)
        _cabi.forward(plan, img, overwrite=True)
        valid = torch.ones(400, dtype=torch.bool)
        valid[2] = valid[3] = False
        p = pts.cpu().long().clone()
        p[:, 0] = torch.where(p[:, 0] < 0, p[:, 0] + H, p[:, 0])
        p[:, 1] = torch.where(p[:, 1] < 0, p[:, 1] + W, p[:, 1])
        want = img.cpu()[p[valid, 0], p[valid, 1], :].t().numpy()
        assert np.abs(out[:, valid.numpy()] - want).max() <= 2e-5
        assert (out[:, ~valid.numpy()] == 0).all()
        go = gout.clone()
        go[:, ~valid.to(dev)] = 0
        pv = p.clone()
        pv[~valid] = 0
        gg = (torch.empty_like(a), torch.empty_like(b), torch.empty_like(c))
        _cabi.backward(plan, a, b, c, _scatter(go, pv, H, W).to(dev), *gg, overwrite=True)
        for got, wnt, name in zip(grads, gg, ("sigmas", "coords", "colors")):
            assert np.isfinite(got).all(), name
            assert _relmax(got, wnt.cpu().numpy()) <= GRAD_RTOL, name
        assert (grads[0][7] == 0).all() and (grads[2][9] == 0).all()


def test_sampled_accumulate_mode_and_resort(dev):
    """without OVERWRITE_GRADS the backward adds into the caller's buffers (dmax contract); `resort` re-sorts the points"""
    from gsasr_amd import _cabi
    sig, xy, col, H, W = _synth(32, 32, 4.0, seed=8)
    a, b, c = (t.to(dev).contiguous() for t in (sig, xy, col))
    pts = _points(H, W, 500, 2, dev)
    gout = torch.rand(3, 500, device=dev)
    plan = _cabi.plan(a, b, c, H, W, 0.2)
    out, state = _cabi.sample_forward(plan, pts)
    g0 = (torch.empty_like(a), torch.empty_like(b), torch.empty_like(c))
    _cabi.sample_backward(plan, state, a, b, c, gout, *g0, overwrite=True)
    g1 = (torch.ones_like(a), torch.ones_like(b), torch.ones_like(c))
    _cabi.sample_backward(plan, state, a, b, c, gout, *g1, overwrite=False, resort=True)
    for x, y in zip(g0, g1):
        assert _relmax((y - 1).cpu().numpy(), x.cpu().numpy()) <= 1e-5
    # zero points: empty output, zero gradient
    out0, st0 = _cabi.sample_forward(plan, pts[:0])
    assert out0.shape == (3, 0)
    _cabi.sample_backward(plan, st0, a, b, c, gout[:, :0].contiguous(), *g1, overwrite=True)
    assert all(float(t.abs().max()) == 0 for t in g1)


def test_host_api_sample_coords_uses_the_sampled_path(dev):
    """generate_2D_gaussian_splatting_step(..., sample_coords=[S,2] tensor) == gather of the full render, with gradients"""
    from gsasr_amd import gaussian_splatting as gsp
    g = torch.Generator().manual_seed(12)
    n_lr = 24
    raw = (0.5 * torch.randn(n_lr * n_lr * 4, 9, generator=g))
    raw[:, 7:9] = torch.rand(n_lr * n_lr * 4, 2, generator=g)
    sr = (96, 96)
    pts = _points(96, 96, 600, 3, dev)
    wgt = torch.rand(3, 600, generator=g).to(dev)
    kw = dict(default_step_size=1.2, mode="scale_modify", if_dmax=True, dmax_mode="fix", dmax=0.3)
    p1 = raw.clone().to(dev).requires_grad_(True)
    o1 = gsp.generate_2D_gaussian_splatting_step(sr, p1, 4.0, torch.tensor([4.0, 4.0]), sample_coords=pts, **kw)
    assert o1.shape == (3, 600)
    (o1 * wgt).sum().backward()
    p2 = raw.clone().to(dev).requires_grad_(True)
    full = gsp.generate_2D_gaussian_splatting_step(sr, p2, 4.0, torch.tensor([4.0, 4.0]), **kw)
    o2 = torch.stack([full[:, c[0], c[1]] for c in pts.cpu()], dim=1)      # the reference's loop
    (o2 * wgt).sum().backward()
    assert float((o1 - o2).abs().max()) <= 2e-5
    assert _relmax(p1.grad.cpu().numpy(), p2.grad.cpu().numpy()) <= GRAD_RTOL
    # a python list of coordinates (what the docstring of the reference shows) takes the same path
    o3 = gsp.generate_2D_gaussian_splatting_step(sr, p1.detach(), 4.0, torch.tensor([4.0, 4.0]),
                                                 sample_coords=[(0, 0), (3, 7), (95, 95)], **kw)
    assert float((o3 - full.detach()[:, [0, 3, 95], [0, 7, 95]]).abs().max()) <= 2e-5


def test_batched_sampled_equals_per_sample(dev):
    """[B,N,9] + [B,S,2] points in one set of launches == B single-image sampled calls (ragged sizes)"""
    from gsasr_amd import gaussian_splatting as gsp
    g = torch.Generator().manual_seed(5)
    B, n = 4, 12 * 12 * 16
    raw = 0.5 * torch.randn(B, n, 9, generator=g)
    raw[:, :, 7:9] = torch.rand(B, n, 2, generator=g)
    sizes = [(48, 48), (40, 48), (48, 36), (33, 47)]
    scales = [4.0, 4.0, 3.0, 2.75]
    S = 300
    pts = torch.stack([torch.stack([torch.randint(0, h, (S,), generator=g), torch.randint(0, w, (S,), generator=g)], dim=1)
                       for h, w in sizes]).to(dev)
    wgt = torch.rand(B, 3, S, generator=g).to(dev)
    kw = dict(default_step_size=1.2, mode="scale_modify", if_dmax=True, dmax_mode="fix", dmax=0.5)
    p1 = raw.clone().to(dev).requires_grad_(True)
    o1 = gsp.generate_2D_gaussian_splatting_batch(sizes, p1, scales, [torch.tensor([s, s]) for s in scales],
                                                  sample_coords=pts, **kw)
    assert o1.shape == (B, 3, S)
    (o1 * wgt).sum().backward()
    p2 = raw.clone().to(dev).requires_grad_(True)
    outs = [gsp.generate_2D_gaussian_splatting_step(sizes[b], p2[b], scales[b], torch.tensor([scales[b]] * 2),
                                                    sample_coords=pts[b], **kw) for b in range(B)]
    o2 = torch.stack(outs)
    (o2 * wgt).sum().backward()
    assert float((o1 - o2).abs().max()) <= 2e-5
    assert _relmax(p1.grad.cpu().numpy(), p2.grad.cpu().numpy()) <= GRAD_RTOL


def test_sampled_path_errors_are_runtimeerrors(dev):
    """bad points, a row band, a mismatching gradient: RuntimeError (the reference's failure mode), never a crash"""
    from gsasr_amd import _cabi
    sig, xy, col, H, W = _synth(16, 16, 4.0, seed=1)
    a, b, c = (t.to(dev).contiguous() for t in (sig, xy, col))
    plan = _cabi.plan(a, b, c, H, W, 0.2)
    with pytest.raises(RuntimeError, match="points"):
        _cabi.sample_forward(plan, torch.zeros(5, 3, dtype=torch.int64, device=dev))
    with pytest.raises(RuntimeError, match="points"):
        _cabi.sample_forward(plan, torch.zeros(5, 2, device=dev))            # floating point
    band = _cabi.plan(a, b, c, H, W, 0.2, rows=(16, 32))
    with pytest.raises(RuntimeError, match="whole image"):
        _cabi.sample_forward(band, torch.zeros(5, 2, dtype=torch.int64, device=dev))
    out, st = _cabi.sample_forward(plan, torch.zeros(5, 2, dtype=torch.int64, device=dev))
    g = (torch.empty_like(a), torch.empty_like(b), torch.empty_like(c))
    with pytest.raises(RuntimeError, match="grad_out"):
        _cabi.sample_backward(plan, st, a, b, c, torch.zeros(3, 4, device=dev), *g)


def test_sampled_path_under_bf16_autocast(dev):
    """AMP configs run the op inside torch.autocast (gsasr_amp_model.py:208): the sampled Functions compute in fp32"""
    from gsasr_amd import gaussian_splatting as gsp
    g = torch.Generator().manual_seed(3)
    raw = 0.5 * torch.randn(16 * 16, 9, generator=g)
    raw[:, 7:9] = torch.rand(16 * 16, 2, generator=g)
    pts = _points(64, 64, 200, 4, dev)
    kw = dict(default_step_size=1.2, mode="scale_modify", if_dmax=True, dmax_mode="fix", dmax=0.4)
    p0 = raw.clone().to(dev).requires_grad_(True)
    ref = gsp.generate_2D_gaussian_splatting_step((64, 64), p0, 4.0, torch.tensor([4.0, 4.0]), sample_coords=pts, **kw)
    ref.sum().backward()
    p1 = raw.clone().to(dev).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = gsp.generate_2D_gaussian_splatting_step((64, 64), p1.bfloat16().float(), 4.0, torch.tensor([4.0, 4.0]),
                                                      sample_coords=pts, **kw)
        out2 = gsp.generate_2D_gaussian_splatting_step((64, 64), p1, 4.0, torch.tensor([4.0, 4.0]), sample_coords=pts, **kw)
    assert out.dtype == torch.float32 and out2.dtype == torch.float32
    out2.sum().backward()
    assert float((out2 - ref).abs().max()) <= 2e-6
    assert _relmax(p1.grad.cpu().numpy(), p0.grad.cpu().numpy()) <= 1e-5
