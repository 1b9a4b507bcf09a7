"""The data-derived support cutoff of the bounded op (include/gsasr_splat.h, `gsasr_plan_cutoff`; adapt_kcut in
gsasr_splat.hip).  Only a Gaussian whose dmax box covers a pixel contributes to it (utils/gs_cuda_dmax/gs.cu:41-50), so
at most K = max over pixels of the number of such Gaussians terms can be skipped on one pixel and tau' = ln(K / 1e-5)
keeps the bound `1e-5 * max|colour|` per pixel of the conservative tau = ln(N / 1e-5).  Checked here:

  * the K the plan derives from its cell histogram is an upper bound of the true K (brute force on the CPU) and
    tau' is the logarithm of THAT K -- on GSASR-shaped input it is well below ln(N / 1e-5);
  * adversarial input (every Gaussian stacked on one spot, or half of them) cannot break the bound: the image stays
    within 1e-5 * max|colour| of the exact render (tau = 104, the reference's set of non-zero fp32 terms);
  * an explicit cutoff, the process default and the unbounded op keep their cutoff;
  * Gaussians that k_classify keeps under the conservative cutoff but whose window under tau' holds no pixel (centres off
    the image, just out of reach) are rendered and differentiated like any other, by all three backward kernels.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

EPS = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run with -m gpu on the MI355X box)"
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def _true_k(xy, h, w, dmax):
    """max over pixels of #{s : |px - x_s| <= dmax and |py - y_s| <= dmax} with the kernels' float pixel grid"""
    px = (2.0 * np.arange(w) / (w - 1) - 1.0).astype(np.float32)
    py = (2.0 * np.arange(h) / (h - 1) - 1.0).astype(np.float32)
    inx = (np.abs(px[None, :] - xy[:, 0:1]) <= np.float32(dmax)).astype(np.float32)   # [N, w]
    iny = (np.abs(py[None, :] - xy[:, 1:2]) <= np.float32(dmax)).astype(np.float32)   # [N, h]
    return int((iny.T @ inx).max())


def _plan(sig, xy, col, h, w, dmax, dev, cutoff=0.0, flags=0):
    from gsasr_amd import _cabi
    a, b, c = (_t(x, dev) for x in (sig, xy, col))
    return _cabi.plan(a, b, c, h, w, dmax, cutoff=cutoff, flags=flags), (a, b, c)


def _image(plan, h, w, dev):
    from gsasr_amd import _cabi
    img = torch.empty(h, w, 3, device=dev)
    _cabi.forward(plan, img, overwrite=True)
    torch.cuda.synchronize()
    return img.cpu().numpy()


def _synth(h_lr, w_lr, scale, seed, gpp=1):
    from gsasr_amd import synthetic
    sig, xy, col, H, W = synthetic.kernel_inputs(h_lr, w_lr, scale, seed=seed, gpp=gpp)
    return sig.numpy(), xy.numpy(), col.numpy(), H, W


@pytest.mark.parametrize("case", [(64, 64, 4.0, 1, 0.1), (48, 40, 4.0, 16, 0.1), (40, 40, 8.0, 1, 0.05), (256, 256, 4.0, 1, 0.1),
                                  (96, 96, 12.0, 1, 0.1), (160, 100, 8.0, 1, 0.2)],
                         ids=["x4", "x4-16-per-lr-px", "x8", "config2", "x12-sparse-cells-block-count", "x8-two-pass-scan"])
def test_plan_k_bounds_the_true_k_and_sets_tau(case, dev):
    from gsasr_amd import _cabi
    h_lr, w_lr, scale, gpp, dmax = case
    sig, xy, col, H, W = _synth(h_lr, w_lr, scale, 7, gpp)
    plan, _ = _plan(sig, xy, col, H, W, dmax, dev)
    tau, k = _cabi.plan_cutoff(plan)
    n = sig.shape[0]
    tau_n = _cabi.resolve_cutoff(0.0, n)
    k_true = _true_k(xy, H, W, dmax)
    assert k >= k_true, (k, k_true)
    assert 16.0 <= tau <= tau_n + 1e-6
    if tau < tau_n - 1e-6 and tau > 16.0:
        assert abs(tau - (math.log(k / EPS) + 1e-3)) <= 2e-4 * tau, (tau, k)   # tau' is ln(K / eps) of the K reported
    if scale >= 12.0:
        # one Gaussian per 144 px: the largest CELL holds several times the mean, the plan counts in 4 x 4-cell blocks as well
        assert k <= 4 * k_true, (k, k_true)
    if h_lr >= 256:
        # BASELINE config 2: the plan's K is within 3x of the true one (cells are 16 px, the box 103) and tau' well under ln(N / eps)
        assert k <= 3 * k_true and tau <= tau_n - 3.0, (k, k_true, tau, tau_n)


def test_explicit_cutoff_process_default_and_unbounded_op_are_not_touched(dev):
    from gsasr_amd import _cabi
    sig, xy, col, H, W = _synth(32, 32, 4.0, 3)
    n = sig.shape[0]
    for cutoff, dmax, want in ((32.0, 0.1, 32.0), (104.0, 0.1, 104.0), (0.0, None, _cabi.resolve_cutoff(0.0, n))):
        plan, _ = _plan(sig, xy, col, H, W, dmax, dev, cutoff=cutoff)
        tau, k = _cabi.plan_cutoff(plan)
        assert abs(tau - want) <= 1e-4 * want and k == 0, (cutoff, dmax, tau, k)
    old = _cabi.get_default_cutoff()
    _cabi.set_default_cutoff(40.0)
    try:
        plan, _ = _plan(sig, xy, col, H, W, 0.1, dev)
        tau, k = _cabi.plan_cutoff(plan)
        assert abs(tau - 40.0) <= 1e-3 and k == 0
    finally:
        _cabi.set_default_cutoff(old)
    plan, _ = _plan(sig, xy, col, H, W, 0.1, dev, cutoff=-1.0)      # never skip: no window cutoff at all
    img = _image(plan, H, W, dev)
    plan104, _ = _plan(sig, xy, col, H, W, 0.1, dev, cutoff=104.0)
    assert np.abs(img - _image(plan104, H, W, dev)).max() <= 1e-6


@pytest.mark.parametrize("stacked", [1.0, 0.5, 0.05], ids=["all-on-one-spot", "half", "a-twentieth"])
def test_adversarial_stack_keeps_the_error_bound(stacked, dev):
    """N Gaussians with colour 1 on ONE spot, wide enough that the ring just outside their support holds N terms of
    exp(-tau) each: the worst case of the bound.  K must count them, and the image must stay within eps of the exact one."""
    from gsasr_amd import _cabi
    h = w = 192
    n = 8192
    g = np.random.default_rng(5)
    sig, xy, col, _, _ = _synth(48, 48, 4.0, 11, 4)
    sig, xy, col = sig[:n].copy(), xy[:n].copy(), col[:n].copy()
    m = int(n * stacked)
    xy[:m] = np.array([0.113, -0.207], np.float32)
    sig[:m, 0] = 0.02 + 0.002 * g.random(m)      # ~2 px: support radius ~12 px under tau ~ 20, inside the dmax box
    sig[:m, 1] = 0.02 + 0.002 * g.random(m)
    sig[:m, 2] = 0.0
    col[:m] = 1.0
    dmax = 0.25
    plan, _ = _plan(sig, xy, col, h, w, dmax, dev)
    tau, k = _cabi.plan_cutoff(plan)
    assert k >= _true_k(xy, h, w, dmax) >= m
    assert tau >= math.log(m / EPS) - 1e-3
    img = _image(plan, h, w, dev)
    exact, _ = _plan(sig, xy, col, h, w, dmax, dev, cutoff=104.0)
    ref = _image(exact, h, w, dev)
    # eps * max|colour| (= 1 here), + the fp32 noise of summing up to N terms in a different grouping (which Gaussians take
    # the dmax test, and are summed second, depends on the cutoff): sqrt(N) roundings of the partial sum, 5e-6 of the value
    err = np.abs(img - ref)
    assert (err <= EPS * 1.002 + 5e-6 * np.abs(ref)).all(), float((err - 5e-6 * np.abs(ref)).max())
    # and the bound is not vacuous on this input: terms ARE skipped
    assert err.max() > 0.0


@pytest.mark.parametrize("kernel", ["gaussian", "tile", "atomic"])
def test_gaussians_out_of_reach_under_the_smaller_cutoff(kernel, dev):
    """Centres left of / above the image at a distance between the reach under tau' and the reach under ln(N / eps):
    k_classify keeps them, their tau' window holds no pixel.  They must still be rendered (their conservative window) and get
    their gradient -- which is tiny but written -- from every backward kernel; off-image ones beyond both stay exactly zero."""
    from gsasr_amd import _cabi
    from oracle import gs_oracle
    sig, xy, col, H, W = _synth(48, 48, 4.0, 21)
    n = sig.shape[0]
    plan, _ = _plan(sig, xy, col, H, W, 0.3, dev)
    tau, k = _cabi.plan_cutoff(plan)
    tau_n = _cabi.resolve_cutoff(0.0, n + 64)
    assert tau < tau_n - 1.0
    # 64 extra Gaussians: sigma 3 px, centre d px outside the left / top edge with sqrt(2 tau') * 3 < d < sqrt(2 tau_n) * 3
    hx = 0.5 * (W - 1)
    k_lo, k_hi = math.sqrt(2.0 * (tau + 0.3)), math.sqrt(2.0 * tau_n)
    extra = 64
    d = (k_lo + (k_hi - k_lo) * (np.arange(extra) + 0.5) / extra) * 3.0 - 0.25        # px beyond the edge
    s2, x2, c2 = np.zeros((extra, 3), np.float32), np.zeros((extra, 2), np.float32), np.ones((extra, 3), np.float32)
    s2[:, 0] = s2[:, 1] = 3.0 / hx
    s2[:, 2] = 0.3
    x2[: extra // 2, 0] = -1.0 - d[: extra // 2] / hx
    x2[: extra // 2, 1] = np.linspace(-0.8, 0.8, extra // 2)
    x2[extra // 2:, 1] = -1.0 - d[extra // 2:] / hx
    x2[extra // 2:, 0] = np.linspace(-0.8, 0.8, extra // 2)
    sig, xy, col = np.concatenate([sig, s2]), np.concatenate([xy, x2]), np.concatenate([col, c2])
    wgt = np.random.default_rng(2).random((H, W, 3)).astype(np.float32)
    flag = {"tile": _cabi.FLAG_BWD_TILE, "atomic": _cabi.FLAG_BWD_ATOMIC, "gaussian": _cabi.FLAG_BWD_GAUSSIAN}[kernel]
    plan, (a, b, c) = _plan(sig, xy, col, H, W, 0.3, dev, flags=flag)
    img = _image(plan, H, W, dev)
    ref = gs_oracle.forward_f64(sig, xy, col, H, W, 0.3)
    assert np.abs(img - ref).max() <= 1e-4
    g = [torch.full_like(t, float("nan")) for t in (a, b, c)]
    _cabi.backward(plan, a, b, c, _t(wgt, dev), *g, overwrite=True)
    torch.cuda.synchronize()
    want = gs_oracle.backward_f64(sig, xy, col, wgt, 0.3)
    for got, ref_g, name in zip(g, want, ("sigmas", "coords", "colors")):
        got = got.cpu().numpy()
        assert np.isfinite(got).all(), name
        assert np.abs(got - ref_g).max() <= 2e-4 * np.abs(ref_g).max(), name
        # the extra ones: their whole gradient is a tail term (< exp(-tau') of anything), written, not NaN, ~0
        assert np.abs(got[n:]).max() <= 1e-4 * np.abs(ref_g).max() + 1e-6, name
