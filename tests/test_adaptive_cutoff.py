"""The data-derived support cutoff (include/gsasr_splat.h, `gsasr_plan_cutoff`; adapt_kcut in gsasr_splat.hip).  A term is
skipped outside its Gaussian's window; tau' = ln(K / budget) <= ln(N / 1e-5) keeps the skipped sum on every pixel below
`1e-5 * max|colour|` when K bounds the live Gaussians that can lose a non-negligible term there -- counted by the plan from
its own cell histogram, either as the Gaussians whose dmax box covers the pixel (bounded op, utils/gs_cuda_dmax/gs.cu:41-50)
or as those binned within the class' largest support of it (+ a geometric tail for everything farther; both ops), the
smaller of the two -- and `budget` is what the tails of the dead Gaussians whose box still reaches the rows leave of 1e-5.
Checked here:

  * THE GUARANTEE itself, on every case: the default render stays within 1e-5 * max|colour| of the exact one (tau = 104, the
    reference's set of non-zero fp32 terms);
  * the K the plan derives is an upper bound of the true count (brute force on the CPU: the smaller of "boxes over a pixel"
    and "centres within the largest support of a pixel") and tau' is the logarithm of THAT K -- on GSASR-shaped input it is
    well below ln(N / 1e-5), for the unbounded op too;
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
    if dmax is None:
        return xy.shape[0]
    px = (2.0 * np.arange(w) / (w - 1) - 1.0).astype(np.float32)
    py = (2.0 * np.arange(h) / (h - 1) - 1.0).astype(np.float32)
    inx = (np.abs(px[None, :] - xy[:, 0:1]) <= np.float32(dmax)).astype(np.float32)   # [N, w]
    iny = (np.abs(py[None, :] - xy[:, 1:2]) <= np.float32(dmax)).astype(np.float32)   # [N, h]
    return int((iny.T @ inx).max())


def _true_near(sig, xy, h, w, dmax, tau):
    """max over pixels of the Gaussians centred within E pixels of it in x and in y, E = the largest half-extent any of
    them has under `tau` (box-capped): what the plan's ring count must cover"""
    hx, hy = 0.5 * (w - 1), 0.5 * (h - 1)
    k = math.sqrt(2.0 * tau)
    cap = np.inf if dmax is None else dmax
    ex, ey = np.minimum(cap, k * np.abs(sig[:, 0])) * hx, np.minimum(cap, k * np.abs(sig[:, 1])) * hy
    live = (ex <= 128.0) & (ey <= 128.0)          # (the large class is counted separately, in full)
    e = float(max(ex[live].max(initial=0.0), ey[live].max(initial=0.0)))
    cx, cy = (xy[:, 0].astype(np.float64) + 1.0) * hx, (xy[:, 1].astype(np.float64) + 1.0) * hy
    inx = (np.abs(np.arange(w)[None, :] - cx[live, None]) <= e).astype(np.float32)
    iny = (np.abs(np.arange(h)[None, :] - cy[live, None]) <= e).astype(np.float32)
    return int((iny.T @ inx).max()) + int((~live).sum())


def _within_eps_of_exact(sig, xy, col, h, w, dmax, dev, plan=None):
    """the guarantee: |default render - exact render| <= eps * max|colour| (+ fp32 noise of a differently grouped sum)"""
    if plan is None:
        plan, _ = _plan(sig, xy, col, h, w, dmax, dev)
    img = _image(plan, h, w, dev)
    exact, _ = _plan(sig, xy, col, h, w, dmax, dev, cutoff=104.0)
    ref = _image(exact, h, w, dev)
    err = np.abs(img - ref)
    lim = EPS * 1.002 * max(1.0, float(np.abs(col).max())) + 5e-6 * np.abs(ref)
    assert (err <= lim).all(), float((err - lim).max())
    return float(err.max())


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
                                  (96, 96, 12.0, 1, 0.1), (160, 100, 8.0, 1, 0.2), (64, 64, 4.0, 1, None), (48, 48, 4.0, 16, 0.5)],
                         ids=["x4", "x4-16-per-lr-px", "x8", "config2", "x12-sparse-cells", "x8-two-pass-scan", "x4-unbounded-op",
                              "training-crop-dmax0.5"])
def test_plan_k_bounds_the_true_k_and_sets_tau(case, dev):
    from gsasr_amd import _cabi
    h_lr, w_lr, scale, gpp, dmax = case
    sig, xy, col, H, W = _synth(h_lr, w_lr, scale, 7, gpp)
    plan, _ = _plan(sig, xy, col, H, W, dmax, dev)
    tau, k = _cabi.plan_cutoff(plan)
    n = sig.shape[0]
    tau_n = _cabi.resolve_cutoff(0.0, n)
    k_true = min(_true_k(xy, H, W, dmax), _true_near(sig, xy, H, W, dmax, tau_n))
    assert k >= k_true, (k, k_true)
    assert 16.0 <= tau <= tau_n + 1e-6
    if tau < tau_n - 1e-6 and tau > 16.0:
        # tau is ln(K / budget) of the K reported; no near-dead here, the tail of the ring takes a few % of the budget (more when N is small)
        assert -2e-4 * tau <= tau - (math.log(k / EPS) + 1e-3) <= 0.35, (tau, k)
    if h_lr >= 256:
        # BASELINE config 2: tau well under ln(N / eps) (K is 1 092 cells-worth against a true near count of ~120: cells are 16 px)
        assert tau <= tau_n - 3.0, (k, k_true, tau, tau_n)
    if scale >= 8.0:
        # x8 and up: the dmax box (0.1: 409 px at 8192) is far wider than any support -- the ring count is what binds
        assert tau <= tau_n - 2.0, (tau, tau_n)
    _within_eps_of_exact(sig, xy, col, H, W, dmax, dev, plan)


def test_explicit_cutoff_and_process_default_are_not_touched(dev):
    from gsasr_amd import _cabi
    sig, xy, col, H, W = _synth(32, 32, 4.0, 3)
    for cutoff, dmax, want in ((32.0, 0.1, 32.0), (104.0, 0.1, 104.0), (40.0, None, 40.0)):
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


@pytest.mark.parametrize("dmax", [0.3, None], ids=["bounded", "unbounded"])
def test_near_dead_gaussians_are_charged_to_the_budget(dmax, dev):
    """Gaussians centred just outside the image whose support (under the conservative tau) stops short of the first column:
    k_classify drops them as dead, yet the op adds their tails (each < exp(-tau)) -- terms the windows' cutoff has to pay
    for.  They must be counted (tau' rises by ln(1 / (1 - n_near / s))) and the guarantee must hold with thousands of them
    stacked on one row."""
    from gsasr_amd import _cabi
    sig, xy, col, H, W = _synth(48, 48, 4.0, 31)
    n = sig.shape[0]
    m = n                                   # as many near-dead as live ones: half of the budget goes to their tails
    s_tot = n + m
    tau_c = _cabi.resolve_cutoff(0.0, s_tot)
    hx = 0.5 * (W - 1)
    sp = 3.0                                 # sigma in pixels; support = sqrt(2 tau_c) * 3 px
    d = math.sqrt(2.0 * tau_c) * sp * 1.000002 + 0.05      # px beyond column 0: just out of reach
    s2, x2, c2 = np.zeros((m, 3), np.float32), np.zeros((m, 2), np.float32), np.ones((m, 3), np.float32)
    s2[:, 0] = s2[:, 1] = sp / hx
    x2[:, 0] = -1.0 - d / hx
    x2[:, 1] = 0.1
    plan0, _ = _plan(sig, xy, col, H, W, dmax, dev)
    tau0, k0 = _cabi.plan_cutoff(plan0)
    sig2, xy2, col2 = np.concatenate([sig, s2]), np.concatenate([xy, x2]), np.concatenate([col, c2])
    plan, _ = _plan(sig2, xy2, col2, H, W, dmax, dev)
    tau, k = _cabi.plan_cutoff(plan)
    assert k == k0                                                   # the dead ones sit in no cell
    want = math.log(k / (EPS * (1.0 - m / s_tot))) + 1e-3            # budget = eps - m * exp(-tau_c) = eps * (1 - m / s) - ring tail
    assert -3e-4 * want <= tau - want <= 0.35, (tau, want, tau0)
    err = _within_eps_of_exact(sig2, xy2, col2, H, W, dmax, dev, plan)
    assert err > 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["gaussian", "home"])
@pytest.mark.parametrize("case", ["dense", "stacked"])
def test_backward_window_of_its_own_cutoff_is_complete(case, kernel):
    """Round 6: under the adaptive default the Gaussian-stationary and home-tile backward sweep the window of
    min(tau', GSASR_SPLAT_GRAD_TAU = 16) -- a gradient sums over its own pixels, so its window does not grow with the K of the
    forward's bound (include/gsasr_splat.h).  Against the same backward with the FORWARD's tau' given explicitly (an explicit
    cutoff is used as given by both directions): every Gaussian's row within a fifth of the per-Gaussian parity bar (the tail
    between the two ellipses holds < 2e-6 of a Gaussian's moments), whatever K is -- 16 Gaussians per LR pixel (tau' ~ 20), and every Gaussian
    stacked on one spot (K = N: tau' = the conservative tau)."""
    from gsasr_amd import _cabi, synthetic
    dev = torch.device("cuda:0")
    sig, xy, col, H, W = synthetic.kernel_inputs(48, 48, 4.0, seed=21, gpp=16)
    if case == "stacked":
        xy = xy.clone()
        xy[:] = torch.tensor([0.1, -0.2]) + 0.01 * torch.randn(xy.shape, generator=torch.Generator().manual_seed(3))
    a, b, c = (t.contiguous().to(dev) for t in (sig, xy, col))
    wgt = synthetic.grad_image(H, W, 22).to(dev)
    flag = _cabi.FLAG_BWD_HOME if kernel == "home" else _cabi.FLAG_BWD_GAUSSIAN
    plan = _cabi.plan(a, b, c, H, W, 0.5, flags=flag)
    tau_w, _ = _cabi.plan_cutoff(plan)
    assert tau_w > 18.0                                   # the forward's cutoff is well above the backward's own
    got = _cabi.backward_new(plan, a, b, c, wgt)
    ref = _cabi.backward_new(_cabi.plan(a, b, c, H, W, 0.5, cutoff=tau_w, flags=flag), a, b, c, wgt)
    differs = False
    for g, r, name in zip(got, ref, ("sigmas", "coords", "colors")):
        g, r = g.cpu().numpy(), r.cpu().numpy()
        differs = differs or not np.array_equal(g, r)
        # a fifth of the per-Gaussian parity bar (5e-4 of the row + 1e-5 of the tensor's max-abs, tests/test_bwd_tile.py).  The
        # floor matters: d/dx, d/dy integrate an ODD weight, so a row's signed value can be far below the Gaussian's unsigned
        # mass, which is what the truncation is 1e-5 of
        tol = 1e-4 * np.abs(r).max(axis=1, keepdims=True) + 2e-6 * np.abs(r).max()
        ratio = float((np.abs(g - r) / tol).max())
        assert ratio <= 1.0, (name, ratio, float(np.abs(g - r).max()), float(np.abs(r).max()))
    assert differs                                        # (the smaller window IS in force)
