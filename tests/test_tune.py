"""Kernel choices registered per shape (include/gsasr_splat.h: gsasr_set_kernel_choice) and gsasr_amd/tune.py."""
import ctypes

import pytest
import torch

from gsasr_amd import _cabi, synthetic, tune


@pytest.fixture(autouse=True)
def _clean():
    tune.reset()
    yield
    tune.reset()


def _bytes(d):
    return _cabi.lib().gsasr_splat_workspace_bytes(ctypes.byref(d))


def test_registry_follows_the_shape_and_sizes_the_workspace():
    d = _cabi.make_dims(4096, 256, 256, 0.1)
    base = _bytes(d)
    assert _cabi.get_kernel_choice(d) is None
    _cabi.set_kernel_choice(d, _cabi.FLAG_BWD_TILE, 0)
    assert _cabi.get_kernel_choice(d) == (_cabi.FLAG_BWD_TILE, 0)
    with_slots = _bytes(d)
    assert with_slots >= base + 4096 * 32 * 8            # slots of the tile-stationary backward
    # explicit flags of a call win over the registration
    assert _bytes(_cabi.make_dims(4096, 256, 256, 0.1, flags=_cabi.FLAG_BWD_GAUSSIAN)) == base
    # the registration is for THIS shape: another dmax, a row band, a forward-only plan, another N are not affected
    for other in (_cabi.make_dims(4096, 256, 256, 0.5), _cabi.make_dims(4096, 256, 256, 0.1, rows=(0, 128)),
                  _cabi.make_dims(4096, 256, 256, 0.1, flags=_cabi.FLAG_FORWARD_ONLY), _cabi.make_dims(4097, 256, 256, 0.1)):
        assert _cabi.get_kernel_choice(other) is None
    # lists: registered capacity, registered "none"
    _cabi.set_kernel_choice(d, 0, 512)
    assert _bytes(d) > base and _bytes(d) != with_slots
    _cabi.set_kernel_choice(d, 0, -1)
    assert _bytes(d) == _bytes(_cabi.make_dims(4096, 256, 256, 0.1, list_cap=-1))
    _cabi.clear_kernel_choices()
    assert _cabi.get_kernel_choice(d) is None and _bytes(d) == base


def test_registry_refuses_other_flags():
    d = _cabi.make_dims(64, 32, 32, 0.1)
    for bad in (_cabi.FLAG_OVERWRITE_IMAGE, _cabi.FLAG_FWD_WIDE | _cabi.FLAG_FWD_NARROW, _cabi.FLAG_BWD_TILE | _cabi.FLAG_BWD_GAUSSIAN,
                _cabi.FLAG_BWD_ATOMIC):
        with pytest.raises(RuntimeError):
            _cabi.set_kernel_choice(d, bad, 0)
    assert _cabi.get_kernel_choice(d) is None


def test_registry_full_table_replaces_its_oldest_entry():
    """300 shapes: no error (round 5 raised from the 257th on -- inside the user's forward under GSASR_AMD_AUTOTUNE);
    the oldest registrations are gone, the newest are in force, and "no choice" for an unknown shape stores nothing"""
    for k in range(300):
        _cabi.set_kernel_choice(_cabi.make_dims(1000 + k, 64, 64, 0.1), _cabi.FLAG_BWD_TILE, 0)
    assert _cabi.get_kernel_choice(_cabi.make_dims(1299, 64, 64, 0.1)) == (_cabi.FLAG_BWD_TILE, 0)
    assert _cabi.get_kernel_choice(_cabi.make_dims(1000, 64, 64, 0.1)) is None
    _cabi.set_kernel_choice(_cabi.make_dims(5000, 64, 64, 0.1), 0, 0)
    assert _cabi.get_kernel_choice(_cabi.make_dims(5000, 64, 64, 0.1)) is None


@pytest.mark.gpu
def test_a_plan_keeps_the_list_capacity_it_was_made_with():
    """the capacity of a plan's tile lists (the stride of its entries) comes from the registry at PLAN time; a
    registration changed or cleared between the plan and its forward / backward must not change how they read the workspace"""
    dev = torch.device("cuda:0")
    sig, xy, col, H, W = synthetic.kernel_inputs(32, 32, 4, seed=11, gpp=16, device=dev)
    g = synthetic.grad_image(H, W, seed=12, device=dev)
    ref = _grads(sig, xy, col, H, W, 0.1, g)
    shape = _cabi.make_dims(sig.shape[0], H, W, 0.1)
    for first, then in ((1024, 4096), (4096, 256), (2048, None)):
        _cabi.set_kernel_choice(shape, _cabi.FLAG_BWD_TILE, first)
        p = _cabi.plan(sig, xy, col, H, W, 0.1)
        if then is None:
            _cabi.clear_kernel_choices()
        else:
            _cabi.set_kernel_choice(shape, _cabi.FLAG_BWD_TILE, then)      # another stride for plans made from now on
        img = torch.empty(H, W, 3, device=dev)
        _cabi.forward(p, img, overwrite=True)
        out = (img, *_cabi.backward_new(p, sig, xy, col, g))
        for a, b, what in zip(out, ref, ("image", "d sigmas", "d coords", "d colors")):
            top = float(b.abs().max())
            assert float((a - b).abs().max()) <= 2e-5 * max(top, 1.0), (first, then, what)


def test_candidates_cover_every_combination():
    names = [n for n, _, _ in tune.candidates(65536, 1024, 1024, True)]
    assert names[0] == "default" and len(names) == 13 and len(set(names)) == 13      # {narrow, wide} x {gaussian, tile, home} x {lists, search}
    assert len(tune.candidates(65536, 1024, 1024, False)) == 5
    for _, flags, cap in tune.candidates(65536, 1024, 1024, True)[1:]:
        assert flags & ~_cabi.CHOICE_FLAGS == 0 and cap != 0


def _grads(sig, xy, col, H, W, dmax, g):
    p = _cabi.plan(sig, xy, col, H, W, dmax)
    img = torch.empty(H, W, 3, device=sig.device)
    _cabi.forward(p, img, overwrite=True)
    return (img, *_cabi.backward_new(p, sig, xy, col, g))


@pytest.mark.gpu
@pytest.mark.parametrize("lr,scale,gpp", [(64, 4, 1), (32, 4, 16), (48, 8, 1)])
def test_every_registered_combination_computes_the_same_sums(lr, scale, gpp):
    dev = torch.device("cuda:0")
    sig, xy, col, H, W = synthetic.kernel_inputs(lr, lr, scale, seed=3, gpp=gpp, device=dev)
    g = synthetic.grad_image(H, W, seed=4, device=dev)
    ref = _grads(sig, xy, col, H, W, 0.1, g)
    shape = _cabi.make_dims(sig.shape[0], H, W, 0.1)
    base_bytes = _bytes(shape)
    for name, flags, cap in tune.candidates(sig.shape[0], W, H, True)[1:]:
        _cabi.set_kernel_choice(shape, flags, cap)
        out = _grads(sig, xy, col, H, W, 0.1, g)
        for a, b, what in zip(out, ref, ("image", "d sigmas", "d coords", "d colors")):
            top = float(b.abs().max())
            assert float((a - b).abs().max()) <= 2e-5 * max(top, 1.0), (name, what)
    _cabi.clear_kernel_choices()
    assert _bytes(shape) == base_bytes


@pytest.mark.gpu
def test_tune_registers_a_measured_choice_and_the_dropin_follows_it():
    from gsasr_amd.gs_cuda_dmax.gswrapper import GSCUDA
    dev = torch.device("cuda:0")
    sig, xy, col, H, W = synthetic.kernel_inputs(128, 128, 4, seed=5, gpp=4, device=dev)      # 512^2, 4 Gaussians per LR pixel
    g = synthetic.grad_image(H, W, seed=6, device=dev)
    ref = _grads(sig, xy, col, H, W, 0.1, g)
    res = tune.tune(sig, xy, col, H, W, 0.1)
    assert res.registered and "default" in res.ms and len(res.ms) >= 5 and res.ms[res.name] <= res.ms["default"]
    shape = _cabi.make_dims(sig.shape[0], H, W, 0.1)
    # (the library's own rule winning registers nothing: the C table is finite)
    assert _cabi.get_kernel_choice(shape) == (None if res.name == "default" else (res.flags, res.list_cap))
    # the drop-in (whichever node serves it) now runs the registered combination and computes what the default computed
    s2, x2, c2 = (t.clone().requires_grad_(True) for t in (sig, xy, col))
    img = GSCUDA.apply(s2, x2, c2, torch.zeros(H, W, 3, device=dev), 0.1)
    (img * g).sum().backward()
    for a, b, what in zip((img.detach(), s2.grad, x2.grad, c2.grad), ref, ("image", "d sigmas", "d coords", "d colors")):
        top = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * max(top, 1.0), what
    # an inference shape: forward-only plans have their own registration
    res_f = tune.tune(sig, xy, col, H, W, 0.1, backward=False)
    assert _cabi.get_kernel_choice(_cabi.make_dims(sig.shape[0], H, W, 0.1, flags=_cabi.FLAG_FORWARD_ONLY)) == \
        (None if res_f.name == "default" else (res_f.flags, res_f.list_cap))
    assert len(res_f.ms) >= 3


@pytest.mark.gpu
def test_autotune_hook_tunes_a_shape_once(monkeypatch):
    from gsasr_amd import _cpp_node
    from gsasr_amd.gs_cuda_dmax.gswrapper import gaussiansplatting_render
    monkeypatch.setattr(_cabi, "_AUTOTUNE", True)
    monkeypatch.setattr(_cpp_node, "_AUTOTUNE", True)
    dev = torch.device("cuda:0")
    sig, xy, col, H, W = synthetic.kernel_inputs(32, 32, 4, seed=7, gpp=16, device=dev)
    calls = []
    real = tune.tune
    monkeypatch.setattr(tune, "tune", lambda *a, **k: calls.append(1) or real(*a, **k))
    a = gaussiansplatting_render(sig, xy, col, (H, W), 0.1)
    b = gaussiansplatting_render(sig, xy, col, (H, W), 0.1)
    assert len(calls) == 1
    assert tune._shape_key(sig.shape[0], H, W, 0.1, None, 0.0, False) in tune._SEEN
    # (the same combination both times; list order -- hence fp32 summation order -- may differ between two plans)
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())


@pytest.mark.gpu
def test_tune_batch_registers_for_the_canvas_and_the_fused_path_follows(monkeypatch):
    """the batched training step: tune_batch times {backward kernel} x {lists, search} on the batch, registers the winner under the
    canvas shape; gaussian_splatting's fused path reads the registration (backward kernel in Python, lists in the C library) and
    computes the gradients of the untuned path"""
    from gsasr_amd import gaussian_splatting as gsp
    dev = torch.device("cuda:0")
    B, lr, scale = 4, 24, 4.0
    sizes = [(96, 96)] * B
    p = torch.stack([synthetic.gs_parameters(lr, lr, seed=20 + b, gpp=16) for b in range(B)]).to(dev)
    sms = [torch.tensor([scale, scale]) for _ in range(B)]
    wgt = torch.rand(B, 3, 96, 96, generator=torch.Generator().manual_seed(9)).to(dev)

    def grads():
        pa = p.clone().requires_grad_(True)
        out = gsp.generate_2D_gaussian_splatting_batch(sizes, pa, [scale] * B, sms, dmax=0.5)
        (out * wgt).sum().backward()
        return out.detach(), pa.grad

    ref = grads()
    steps = torch.full((B,), 1.2 / scale, device=dev)
    res = tune.tune_batch(p, steps, sizes, 0.5)
    assert res.registered and set(res.ms) >= {"default", "gaussian-search", "tile-search"}
    shape = _cabi.make_batch_dims(p.shape[1], sizes, 96, 96, 0.5)
    assert _cabi.get_kernel_choice(shape) is not None
    # force the combination the rule would NOT pick, through the registry alone, and check the fused path really switches
    seen = []
    real = _cabi.batch_forward
    monkeypatch.setattr(_cabi, "batch_forward", lambda *a, **k: seen.append(a[4] if len(a) > 4 else k.get("extra_flags", 0)) or real(*a, **k))
    from gsasr_amd import _cpp_node
    monkeypatch.setattr(_cpp_node, "load", lambda: None)        # (the Python node: its flags are visible here)
    for flag in (_cabi.FLAG_BWD_TILE, _cabi.FLAG_BWD_GAUSSIAN):
        _cabi.set_kernel_choice(shape, flag, -1)
        out = grads()
        assert seen[-1] & (_cabi.FLAG_BWD_TILE | _cabi.FLAG_BWD_GAUSSIAN) == flag
        for a, b, what in zip(out, ref, ("image", "d gs_parameters")):
            top = float(b.abs().max())
            assert float((a - b).abs().max()) <= 3e-5 * max(top, 1.0), (flag, what)


@pytest.mark.gpu
def test_tune_step_on_a_single_fused_image():
    from gsasr_amd import gaussian_splatting as gsp
    dev = torch.device("cuda:0")
    p = synthetic.gs_parameters(64, 64, seed=31, gpp=4).to(dev)
    step = torch.tensor([0.3], device=dev)
    res = tune.tune_step(p, step, 256, 256, 0.1)
    assert res.registered and "default" in res.ms and res.ms[res.name] <= res.ms["default"]
    pa = p.clone().requires_grad_(True)
    out = gsp.generate_2D_gaussian_splatting_step((256, 256), pa, 4.0, torch.tensor([4.0, 4.0]), dmax=0.1)
    out.sum().backward()
    assert torch.isfinite(pa.grad).all() and float(out.abs().max()) > 0
