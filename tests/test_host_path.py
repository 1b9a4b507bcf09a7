"""Host path of the fused step: no per-call synchronisation, the planar gradient handed to the C
call as it is, and the whole forward + backward capturable in a hipGraph."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_scale_modify_check_is_deferred_not_dropped(dev):
    """a CUDA `scale_modify` with unequal entries still raises the reference's AssertionError (:169) -- at a later
    call or at flush(), not by draining the pipeline at every call"""
    from gsasr_amd import gaussian_splatting as gsp, synthetic
    p = synthetic.gs_parameters(8, 8, seed=1).to(dev)
    gsp.deferred_asserts.flush()
    out = gsp.generate_2D_gaussian_splatting_step((32, 32), p, 4.0, torch.tensor([4.0, 4.0], device=dev), dmax=0.3)
    gsp.deferred_asserts.flush()                                   # equal entries: nothing raised
    assert out.shape == (3, 32, 32)
    gsp.generate_2D_gaussian_splatting_step((32, 32), p, 4.0, torch.tensor([4.0, 3.0], device=dev), dmax=0.3)
    with pytest.raises(AssertionError, match="scale_modify is not the same"):
        gsp.deferred_asserts.flush()
    assert not gsp.deferred_asserts.pending
    with pytest.raises(AssertionError, match="scale_modify is not the same"):   # CPU tensors / numbers: on the spot
        gsp.generate_2D_gaussian_splatting_step((32, 32), p, 4.0, torch.tensor([4.0, 3.0]), dmax=0.3)


@pytest.mark.parametrize("kernel", ["gaussian", "tile"])
def test_fused_step_forward_backward_in_a_hipgraph(kernel, dev):
    """the step is a fixed sequence of launches on the current stream with no host synchronisation: forward AND
    backward capture into one graph, and replaying it on new parameter values gives the eager result"""
    from gsasr_amd import gaussian_splatting as gsp, synthetic
    old = gsp.BACKWARD_KERNEL
    gsp.BACKWARD_KERNEL = kernel
    try:
        H, W = 96, 80
        static_p = synthetic.gs_parameters(24, 20, seed=2).to(dev).requires_grad_(True)
        wgt = synthetic.grad_image(H, W, 3).permute(2, 0, 1).contiguous().to(dev)

        def step():
            out = gsp.generate_2D_gaussian_splatting_step((H, W), static_p, 4.0, (4.0, 4.0), dmax=0.3)
            g, = torch.autograd.grad(out, static_p, wgt)
            return out, g

        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out_g, grad_g = step()
        new = synthetic.gs_parameters(24, 20, seed=7).to(dev)
        with torch.no_grad():
            static_p.copy_(new)
        graph.replay()
        torch.cuda.synchronize()
        ref_p = new.clone().requires_grad_(True)
        ref = gsp.generate_2D_gaussian_splatting_step((H, W), ref_p, 4.0, (4.0, 4.0), dmax=0.3)
        ref.backward(wgt)
        assert float((out_g - ref).abs().max()) <= 1e-5
        assert float((grad_g - ref_p.grad).abs().max()) <= 2e-4 * float(ref_p.grad.abs().max())
    finally:
        gsp.BACKWARD_KERNEL = old


def test_forward_only_plan_under_no_grad(dev):
    """inference (no_grad): the plan is made without backward scratch (GSASR_FLAG_FORWARD_ONLY)"""
    from gsasr_amd import _cabi, gaussian_splatting as gsp, synthetic
    p = synthetic.gs_parameters(16, 16, seed=4).to(dev)
    with torch.no_grad():
        a = gsp.generate_2D_gaussian_splatting_step((64, 64), p, 4.0, (4.0, 4.0), dmax=0.3)
    b = gsp.generate_2D_gaussian_splatting_step((64, 64), p.clone().requires_grad_(True), 4.0, (4.0, 4.0), dmax=0.3)
    assert float((a - b).abs().max()) <= 1e-5
    import ctypes
    fwd = _cabi.make_dims(256, 64, 64, 0.3, flags=_cabi.FLAG_FORWARD_ONLY | _cabi.FLAG_CHW_GRAD | _cabi.FLAG_BWD_TILE)
    full = _cabi.make_dims(256, 64, 64, 0.3, flags=_cabi.FLAG_CHW_GRAD | _cabi.FLAG_BWD_TILE)
    L = _cabi.lib()
    assert L.gsasr_step_workspace_bytes(ctypes.byref(fwd)) < L.gsasr_step_workspace_bytes(ctypes.byref(full))


def test_pooled_workspaces_are_not_shared_between_layouts(dev):
    """two grids whose workspaces have the same byte count (64x64 and 48x64 with the same N) but different cell
    counts: a workspace that comes back from one must not be handed to the other as 'counters clean'"""
    import ctypes
    from gsasr_amd import _cabi, gaussian_splatting as gsp, synthetic
    da, db = _cabi.make_dims(256, 64, 64, 0.3), _cabi.make_dims(256, 48, 64, 0.3)
    L = _cabi.lib()
    assert L.gsasr_splat_workspace_bytes(ctypes.byref(da)) == L.gsasr_splat_workspace_bytes(ctypes.byref(db))
    p = synthetic.gs_parameters(16, 16, seed=9).to(dev)
    ref = {}
    for it in range(6):
        for (H, W) in ((48, 64), (64, 64)):
            with torch.no_grad():
                out = gsp.generate_2D_gaussian_splatting_step((H, W), p, 4.0, (4.0, 4.0), dmax=0.3)
            sig, xy, col, _, _ = gsp._to_kernel_frame(*gsp._activate(p), (H, W), 1.2 / 4.0)
            plan = _cabi.plan(sig, xy, col, H, W, 0.3)                 # the plan-API pool as well
            img = torch.empty(H, W, 3, device=dev)
            _cabi.forward(plan, img, overwrite=True)
            del plan
            if (H, W) not in ref:
                ref[(H, W)] = out.clone()
            assert float((out - ref[(H, W)]).abs().max()) <= 1e-5
            assert float((img.permute(2, 0, 1) - ref[(H, W)]).abs().max()) <= 1e-5


def _host_prologue_reference(p, H, W, scale_modify, dmax, wgt):
    """the unfused torch path of this package (activations + kernel frame as torch ops, reference :174-180, :121-123)
    with the reference's own step expression on a CUDA tensor"""
    from gsasr_amd import gaussian_splatting as gsp
    pa = p.detach().clone().requires_grad_(True)
    a5 = gsp._activate(pa)
    out = gsp.rendering_cuda_dmax(*a5, (H, W), 1.2 / scale_modify[0], p.device, dmax=dmax)
    out.backward(wgt)
    return out.detach(), pa.grad


@pytest.mark.parametrize("kernel", ["gaussian", "tile"])
def test_scale_modify_read_on_the_device_single_and_batched(kernel, dev):
    """a float32 CUDA `scale_modify` goes to the plan's first kernel as a pointer
    (gsasr_step_forward_sm): step = default_step_size / scale_modify[0] is formed there (torch's reciprocal-then-multiply,
    bit for bit) and kept in the workspace for the backward.  Equal (to summation-order noise) to the same fused kernels fed the step as a
    torch tensor, and within rounding of the unfused torch prologue -- for one image (a row view of a [B,2] tensor,
    gsasr_model.py:202) and for the batched canvas ([B,2] tensor and list of rows)."""
    from gsasr_amd import gaussian_splatting as gsp, synthetic
    old = gsp.BACKWARD_KERNEL
    gsp.BACKWARD_KERNEL = kernel
    try:
        H, W = 96, 80
        p = synthetic.gs_parameters(24, 20, seed=2).to(dev)
        wgt = synthetic.grad_image(H, W, 3).permute(2, 0, 1).contiguous().to(dev)
        sms = torch.tensor([[3.0, 3.0], [4.0, 4.0], [2.7, 2.7]], device=dev)
        for b in range(3):
            pa = p.clone().requires_grad_(True)
            out = gsp.generate_2D_gaussian_splatting_step((H, W), pa, 4.0, sms[b], dmax=0.3, default_step_size=1.2)
            out.backward(wgt)
            pt = p.clone().requires_grad_(True)      # the same kernels, the step evaluated by torch
            ref = gsp._FusedStep.apply(pt, (1.2 / sms[b][0]).reshape(1), H, W, 0.3)
            ref.backward(wgt)
            # (not bit-identical run to run: the order of a cell's Gaussians comes from atomics; summation-order noise only)
            assert float((out - ref).abs().max()) <= 2e-6 and float((pa.grad - pt.grad).abs().max()) <= 2e-6 * float(pt.grad.abs().max())
            want, gwant = _host_prologue_reference(p, H, W, sms[b], 0.3, wgt)
            assert float((out - want).abs().max()) <= 1e-5
            assert float((pa.grad - gwant).abs().max()) <= 1e-4 * float(gwant.abs().max())
        # batched canvas: the [B,2] tensor itself, and a list of its rows
        B = 3
        pb = torch.stack([synthetic.gs_parameters(24, 20, seed=2 + i) for i in range(B)]).to(dev)
        for sm_arg in (sms, [sms[i] for i in range(B)]):
            pa = pb.clone().requires_grad_(True)
            out = gsp.generate_2D_gaussian_splatting_batch([(H, W)] * B, pa, [4.0] * B, sm_arg, dmax=0.3)
            out.backward(wgt.expand(B, -1, -1, -1).contiguous())
            for i in range(B):
                want, gwant = _host_prologue_reference(pb[i], H, W, sms[i], 0.3, wgt)
                assert float((out[i] - want).abs().max()) <= 1e-5
                assert float((pa.grad[i] - gwant).abs().max()) <= 1e-4 * float(gwant.abs().max())
        gsp.deferred_asserts.flush()
    finally:
        gsp.BACKWARD_KERNEL = old


def test_scale_modify_mismatch_on_the_device_is_reported(dev):
    """the reference's `assert scale_modify[0] == scale_modify[1]` (:169) for the on-device path: the kernel sets a
    sticky word, the host reads it every few calls / at flush() and raises with the sample and the value"""
    from gsasr_amd import _cabi, gaussian_splatting as gsp, synthetic
    p = synthetic.gs_parameters(8, 8, seed=1).to(dev)
    gsp.deferred_asserts.flush()
    assert int(_cabi.mismatch_flag(dev)[0]) == 0
    pb = torch.stack([p, p, p])
    sms = torch.tensor([[4.0, 4.0], [4.0, 4.0], [4.0, 3.5]], device=dev)
    gsp.generate_2D_gaussian_splatting_batch([(32, 32)] * 3, pb, [4.0] * 3, sms, dmax=0.3)
    with pytest.raises(AssertionError, match=r"scale_modify is not the same \(sample 2 .*4\.0"):
        gsp.deferred_asserts.flush()
    gsp.deferred_asserts.flush()                       # the word was re-armed: nothing pending, nothing raised
    assert int(_cabi.mismatch_flag(dev)[0]) == 0
    # it is polled without anyone calling flush(): WATCH_EVERY calls later the error surfaces at a call site
    gsp.generate_2D_gaussian_splatting_step((32, 32), p, 4.0, torch.tensor([2.0, 3.0], device=dev), dmax=0.3)
    ok = torch.tensor([4.0, 4.0], device=dev)
    with pytest.raises(AssertionError, match="scale_modify is not the same"):
        for _ in range(3 * gsp.deferred_asserts.WATCH_EVERY):
            gsp.generate_2D_gaussian_splatting_step((32, 32), p, 4.0, ok, dmax=0.3)
            torch.cuda.synchronize()
    gsp.deferred_asserts.flush()


def test_backward_kernel_flag_that_disagrees_with_the_plan(dev):
    """whether a workspace carries slots is the PLAN's decision.  A backward that asks for the tile-stationary
    kernel on a plan made without slots -- in a workspace large enough for a plan WITH slots, so that no size check can
    catch it -- must not read slots and spans nobody wrote: it runs another kernel and returns the right gradient; and
    the other way round (slots planned, Gaussian-stationary asked)."""
    import ctypes
    from gsasr_amd import _cabi, synthetic
    from oracle import gs_oracle
    sig, xy, col, H, W = synthetic.kernel_inputs(20, 20, 4.0, seed=4)
    wgt = synthetic.grad_image(H, W, 5)
    want = gs_oracle.backward_f64(sig.numpy(), xy.numpy(), col.numpy(), wgt.numpy(), 0.3)
    a, b, c, wg = (t.to(dev) for t in (sig, xy, col, wgt))
    L = _cabi.lib()
    big = _cabi.make_dims(sig.shape[0], H, W, 0.3, flags=_cabi.FLAG_BWD_TILE)
    ws = torch.full((L.gsasr_splat_workspace_bytes(ctypes.byref(big)) + 4096,), 0xFF, dtype=torch.uint8, device=dev)   # garbage, not zeros
    for plan_flags, bwd_flags in ((_cabi.FLAG_BWD_GAUSSIAN, _cabi.FLAG_BWD_TILE), (0, _cabi.FLAG_BWD_TILE),
                                  (_cabi.FLAG_BWD_TILE, _cabi.FLAG_BWD_GAUSSIAN), (_cabi.FLAG_BWD_TILE, 0)):
        ws.fill_(0xFF)
        dp = _cabi.make_dims(sig.shape[0], H, W, 0.3, flags=plan_flags)
        _cabi.check(L.gsasr_splat_plan(a.data_ptr(), b.data_ptr(), c.data_ptr(), ctypes.byref(dp), ws.data_ptr(), ws.numel(),
                                       _cabi._stream(dev)), "plan")
        db = _cabi.make_dims(sig.shape[0], H, W, 0.3, flags=bwd_flags | _cabi.FLAG_OVERWRITE_GRADS)
        g = [torch.full_like(t, float("nan")) for t in (a, b, c)]
        _cabi.check(L.gsasr_splat_backward(a.data_ptr(), b.data_ptr(), c.data_ptr(), wg.data_ptr(), g[0].data_ptr(), g[1].data_ptr(),
                                           g[2].data_ptr(), ctypes.byref(db), ws.data_ptr(), ws.numel(), _cabi._stream(dev)), "backward")
        for got, w_ in zip(g, want):
            got = got.cpu().numpy()
            assert np.isfinite(got).all() and np.abs(got - w_).max() <= 2e-4 * np.abs(w_).max(), (plan_flags, bwd_flags)


def test_grad_rows_is_validated(dev):
    """a planar batched gradient whose planes are shorter than a sample is an error, not an out-of-bounds read;
    grad_rows on a single image is rejected"""
    import ctypes
    from gsasr_amd import _cabi, synthetic
    B, H, W = 2, 48, 40
    p = torch.stack([synthetic.gs_parameters(12, 10, seed=i) for i in range(B)]).to(dev)
    steps = torch.full((B,), 0.3, device=dev)
    img, plan = _cabi.batch_forward(p, steps, [(H, W), (H - 8, W)], 0.3, _cabi.FLAG_CHW_GRAD | _cabi.FLAG_BWD_TILE)
    good = torch.rand(B, 3, H, W, device=dev)
    assert _cabi.batch_backward(plan, p, steps, good, chw=True).shape == p.shape
    with pytest.raises(RuntimeError, match="rows per plane"):
        _cabi.batch_backward(plan, p, steps, torch.rand(B, 3, H - 4, W, device=dev), chw=True)
    d = _cabi.Dims.from_buffer_copy(plan.dims)
    d._keepalive = plan.dims._keepalive
    d.flags |= _cabi.FLAG_CHW_GRAD
    d.grad_rows = H - 4
    gp = torch.empty_like(p)
    rc = _cabi.lib().gsasr_step_backward(p.data_ptr(), steps.data_ptr(), good.data_ptr(), gp.data_ptr(), ctypes.byref(d),
                                         plan.workspace.data_ptr(), plan.workspace.numel(), _cabi._stream(dev))
    assert rc == -1 and b"grad_rows" in _cabi.lib().gsasr_last_error()
    one = _cabi.make_dims(10, 32, 32, 0.3)
    one.grad_rows = 7
    assert _cabi.lib().gsasr_splat_workspace_bytes(ctypes.byref(one)) > 0     # (sizes do not depend on it)
    ws = torch.empty(_cabi.lib().gsasr_splat_workspace_bytes(ctypes.byref(one)), dtype=torch.uint8, device=dev)
    z = torch.zeros(10, 3, device=dev)
    rc = _cabi.lib().gsasr_splat_plan(z.data_ptr(), z.data_ptr(), z.data_ptr(), ctypes.byref(one), ws.data_ptr(), ws.numel(), _cabi._stream(dev))
    assert rc == -1


def test_expanded_and_foreign_scale_modifies_take_the_evaluated_path(dev):
    """a `[B,2]` scale_modifies made by `.expand(B, 2)` has row stride 0: it must not reach the fused entry
    point as a pointer (the C check `stride >= 2` would refuse it) but render through the evaluated step sizes, with the
    same result as a contiguous copy."""
    from gsasr_amd import gaussian_splatting as gsp, synthetic
    B, H, W = 3, 64, 48
    pb = torch.stack([synthetic.gs_parameters(16, 12, seed=5 + i) for i in range(B)]).to(dev)
    sm_row = torch.tensor([4.0, 4.0], device=dev)
    expanded = sm_row.expand(B, 2)
    assert expanded.stride(0) == 0
    a = gsp.generate_2D_gaussian_splatting_batch([(H, W)] * B, pb, [4.0] * B, expanded, dmax=0.3)
    b = gsp.generate_2D_gaussian_splatting_batch([(H, W)] * B, pb, [4.0] * B, expanded.contiguous(), dmax=0.3)
    gsp.deferred_asserts.flush()
    assert float((a - b).abs().max()) <= 2e-6


def test_scale_modify_mismatch_surfaces_on_an_early_call(dev):
    """the sticky word of the fused path is looked at after the 1st, 2nd, 4th ... fused call on a device, not only
    every WATCH_EVERY calls: a pair that is wrong from the start fails within a handful of calls."""
    from gsasr_amd import gaussian_splatting as gsp, synthetic
    gsp.deferred_asserts.flush()
    gsp.deferred_asserts.seen.pop(dev, None)
    gsp.deferred_asserts.watched.pop(dev, None)
    p = synthetic.gs_parameters(8, 8, seed=1).to(dev)
    bad = torch.tensor([2.0, 3.0], device=dev)
    with pytest.raises(AssertionError, match="scale_modify is not the same"):
        for _ in range(5):
            gsp.generate_2D_gaussian_splatting_step((32, 32), p, 4.0, bad, dmax=0.3)
            torch.cuda.synchronize()
    gsp.deferred_asserts.flush()


def test_mismatch_left_pending_at_exit_fails_the_process(dev):
    """a short script with a mismatched pair and no flush() must not exit with status 0"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import torch\n"
            "from gsasr_amd import gaussian_splatting as gsp, synthetic\n"
            "gsp._DeferredAsserts.WATCH_EVERY = 1 << 30\n"
            "d = torch.device('cuda:0')\n"
            "p = synthetic.gs_parameters(8, 8, seed=1).to(d)\n"
            "ok, bad = torch.tensor([4.0, 4.0], device=d), torch.tensor([2.0, 3.0], device=d)\n"
            "gsp.generate_2D_gaussian_splatting_step((32, 32), p, 4.0, ok, dmax=0.3)\n"
            "gsp.generate_2D_gaussian_splatting_step((32, 32), p, 4.0, ok, dmax=0.3)\n"
            "gsp.generate_2D_gaussian_splatting_step((32, 32), p, 4.0, bad, dmax=0.3)\n"      # third call: not an early look
            "torch.cuda.synchronize(); print('rendered')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "rendered" in r.stdout
    assert r.returncode == 1 and "deferred check failed at exit" in r.stderr, (r.returncode, r.stderr[-500:])


def test_reference_shaped_launchers_reuse_their_scratch(dev):
    """module `gscuda` (the pybind surface, gswrapper.cpp:9-73 -> gsasr_gs_render*) keeps its plan
    scratch per stream between calls; alternating shapes, repeated calls and a release in between all give the oracle's
    numbers (the counters-clean / parity bookkeeping of a reused workspace must survive a shape change)."""
    import numpy as np
    from gsasr_amd import _cabi, gscuda, synthetic
    from oracle import gs_oracle
    cases = []
    for (hl, wl, sc, seed) in ((12, 10, 4.0, 1), (20, 16, 3.0, 2)):
        sig, xy, col, H, W = synthetic.kernel_inputs(hl, wl, sc, seed=seed)
        ref = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, 0.3)
        wgt = synthetic.grad_image(H, W, seed)
        gref = gs_oracle.backward_f64(sig.numpy(), xy.numpy(), col.numpy(), wgt.numpy(), 0.3)
        cases.append((sig.to(dev), xy.to(dev), col.to(dev), H, W, ref, wgt.to(dev), gref))
    for rep in range(6):
        sig, xy, col, H, W, ref, wgt, gref = cases[rep % 2 if rep < 4 else 0]
        img = torch.zeros(H, W, 3, device=dev)
        gscuda.gs_render(sig, xy, col, img, sig.shape[0], H, W, 3, 0.3)
        g = [torch.zeros_like(t) for t in (sig, xy, col)]
        gscuda.gs_render_backward(sig, xy, col, wgt, *g, sig.shape[0], H, W, 3, 0.3)
        torch.cuda.synchronize()
        assert np.abs(img.cpu().numpy() - ref).max() <= 1e-4
        for got, want in zip(g, gref):
            assert np.abs(got.cpu().numpy() - want).max() <= 2e-4 * np.abs(want).max()
        if rep == 3:
            _cabi.clear_workspace_pool()      # frees the launchers' scratch as well: the next call allocates again


def test_cpp_autograd_node_is_loaded_and_equals_the_python_node(dev):
    """`GSCUDA.apply` runs as a C++ torch::autograd::Function (gsasr_amd/csrc/gsasr_autograd.cpp) on the GPU
    box; same image and gradients as the Python Function (`GSCUDA.python_apply`), for both ops, against the oracle; a graph
    kept with retain_graph runs its backward twice on the same plan; autocast inputs are cast; errors stay RuntimeErrors."""
    import numpy as np
    from gsasr_amd import _cpp_node, synthetic
    from gsasr_amd.gs_cuda.gswrapper import GSCUDA as G0
    from gsasr_amd.gs_cuda_dmax.gswrapper import GSCUDA as G1
    from oracle import gs_oracle
    assert _cpp_node.load() is not None, "gsasr_amd/lib/_gsasr_autograd.so not built / not loadable on the GPU box"
    sig, xy, col, H, W = synthetic.kernel_inputs(20, 16, 4.0, seed=4)
    wgt = synthetic.grad_image(H, W, 5).to(dev)
    for dmax in (0.3, None):
        G = G0 if dmax is None else G1
        args = () if dmax is None else (dmax,)
        out = {}
        for name, fn in (("cpp", G.apply), ("python", G.python_apply)):
            a, b, c = (t.to(dev).requires_grad_(True) for t in (sig, xy, col))
            img = fn(a, b, c, torch.zeros(H, W, 3, device=dev), *args)
            img.backward(wgt, retain_graph=(name == "cpp"))
            first = [t.grad.clone() for t in (a, b, c)]
            if name == "cpp":       # the node still owns its plan: a second backward gives the same gradient again
                for t in (a, b, c):
                    t.grad = None
                img.backward(wgt)
                for g1, t in zip(first, (a, b, c)):
                    assert torch.equal(g1, t.grad)
            out[name] = (img.detach().cpu().numpy(), [g.cpu().numpy() for g in first])
        assert np.abs(out["cpp"][0] - out["python"][0]).max() <= 2e-6
        ref = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, dmax)
        assert np.abs(out["cpp"][0] - ref).max() <= 1e-4
        gref = gs_oracle.backward_f64(sig.numpy(), xy.numpy(), col.numpy(), wgt.cpu().numpy(), dmax)
        for got, py, want in zip(out["cpp"][1], out["python"][1], gref):
            assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max()
            assert np.abs(got - py).max() <= 1e-5 * np.abs(want).max()
    # the reference's contract: the tensor handed in is accumulated into and returned
    a, b, c = (t.to(dev) for t in (sig, xy, col))
    canvas = torch.ones(H, W, 3, device=dev)
    with torch.no_grad():
        back = G1.apply(a, b, c, canvas, 0.3)
    assert back.data_ptr() == canvas.data_ptr()
    assert float((canvas - 1.0 - torch.from_numpy(gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, 0.3)).to(dev)).abs().max()) <= 1e-4
    with torch.autocast("cuda", dtype=torch.bfloat16):
        img = G1.apply(a.bfloat16(), b, c, torch.zeros(H, W, 3, device=dev), 0.3)
    assert img.dtype == torch.float32 and torch.isfinite(img).all()
    with pytest.raises(RuntimeError):
        G1.apply(a.t().contiguous().t(), b, c, torch.zeros(H, W, 3, device=dev), 0.3)      # non-contiguous sigmas
    with pytest.raises(RuntimeError):
        G1.apply(a, b, c, torch.zeros(H, W, 4, device=dev), 0.3)


def test_scale_hint_picks_the_wide_forward_without_changing_the_numbers(dev):
    """gaussian_splatting._forward_flag: from x5 up on >= 2 Mpx the fused path asks for the wide forward whatever the Gaussians per LR
    pixel (the library's pixels-per-Gaussian rule reads x8 at 16 per LR pixel as small windows); same image and gradient either way"""
    from gsasr_amd import _cabi, gaussian_splatting as gsp, synthetic
    assert gsp._forward_flag(8, 2048, 2048) == _cabi.FLAG_FWD_WIDE and gsp._forward_flag(8.0, 1024, 1024) == 0
    assert gsp._forward_flag(4, 4096, 4096) == 0 and gsp._forward_flag(torch.tensor(8.0), 4096, 4096) == 0
    p = synthetic.gs_parameters(184, 184, seed=41, gpp=4).to(dev)           # x8 -> 1472^2 = 2.17 Mpx, 4 Gaussians per LR pixel
    sm = torch.tensor([8.0, 8.0])
    out = {}
    old = gsp.SCALE_HINT
    try:
        for hint in (True, False):
            gsp.SCALE_HINT = hint
            pa = p.clone().requires_grad_(True)
            img = gsp.generate_2D_gaussian_splatting_step((1472, 1472), pa, 8, sm, dmax=0.1)
            img.square().sum().backward()
            out[hint] = (img.detach(), pa.grad)
    finally:
        gsp.SCALE_HINT = old
    for a, b in zip(out[True], out[False]):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
