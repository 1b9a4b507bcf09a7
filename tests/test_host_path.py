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
