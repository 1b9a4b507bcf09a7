"""Mirror of the reference's utils/gs_cuda_dmax/gswrapper.py (bounded-window op).

    GSCUDA.apply(sigmas[N,3], coords[N,2], colors[N,3], rendered_img[H,W,3], dmax) -> rendered_img
    gaussiansplatting_render(sigmas, coords, colors, image_size, dmax=100) -> [H,W,3]
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .._amp import fp32_boundary_bwd, fp32_boundary_fwd

from .. import _cabi


class GSCUDA(Function):
    """reference: utils/gs_cuda_dmax/gswrapper.py:22-44"""

    @staticmethod
    @fp32_boundary_fwd
    def forward(ctx, sigmas, coords, colors, rendered_img, dmax):
        ctx.save_for_backward(sigmas, coords, colors)
        ctx.dmax = dmax
        if rendered_img.dim() != 3 or rendered_img.shape[2] != 3:
            raise RuntimeError("rendered_img must be [H,W,3]")
        ctx.plan = _cabi.plan_forward(sigmas, coords, colors, rendered_img, float(dmax))   # plan + splat, one host call
        return rendered_img

    @staticmethod
    @once_differentiable
    @fp32_boundary_bwd
    def backward(ctx, grad_output):
        sigmas, coords, colors = ctx.saved_tensors
        # (the reference zero-fills three tensors and lets the kernel add into them; the backward
        # stores instead, which saves three memsets per step)
        return (*_cabi.backward_new(ctx.plan, sigmas, coords, colors, grad_output), None, None)


# The same node in C++ when the extension has been built (gsasr_amd/_cpp_node.py): the autograd engine then calls the backward
# without taking the GIL.  `GSCUDA.apply` keeps its signature; `GSCUDA.forward` / `.backward` above remain the Python path.
from .. import _cpp_node  # noqa: E402

if _cpp_node.load() is not None:
    GSCUDA.python_apply = GSCUDA.apply
    GSCUDA.apply = staticmethod(lambda sigmas, coords, colors, rendered_img, dmax: _cpp_node.fast_apply(sigmas, coords, colors, rendered_img, dmax))


def gaussiansplatting_render(sigmas, coords, colors, image_size, dmax=100):
    """reference: utils/gs_cuda_dmax/gswrapper.py:46-53"""
    sigmas = sigmas.contiguous()
    coords = coords.contiguous()
    colors = colors.contiguous()
    h, w = image_size[:2]
    c = colors.shape[-1]
    rendered_img = torch.zeros(int(h), int(w), c, device=colors.device, dtype=torch.float32)
    return GSCUDA.apply(sigmas, coords, colors, rendered_img, dmax)
