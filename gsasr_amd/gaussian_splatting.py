"""Rasterizer host API -- mirror of the reference's utils/gaussian_splatting.py (same function names,
arguments, defaults and error behaviour), with the CUDA ops replaced by the HIP rasterizer.

    generate_2D_gaussian_splatting_step(sr_size, gs_parameters, scale, scale_modify, sample_coords=None,
        default_step_size=1.2, cuda_rendering=True, mode='scale_modify', if_dmax=True,
        dmax_mode='fix', dmax=25) -> [3,H,W]                      (reference :158-217)
    generate_2D_gaussian_splatting_step_buffer(..., buffer_size=4000000)   (reference :219-265)
    rendering_cuda / rendering_cuda_dmax / rendering_cuda_buffer / rendering_cuda_dmax_buffer
                                                                  (reference :86-155)
    rendering_python                                               (reference :11-84)

`gs_parameters[N,9]` columns are the decoder's raw `[sigma_x, sigma_y, rho, alpha, r, g, b, mu_x, mu_y]`
(reference utils/fea2gs.py:632-635).  Everything here is ordinary differentiable torch code on the
tensors' own device; the only custom op is `GSCUDA.apply` (gsasr_amd/gs_cuda*/gswrapper.py).
`cuda_rendering=False` selects the reference's pure-PyTorch *approximation* (`rendering_python`); it
is part of the API surface and is never used as a fallback for the HIP path.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from ._amp import fp32_boundary_bwd, fp32_boundary_fwd


def _capturing() -> bool:
    """is the current stream being captured into a graph (no caching of tensors created then, no deferred checks)"""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _hw(sr_size):
    # sr_size arrives as list, CPU tensor or GPU int tensor (gsasr_model.py:148,202); make ints once -- a GPU tensor
    # with ONE device-to-host copy (the reference's `int(sr_size[0])`, `int(sr_size[1])` are two synchronisations)
    if torch.is_tensor(sr_size):
        v = sr_size.tolist()
        return int(v[0]), int(v[1])
    return int(sr_size[0]), int(sr_size[1])


def _to_kernel_frame(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size, step_size):
    """sigmas/coords in the kernels' align-corners frame (reference :121-124): note the x/y swap -- the
    network's sigma_x is the ROW std, the kernel's first sigma pairs with WIDTH."""
    H, W = _hw(sr_size)
    sigmas = torch.cat([sigma_y / step_size * 2 / (W - 1), sigma_x / step_size * 2 / (H - 1), rho],
                       dim=-1).contiguous()
    cx = (coords[:, 0:1] + 1 - 1 / W) * W / (W - 1) - 1.0
    cy = (coords[:, 1:2] + 1 - 1 / H) * H / (H - 1) - 1.0
    return sigmas, torch.cat([cx, cy], dim=-1).contiguous(), colours_with_alpha.contiguous(), H, W


class _Splat(torch.autograd.Function):
    """`GSCUDA.apply(sigmas, coords, colors, zeros(H,W,3)[, dmax])` without the zero image: the forward kernel
    stores into an uninitialised buffer (GSASR_FLAG_OVERWRITE_IMAGE) -- same values, one memset and one
    12 B/px read less.  Used only where this module itself owns the image (the non-chunked renderers)."""

    @staticmethod
    @fp32_boundary_fwd
    def forward(ctx, sigmas, coords, colors, H, W, dmax):
        from . import _cabi
        plan = _cabi.plan(sigmas, coords, colors, H, W, dmax)
        img = torch.empty(H, W, 3, device=sigmas.device, dtype=torch.float32)
        _cabi.forward(plan, img, overwrite=True)
        ctx.save_for_backward(sigmas, coords, colors)
        ctx.plan = plan
        return img

    @staticmethod
    @torch.autograd.function.once_differentiable
    @fp32_boundary_bwd
    def backward(ctx, grad_output):
        from . import _cabi
        sigmas, coords, colors = ctx.saved_tensors
        g = (torch.empty_like(sigmas), torch.empty_like(coords), torch.empty_like(colors))
        _cabi.backward(ctx.plan, sigmas, coords, colors, grad_output.contiguous(), *g, overwrite=True)
        return (*g, None, None, None)


class _SplatInto(torch.autograd.Function):
    """One chunk of a chunked render: `img += splat(chunk)` into the image this module owns, with the support cutoff of
    the WHOLE set (`cutoff`): the adaptive default is tau = ln(N/1e-5), and chunks planned on their own would each cull
    with their own smaller tau -- the skipped mass would grow with the number of chunks (tools/fuzz_host.py: 1e-4 of an
    image rendered in 1 700 chunks) instead of staying below 1e-5.  Unlike a chain of the reference-shaped
    `GSCUDA.apply` calls (whose backward returns None for `rendered_img`, so only the LAST chunk would see a gradient --
    the reference's chain has the same gap, utils/gaussian_splatting.py:146-151), the image's gradient is passed on: every
    chunk is differentiated."""

    @staticmethod
    @fp32_boundary_fwd
    def forward(ctx, sigmas, coords, colors, img, dmax, cutoff):
        from . import _cabi
        sigmas, coords, colors = sigmas.contiguous(), coords.contiguous(), colors.contiguous()
        plan = _cabi.plan(sigmas, coords, colors, img.shape[0], img.shape[1], dmax, cutoff=cutoff)
        _cabi.forward(plan, img, overwrite=False)
        ctx.mark_dirty(img)
        ctx.save_for_backward(sigmas, coords, colors)
        ctx.plan = plan
        return img

    @staticmethod
    @torch.autograd.function.once_differentiable
    @fp32_boundary_bwd
    def backward(ctx, grad_output):
        from . import _cabi
        sigmas, coords, colors = ctx.saved_tensors
        g = (torch.empty_like(sigmas), torch.empty_like(coords), torch.empty_like(colors))
        _cabi.backward(ctx.plan, sigmas, coords, colors, grad_output.contiguous(), *g, overwrite=True)
        return (*g, grad_output, None, None)


def _render_chunked(sigmas, xy, col, H, W, dmax, device, buffer_size):
    from . import _cabi
    final_image = torch.zeros(H, W, 3, device=device, dtype=torch.float32)
    tau = _cabi.resolve_cutoff(0.0, max(1, sigmas.shape[0]))
    for a, b in _chunks(sigmas.shape[0], buffer_size):
        if sigmas[a:b].shape[0] == 0:
            continue
        final_image = _SplatInto.apply(sigmas[a:b], xy[a:b], col[a:b], final_image, dmax, tau)   # kernels accumulate (+=)
    return final_image.permute(2, 0, 1).contiguous()


# Which backward kernel the fused entry points plan for: "gaussian" (one wave per Gaussian; needs the upstream gradient
# permuted to [H,W,3]), "tile" (one workgroup per 32x16-px tile; reads the planar gradient in place; deterministic), "home"
# (round 6: a workgroup finishes the Gaussians binned in its tile from a staged region; interleaved gradient; deterministic)
# or "auto" (the measured choice per shape).
BACKWARD_KERNEL = "auto"


def _backward_kernel(n_pixels: int, n_gaussians: int, shape=None) -> int:
    """the C flag of the backward kernel the fused entry points plan for (FLAG_BWD_TILE / _GAUSSIAN / _HOME)"""
    from . import _cabi
    forced = {"tile": _cabi.FLAG_BWD_TILE, "gaussian": _cabi.FLAG_BWD_GAUSSIAN, "home": _cabi.FLAG_BWD_HOME}.get(BACKWARD_KERNEL)
    if forced is not None:
        return forced
    if shape is not None and _cabi.kernel_choices_registered():      # a choice measured and registered for this shape (gsasr_amd/tune.py)
        hit = _cabi.get_kernel_choice(shape())
        if hit is not None:
            for f in (_cabi.FLAG_BWD_TILE, _cabi.FLAG_BWD_GAUSSIAN, _cabi.FLAG_BWD_HOME):
                if hit[0] & f:
                    return f
    if _tile_backward(n_pixels, n_gaussians):
        return _cabi.FLAG_BWD_TILE
    # round 6 (the library's own rule, splat_common.h:bwd_wants_home): denser than one Gaussian per two pixels on at least 1024 tiles
    # of 32 x 16 px -- through this API 1024^2 at 16 per LR pixel -5%, 1280^2 -7%, 1152^2 x3 -17%, a batch of 16 x 256^2 -4%
    # (profiles/r06_home_default.txt)
    if n_pixels < 2 * n_gaussians and n_pixels >= 1024 * 512:
        return _cabi.FLAG_BWD_HOME
    return _cabi.FLAG_BWD_GAUSSIAN


def _tile_backward(n_pixels: int, n_gaussians: int, shape=None) -> bool:
    """Measured through this API on MI355X (tools/e2e_modes.py, profiles/history/r02_e2e_modes.txt): with >= 4 HR pixels per
    Gaussian (one Gaussian per LR pixel at x2 and up) the tile-stationary backward is level or ahead end to end -- it reads
    the planar gradient in place, where the Gaussian-stationary kernel needs it interleaved first -- and it is
    deterministic; at 16 Gaussians per LR pixel (the training crops: ~1 pixel per Gaussian) a tile holds thousands of
    Gaussians and the Gaussian-stationary kernel is 15-50% faster.  Small images have too few tiles to fill the chip."""
    if BACKWARD_KERNEL != "auto":
        return BACKWARD_KERNEL == "tile"
    if shape is not None:       # a choice measured and registered for this shape (gsasr_amd/tune.py) goes before the rule
        from . import _cabi
        if _cabi.kernel_choices_registered():
            hit = _cabi.get_kernel_choice(shape())
            if hit is not None and hit[0] & (_cabi.FLAG_BWD_TILE | _cabi.FLAG_BWD_GAUSSIAN):
                return bool(hit[0] & _cabi.FLAG_BWD_TILE)
    return n_pixels >= 4 * n_gaussians and n_pixels >= 128 * 1024


def _step_shape(n, H, W, dm):
    from . import _cabi
    return lambda: _cabi.make_dims(n, H, W, dm)


def _batch_shape(n_per, sizes, dm):
    from . import _cabi
    return lambda: _cabi.make_batch_dims(n_per, sizes, max(w for _, w in sizes), max(h for h, _ in sizes), dm)


# The caller's scale factor as a HINT for the forward kernel (never for the numbers).  The C library sees pixels per Gaussian, which
# says "x2-sized windows" for x8 at Fea2GS's 16 Gaussians per LR pixel (4 px per Gaussian) -- the windows there are x8's, and the
# wide forward (16 x 16-px sub-tiles) is 5..8% ahead (profiles/history/r05_inference_sweep.txt: x8d16_2048).  This module knows the scale.
SCALE_HINT = True


def _forward_flag(scale, H: int, W: int) -> int:
    """FLAG_FWD_WIDE from x5 up on images of 2 Mpx and more (the library's own line at one Gaussian per LR pixel), else 0"""
    if not SCALE_HINT or isinstance(scale, bool) or not isinstance(scale, (int, float)):
        return 0            # (a tensor would cost a synchronisation to read: the library's rule stays)
    from . import _cabi
    return _cabi.FLAG_FWD_WIDE if scale >= 5.0 and H * W >= 2 * 1024 * 1024 else 0


def _plan_flags(needs_grad: bool, kernel) -> int:
    """flags of a fused step's plan: the backward kernel is chosen HERE, explicitly (the library's own default would
    otherwise plan slots for large images that this module then never uses); planar gradient in, forward-only plans
    for inference.  `kernel`: the C flag from _backward_kernel (or True / False = tile / Gaussian-stationary)"""
    from . import _cabi
    if not needs_grad:
        return _cabi.FLAG_FORWARD_ONLY
    if isinstance(kernel, bool):
        kernel = _cabi.FLAG_BWD_TILE if kernel else _cabi.FLAG_BWD_GAUSSIAN
    return _cabi.FLAG_CHW_GRAD | int(kernel)


class _FusedStep(torch.autograd.Function):
    """raw decoder output `gs_parameters[N,9]` -> `[3,H,W]` image with ONE prologue kernel (activations +
    kernel-frame conversion, reference :174-180 and :121-123) in front of the splat, the splat writing the
    planar layout directly (no `permute(2,0,1).contiguous()` pass, reference :129), and the matching chain
    rule behind the splat's backward.  Replaces ~15 elementwise launches in forward
    and ~30 in backward; numerically the same expressions evaluated in fp32.  The step size is `step` (a `[1]`
    device tensor), or -- `step is None` -- `default_step / scale_modify[0]` formed on the device from the caller's
    `scale_modify` tensor (`_StepSource`)."""

    @staticmethod
    @fp32_boundary_fwd
    def forward(ctx, gs_parameters, step, H, W, dmax, scale_modify=None, default_step=1.2, extra_flags=0):
        from . import _cabi
        # the planar gradient autograd hands back goes to the C call as it is (GSASR_FLAG_CHW_GRAD): the
        # tile-stationary backward stages the planes directly, the Gaussian-stationary one behind one interleaving
        # kernel inside the same call -- no torch permute / allocation on the host path either way
        flags = _plan_flags(ctx.needs_input_grad[0], _backward_kernel(H * W, gs_parameters.shape[0], _step_shape(gs_parameters.shape[0], H, W, dmax))) | int(extra_flags)
        img, plan = _cabi.step_forward(gs_parameters, step, H, W, dmax, flags, scale_modify, default_step)   # one C call: prologue + plan + splat
        ctx.save_for_backward(gs_parameters, step)
        ctx.plan = plan
        return img

    @staticmethod
    @torch.autograd.function.once_differentiable
    @fp32_boundary_bwd
    def backward(ctx, grad_output):
        from . import _cabi
        gs_parameters, step = ctx.saved_tensors
        g = _cabi.step_backward(ctx.plan, gs_parameters, step, grad_output.contiguous(), chw=True)
        return g, None, None, None, None, None, None, None


class _FusedStepSampled(torch.autograd.Function):
    """`_FusedStep` for `sample_coords`: only the requested pixels are evaluated (`[3,S]`), instead of rendering
    `[3,H,W]` and indexing it once per point as the reference does (:214-216).  The backward is
    the sampled one too: each Gaussian visits the points inside its window, not every pixel of it."""

    @staticmethod
    @fp32_boundary_fwd
    def forward(ctx, gs_parameters, step, H, W, dmax, points, scale_modify=None, default_step=1.2):
        from . import _cabi
        out, plan, state = _cabi.step_sample_forward(gs_parameters, step, H, W, dmax, points, scale_modify, default_step)
        ctx.save_for_backward(gs_parameters, step)
        ctx.plan, ctx.state = plan, state
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    @fp32_boundary_bwd
    def backward(ctx, grad_output):
        from . import _cabi
        gs_parameters, step = ctx.saved_tensors
        return (_cabi.step_sample_backward(ctx.plan, ctx.state, gs_parameters, step, grad_output.contiguous()),
                None, None, None, None, None, None, None)


class _FusedBatchSampled(torch.autograd.Function):
    """`_FusedBatch` for `sample_coords[B,S,2]` -> `[B,3,S]`."""

    @staticmethod
    @fp32_boundary_fwd
    def forward(ctx, gs_parameters, steps, sizes, dmax, points):
        from . import _cabi
        out, plan, state = _cabi.batch_sample_forward(gs_parameters, steps, sizes, dmax, points)
        ctx.save_for_backward(gs_parameters, steps)
        ctx.plan, ctx.state = plan, state
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    @fp32_boundary_bwd
    def backward(ctx, grad_output):
        from . import _cabi
        gs_parameters, steps = ctx.saved_tensors
        return (_cabi.step_sample_backward(ctx.plan, ctx.state, gs_parameters, steps, grad_output.contiguous()),
                None, None, None, None)


# Kernel time of the sampled path equals the full render's at about a quarter of the pixels, but the alternative ends
# in a torch advanced-indexing gather whose backward (index_put_ with accumulate) takes longer than either rasterizer
# (DESIGN.md 3b): the sampled kernels are used unless the points outnumber the pixels.
SAMPLED_MAX_FRACTION = 1.0


def _as_points(sample_coords):
    """`sample_coords` as an integer `[S,2]` tensor, or None if it is not a plain list/tensor of (row, column) pairs"""
    sc = sample_coords
    if not torch.is_tensor(sc):
        try:
            sc = torch.as_tensor(sc)
        except Exception:
            return None
    if sc.dim() != 2 or sc.shape[1] != 2 or sc.dtype.is_floating_point or sc.dtype == torch.bool:
        return None
    return sc


def _fused_ok(gs_parameters) -> bool:
    return gs_parameters.is_cuda and gs_parameters.dtype == torch.float32 and gs_parameters.dim() == 2 \
        and gs_parameters.shape[1] == 9


_STEP_TENSORS = {}      # (value, device) -> [1] float32 device tensor of a python-number step size (read-only)


def _step_tensor(step_size, dev):
    if torch.is_tensor(step_size):
        return step_size.detach().to(device=dev, dtype=torch.float32).reshape(1)   # stays on the device: no sync
    # (keyed by the stream as well: the tensor is filled asynchronously on the stream that first asks for the value)
    key = (float(step_size), dev, None if _capturing() else torch.cuda.current_stream(dev).cuda_stream)
    t = _STEP_TENSORS.get(key)
    if t is None or _capturing():
        t = torch.full((1,), float(step_size), device=dev, dtype=torch.float32)
        if not _capturing():
            if len(_STEP_TENSORS) > 256:
                _STEP_TENSORS.clear()
            _STEP_TENSORS[key] = t
    return t


class _StepSource:
    """`default_step_size / scale_modify[0]`, not evaluated: the fused entry points hand the caller's `scale_modify`
    tensor to the plan's first kernel, which forms the step size and checks `[0] == [1]` itself (gsasr_step_forward_sm)"""
    __slots__ = ("scale_modify", "default_step")

    def __init__(self, scale_modify, default_step):
        self.scale_modify, self.default_step = scale_modify, default_step


def _fused_render(gs_parameters, sr_size, step_size, dmax, scale=None):
    """[3,H,W] through the fused prologue; `step_size` may be a python number, a (GPU) tensor or a `_StepSource`."""
    H, W = _hw(sr_size)
    dm = None if dmax is None else float(dmax)
    if step_size.__class__ is _StepSource:
        out = _fused_step(gs_parameters.contiguous(), None, H, W, dm, step_size.scale_modify, step_size.default_step, _forward_flag(scale, H, W))
        deferred_asserts.watch(gs_parameters.device)      # (after the launch: a look covers this call's own pair)
        return out
    step = _step_tensor(step_size, gs_parameters.device)
    return _fused_step(gs_parameters.contiguous(), step, H, W, dm, extra_flags=_forward_flag(scale, H, W))


def _fused_step(gs_parameters, step, H, W, dm, scale_modify=None, default_step=1.2, extra_flags=0):
    """`_FusedStep.apply`, as a C++ autograd node when the extension is there (gsasr_amd/_cpp_node.py: the engine calls its
    backward without taking the GIL -- the reference's training loop makes sixteen of these nodes per step)"""
    from . import _cpp_node
    if _cpp_node.load() is None:
        return _FusedStep.apply(gs_parameters, step, H, W, dm, scale_modify, default_step, extra_flags)
    needs_grad = gs_parameters.requires_grad and torch.is_grad_enabled()
    flags = _plan_flags(needs_grad, _backward_kernel(H * W, gs_parameters.shape[0], _step_shape(gs_parameters.shape[0], H, W, dm))) | int(extra_flags)
    return _cpp_node.fused_step_apply(gs_parameters, step, H, W, dm, flags, scale_modify, default_step)


def _fused_batch(gs_parameters, steps, sizes, dm, scale_modify=None, default_step=1.2):
    """`_FusedBatch.apply`, through the same C++ node when it is there"""
    from . import _cpp_node
    if _cpp_node.load() is None:
        return _FusedBatch.apply(gs_parameters, steps, sizes, dm, scale_modify, default_step)
    needs_grad = gs_parameters.requires_grad and torch.is_grad_enabled()
    tile = _backward_kernel(sum(h * w for h, w in sizes), gs_parameters.shape[0] * gs_parameters.shape[1],
                            _batch_shape(gs_parameters.shape[1], sizes, dm))
    return _cpp_node.fused_step_apply(gs_parameters, steps, 0, 0, dm, _plan_flags(needs_grad, tile), scale_modify, default_step, sizes=sizes)


def rendering_cuda(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size, step_size, device):
    sigmas, xy, col, H, W = _to_kernel_frame(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size, step_size)
    final_image = _Splat.apply(sigmas, xy, col, H, W, None)
    return final_image.permute(2, 0, 1).contiguous()


def rendering_cuda_dmax(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size, step_size, device, dmax=1):
    sigmas, xy, col, H, W = _to_kernel_frame(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size, step_size)
    final_image = _Splat.apply(sigmas, xy, col, H, W, float(dmax))
    return final_image.permute(2, 0, 1).contiguous()


def _chunks(n, buffer_size):
    # the reference runs len//buffer_size + 1 slices, the last possibly empty (:146-151)
    for k in range(n // buffer_size + 1):
        yield k * buffer_size, (k + 1) * buffer_size


def rendering_cuda_buffer(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size, step_size, device,
                          buffer_size=1000000):
    sigmas, xy, col, H, W = _to_kernel_frame(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size, step_size)
    return _render_chunked(sigmas, xy, col, H, W, None, device, buffer_size)


def rendering_cuda_dmax_buffer(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size, step_size, device,
                               dmax=1, buffer_size=1000000):
    sigmas, xy, col, H, W = _to_kernel_frame(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size, step_size)
    return _render_chunked(sigmas, xy, col, H, W, float(dmax), device, buffer_size)


def rendering_python(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size, step_size, device):
    """The reference's `cuda_rendering=False` path (:11-84): every Gaussian is sampled on a
    `num_step x num_step` grid in sigma units, normalised by its sampled peak (+1e-4) and bilinearly
    resampled onto the HR grid centred at its mean.  An approximation of the kernels, kept for API
    completeness (BASELINE.json config 1); plain torch ops on `device`."""
    H, W = _hw(sr_size)
    # `step_size` is used as it comes: in the reference's default mode it is the 0-dim fp32 tensor
    # `default_step_size / scale_modify[0]`, so `10 * 2 / step_size` and `i * step_size` are fp32 operations -- at scale 3.3
    # the grid has int(55.0) = 55 steps where the same expression on the Python float gives int(54.9999...) = 54
    # (tests/golden/tiled_frac_s3p3_30x16.npz)
    step = step_size
    n = sigma_x.shape[0]
    cxy = rho * sigma_x * sigma_y
    if ((sigma_x ** 2) * (sigma_y ** 2) - cxy ** 2 < 0).any():
        raise ValueError("Covariance matrix must be positive semi-definite")
    cov = torch.stack([torch.cat([sigma_x ** 2, cxy], -1), torch.cat([cxy, sigma_y ** 2], -1)], dim=-2)
    inv = torch.inverse(cov)
    num_step = int(10 * 2 / step)
    ax = torch.tensor([k * step for k in range(num_step)], device=device)
    ax = ax - ax.mean()
    xy = torch.stack([ax[:, None].expand(num_step, num_step), ax[None, :].expand(num_step, num_step)], dim=-1)
    final_image = torch.zeros((3, H, W), device=device)
    max_buffer = 2000
    for s0 in range(0, n, max_buffer):
        s1 = min(s0 + max_buffer, n)
        b = s1 - s0
        z = torch.einsum("xyi,bij,xyj->bxy", xy, -0.5 * inv[s0:s1], xy)
        kernel = torch.exp(z) / (2 * math.pi * torch.sqrt(torch.det(cov[s0:s1])).view(b, 1, 1))
        kernel = kernel / (kernel.amax(dim=(-1, -2), keepdim=True) + 1e-4)
        kernel = kernel[:, None].expand(b, 3, num_step, num_step)
        theta = torch.zeros(b, 2, 3, dtype=torch.float32, device=device)
        theta[:, 0, 0] = W / num_step
        theta[:, 1, 1] = H / num_step
        theta[:, 0, 2] = -coords[s0:s1, 0] * W / num_step
        theta[:, 1, 2] = -coords[s0:s1, 1] * H / num_step
        grid = F.affine_grid(theta, size=(b, 3, H, W), align_corners=False)
        moved = F.grid_sample(kernel, grid, align_corners=False)
        final_image = final_image + (colours_with_alpha[s0:s1, :, None, None] * moved).sum(0)
    return final_image


# The reference's `assert scale_modify[0] == scale_modify[1]` without a host synchronisation per call: gsasr_amd/_deferred.py
from ._deferred import _DeferredAsserts, deferred_asserts  # noqa: E402,F401


def _sm_source_ok(scale_modify) -> bool:
    """can the fused entry points read this `scale_modify` on the device themselves (a `[2]`-like float32 CUDA tensor)"""
    return (scale_modify.__class__ is torch.Tensor and scale_modify.is_cuda and scale_modify.dtype is torch.float32
            and scale_modify.dim() == 1 and scale_modify.shape[0] >= 2 and scale_modify.stride(0) == 1)


def _step_size(scale, scale_modify, default_step_size, mode, fused=False):
    """the reference's step size (:163-172).  `fused`: the caller is the fused path, which can take a `_StepSource`
    (scale_modify left on the device, nothing evaluated here) instead of a value"""
    if mode == 'scale':
        final_scale = scale
    elif mode == 'scale_modify':
        if fused and _sm_source_ok(scale_modify):
            return _StepSource(scale_modify, float(default_step_size))
        if torch.is_tensor(scale_modify) and scale_modify.is_cuda and not _capturing():
            deferred_asserts.add(scale_modify, "scale_modify is not the same")
        elif not (torch.is_tensor(scale_modify) and scale_modify.is_cuda):
            assert scale_modify[0] == scale_modify[1], f"scale_modify is not the same-{scale_modify}"
        final_scale = scale_modify[0]
    else:  # the reference leaves final_scale unbound here (UnboundLocalError, a NameError subclass)
        raise UnboundLocalError(f"mode-{mode} must be scale or scale_modify")
    return default_step_size / final_scale


def _activate(gs_parameters):
    # reference :174-180
    sigma_x = 0.99999 * torch.sigmoid(gs_parameters[:, 0:1]) + 1e-6
    sigma_y = 0.99999 * torch.sigmoid(gs_parameters[:, 1:2]) + 1e-6
    rho = 0.999999 * torch.tanh(gs_parameters[:, 2:3])
    alpha = torch.sigmoid(gs_parameters[:, 3:4])
    colours = torch.sigmoid(gs_parameters[:, 4:7])
    coords = gs_parameters[:, 7:9] * 2 - 1
    return sigma_x, sigma_y, rho, coords, colours * alpha


def _resolve_dmax(dmax, dmax_mode, sr_size):
    if dmax_mode == 'dynamic':
        H, W = _hw(sr_size)
        return (dmax + 2) / min(H, W)
    if dmax_mode == 'fix':
        return dmax
    raise ValueError(f"dmax_mode-{dmax_mode} must be fix or dynamic")


def _sample(final_image, sample_coords):
    """reference :214-216: `stack([img[:, c[0], c[1]] for c in sample_coords], dim=1)` -> `[3, S]`.  An `[S,2]`
    integer tensor (what the datasets produce, continuous_bicubic_downsample_dataset.py:87-88) is gathered with
    ONE indexing op instead of S of them (same values, same gradient scatter); anything else takes the loop."""
    if sample_coords is None:
        return final_image
    if torch.is_tensor(sample_coords) and sample_coords.dim() == 2 and sample_coords.shape[1] == 2 \
            and not sample_coords.dtype.is_floating_point and sample_coords.dtype != torch.bool:
        sc = sample_coords.to(device=final_image.device, dtype=torch.long)
        return final_image[:, sc[:, 0], sc[:, 1]]
    return torch.stack([final_image[:, c[0], c[1]] for c in sample_coords], dim=1)


def generate_2D_gaussian_splatting_step(sr_size, gs_parameters, scale, scale_modify, sample_coords=None,
                                        default_step_size=1.2, cuda_rendering=True, mode='scale_modify',
                                        if_dmax=True, dmax_mode='fix', dmax=25):
    if gs_parameters.dtype != torch.float32:
        # under bf16 autocast the decoder happens to emit fp32; make that explicit
        gs_parameters = gs_parameters.float()
    fused = cuda_rendering and _fused_ok(gs_parameters)
    step_size = _step_size(scale, scale_modify, default_step_size, mode, fused=fused)
    if fused:
        # fused prologue + splat (same maths as the unfused branch below, one kernel instead of ~15)
        H, W = _hw(sr_size)
        dmax_eff = _resolve_dmax(dmax, dmax_mode, (H, W)) if if_dmax else None
        pts = _as_points(sample_coords) if sample_coords is not None else None
        if pts is not None and 0 < pts.shape[0] <= SAMPLED_MAX_FRACTION * H * W:
            dm = None if dmax_eff is None else float(dmax_eff)
            if step_size.__class__ is _StepSource:
                out = _FusedStepSampled.apply(gs_parameters.contiguous(), None, H, W, dm, pts, step_size.scale_modify,
                                              step_size.default_step)
                deferred_asserts.watch(gs_parameters.device)
                return out
            return _FusedStepSampled.apply(gs_parameters.contiguous(), _step_tensor(step_size, gs_parameters.device), H, W, dm, pts)
        return _sample(_fused_render(gs_parameters, (H, W), step_size, dmax_eff, scale), sample_coords)
    sigma_x, sigma_y, rho, coords, colours_with_alpha = _activate(gs_parameters)
    dev = sigma_x.device
    if cuda_rendering:
        if if_dmax:
            dmax = _resolve_dmax(dmax, dmax_mode, sr_size)
            final_image = rendering_cuda_dmax(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size,
                                              step_size, dmax=dmax, device=dev)
        else:
            final_image = rendering_cuda(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size, step_size,
                                         device=dev)
    else:
        final_image = rendering_python(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size, step_size,
                                       device=dev)
    return _sample(final_image, sample_coords)


class _FusedBatch(torch.autograd.Function):
    """A whole training batch in one set of launches: `gs_parameters[B,N,9]` ->
    `[B,3,Hmax,Wmax]`, sample b rendered on its own `sizes[b]` pixel grid in the top-left corner of its slot
    and zero elsewhere.  Replaces the reference's per-sample Python loop of `generate_2D_gaussian_splatting_step`
    + `F.pad` (basicsr/models/gsasr_model.py:191-233): B x (prologue, plan, splat) launches and B autograd nodes
    become one of each."""

    @staticmethod
    @fp32_boundary_fwd
    def forward(ctx, gs_parameters, steps, sizes, dmax, scale_modify=None, default_step=1.2):
        from . import _cabi
        tile = _backward_kernel(sum(h * w for h, w in sizes), gs_parameters.shape[0] * gs_parameters.shape[1],
                                _batch_shape(gs_parameters.shape[1], sizes, dmax))
        flags = _plan_flags(ctx.needs_input_grad[0], tile)
        img, plan = _cabi.batch_forward(gs_parameters, steps, sizes, dmax, flags, scale_modify, default_step)
        ctx.save_for_backward(gs_parameters, steps)
        ctx.plan = plan
        ctx.h_max = max(h for h, _ in sizes)
        return img[:, :, : ctx.h_max]          # the slot is h_max rounded up to whole 16-row tiles

    @staticmethod
    @torch.autograd.function.once_differentiable
    @fp32_boundary_bwd
    def backward(ctx, grad_output):
        from . import _cabi
        gs_parameters, steps = ctx.saved_tensors
        # [B,3,Hmax,Wmax] read in place (rows per plane = Hmax <= slot): pixels outside a sample's own grid are never read
        return _cabi.batch_backward(ctx.plan, gs_parameters, steps, grad_output.contiguous(), chw=True), None, None, None, None, None


def _batch_step_sizes(scales, scale_modifies, default_step_size, mode, dev):
    """`[B]` float32 device tensor of `_step_size(...)` per sample with ONE stack / division / check for the batch
    (the per-sample form costs a kernel and, through the reference's `assert scale_modify[0] == scale_modify[1]`,
    a host synchronisation per sample)."""
    def col(vals):
        if all(torch.is_tensor(v) for v in vals):
            return torch.stack([v.reshape(()) for v in vals]).to(device=dev, dtype=torch.float32)
        # python numbers: one host-to-device copy per DISTINCT tuple of values, not per call
        key = (tuple(float(v) for v in vals), dev, None if _capturing() else torch.cuda.current_stream(dev).cuda_stream)
        t = _STEP_TENSORS.get(key)
        if t is None or _capturing():
            t = torch.tensor(key[0], dtype=torch.float32, device=dev)
            if not _capturing():
                if len(_STEP_TENSORS) > 256:
                    _STEP_TENSORS.clear()
                _STEP_TENSORS[key] = t
        return t
    if mode == 'scale':
        final = col(list(scales))
    elif mode == 'scale_modify':
        if all(torch.is_tensor(sm) and sm.is_cuda for sm in scale_modifies):
            both = torch.stack([sm.reshape(-1)[:2] for sm in scale_modifies]).to(device=dev, dtype=torch.float32)   # [B,2], one kernel
            a, b = both[:, 0], both[:, 1]
            if not _capturing():      # the reference's assert, without draining the pipeline
                bad = (a != b).to(torch.float32)
                deferred_asserts.add(torch.stack([bad.sum(), bad.new_zeros(())]), "scale_modify is not the same (batched step): differing pairs, 0")
        else:
            a, b = col([sm[0] for sm in scale_modifies]), col([sm[1] for sm in scale_modifies])
            assert bool((a == b).all()), f"scale_modify is not the same-{scale_modifies}"
        final = a
    else:
        raise UnboundLocalError(f"mode-{mode} must be scale or scale_modify")
    return default_step_size / final


def max_canvas_batch(h_max: int) -> int:
    """samples of up to `h_max` rows that fit ONE batched canvas: 64 slots (GSASR_MAX_BATCH) of h_max rounded up to
    whole 16-row tiles, 32 767 canvas rows in all (the plan packs pixel indices in 15 bits)"""
    slot = (int(h_max) + 15) // 16 * 16
    return max(1, min(64, 32767 // slot))


def generate_2D_gaussian_splatting_batch(sr_sizes, gs_parameters, scales, scale_modifies, default_step_size=1.2,
                                         mode='scale_modify', if_dmax=True, dmax_mode='fix', dmax=25, sample_coords=None):
    """Batched `generate_2D_gaussian_splatting_step`: `gs_parameters` `[B,N,9]`, per-sample
    `sr_sizes[b]`, `scales[b]`, `scale_modifies[b]`; returns `[B,3,Hmax,Wmax]` with every sample zero-padded to
    the largest size -- exactly `torch.stack([F.pad(step(...), ...)])` of the reference's loop.  With
    `sample_coords` `[B,S,2]` (row, column on each sample's own grid; gsasr_model.py:196-197) it returns the
    `[B,3,S]` stack of the per-sample `[3,S]` results instead."""
    B = gs_parameters.shape[0]
    if torch.is_tensor(sr_sizes) and sr_sizes.dim() == 2:      # e.g. the [B,2] GPU tensor of gsasr_model.py:147: ONE copy to the host
        sizes = [(int(r[0]), int(r[1])) for r in sr_sizes.tolist()]
    else:
        sizes = [_hw(s) for s in sr_sizes]
    if not (len(sizes) == B == len(scales) == len(scale_modifies)):
        raise ValueError("one sr_size, scale and scale_modify per sample")
    if gs_parameters.dtype != torch.float32:
        gs_parameters = gs_parameters.float()
    uniform_dmax = (not if_dmax) or dmax_mode == 'fix' or len(set(sizes)) == 1
    cap = max_canvas_batch(max(h for h, _ in sizes))
    if B > cap >= 2 and gs_parameters.is_cuda and gs_parameters.dim() == 3 and uniform_dmax:
        # more samples than one canvas holds (64 slots, 32 767 rows): several canvases of `cap` samples, the last
        # one possibly a single sample (which takes the per-sample path below)
        parts = [generate_2D_gaussian_splatting_batch(sr_sizes[a: a + cap], gs_parameters[a: a + cap], scales[a: a + cap],
                                                      scale_modifies[a: a + cap], default_step_size, mode, if_dmax, dmax_mode,
                                                      dmax, None if sample_coords is None else sample_coords[a: a + cap])
                 for a in range(0, B, cap)]
        if sample_coords is None:
            h_max, w_max = max(h for h, _ in sizes), max(w for _, w in sizes)
            parts = [F.pad(o, (0, w_max - o.shape[3], 0, h_max - o.shape[2])) for o in parts]
        return torch.cat(parts)
    if 1 < B <= cap and gs_parameters.is_cuda and gs_parameters.dim() == 3 and gs_parameters.shape[2] == 9 and uniform_dmax:
        dev = gs_parameters.device
        dmax_eff = _resolve_dmax(dmax, dmax_mode, sizes[0]) if if_dmax else None
        dm = None if dmax_eff is None else float(dmax_eff)
        if sample_coords is None and mode == 'scale_modify':
            # scale_modify pairs that are already on the device go to the plan's first kernel as they are (one [B,2]
            # tensor: no kernel at all; a list of [2] tensors: one torch.stack): no division, comparison or copy here
            sm = None
            # (a [B,2] tensor is read in place as rows of stride >= 2 on the Gaussians' device: an `.expand(B, 2)` view --
            # row stride 0 -- or a tensor on another GPU takes the evaluated path below)
            if torch.is_tensor(scale_modifies) and scale_modifies.dim() == 2 and _sm_source_ok(scale_modifies[0]) \
                    and scale_modifies.stride(0) >= 2 and scale_modifies.device == dev:
                sm = scale_modifies
            elif not torch.is_tensor(scale_modifies) and all(_sm_source_ok(v) and v.device == dev for v in scale_modifies):
                sm = torch.stack([v[:2] for v in scale_modifies])
            if sm is not None:
                out = _fused_batch(gs_parameters.contiguous(), None, tuple(sizes), dm, sm, float(default_step_size))
                deferred_asserts.watch(dev)
                return out
        steps = _batch_step_sizes(scales, scale_modifies, default_step_size, mode, dev)
        if sample_coords is None:
            return _fused_batch(gs_parameters.contiguous(), steps, tuple(sizes), dm)
        pts = sample_coords if torch.is_tensor(sample_coords) else torch.as_tensor(sample_coords)
        if pts.dim() == 3 and pts.shape[0] == B and pts.shape[2] == 2 and not pts.dtype.is_floating_point \
                and 0 < pts.shape[1] <= SAMPLED_MAX_FRACTION * min(h * w for h, w in sizes):
            return _FusedBatchSampled.apply(gs_parameters.contiguous(), steps, tuple(sizes), dm, pts)
    # per-sample path (single sample, > 64 samples, or a per-sample dmax): same kernels, one sample at a time
    h_max, w_max = max(h for h, _ in sizes), max(w for _, w in sizes)
    outs = []
    for b in range(B):
        o = generate_2D_gaussian_splatting_step(sr_sizes[b], gs_parameters[b], scales[b], scale_modifies[b],
                                                sample_coords=None if sample_coords is None else sample_coords[b],
                                                default_step_size=default_step_size, mode=mode, if_dmax=if_dmax,
                                                dmax_mode=dmax_mode, dmax=dmax)
        if sample_coords is None:
            o = torch.nn.functional.pad(o, (0, w_max - sizes[b][1], 0, h_max - sizes[b][0]))
        outs.append(o)
    return torch.stack(outs)


def generate_2D_gaussian_splatting_step_buffer(sr_size, gs_parameters, scale, scale_modify, sample_coords=None,
                                               default_step_size=1.2, cuda_rendering=True, mode='scale_modify',
                                               if_dmax=True, dmax_mode='fix', dmax=25, buffer_size=4000000):
    step_size = _step_size(scale, scale_modify, default_step_size, mode)
    if gs_parameters.dtype != torch.float32:
        gs_parameters = gs_parameters.float()
    sigma_x, sigma_y, rho, coords, colours_with_alpha = _activate(gs_parameters)
    dev = sigma_x.device
    if cuda_rendering:
        if if_dmax:
            dmax = _resolve_dmax(dmax, dmax_mode, sr_size)
            final_image = rendering_cuda_dmax_buffer(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size,
                                                     step_size, dmax=dmax, device=dev, buffer_size=buffer_size)
        else:
            final_image = rendering_cuda_buffer(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size,
                                                step_size, device=dev, buffer_size=buffer_size)
    else:
        final_image = rendering_python(sigma_x, sigma_y, rho, coords, colours_with_alpha, sr_size, step_size,
                                       device=dev)
    return _sample(final_image, sample_coords)
