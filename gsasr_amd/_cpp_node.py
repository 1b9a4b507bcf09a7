"""The drop-in `GSCUDA` node as a C++ torch::autograd::Function (gsasr_amd/csrc/gsasr_autograd.cpp), when it has been built.

The reference renders one image per autograd node (utils/gs_cuda_dmax/gswrapper.py:22-44; sixteen nodes per training step in
basicsr/models/gsasr_model.py:191-233).  A Python `torch.autograd.Function` makes the autograd engine's worker thread take
the GIL to run `backward`; a C++ node is called by the engine directly (−12 us per apply + backward on the development
hosts).  It calls the same C ABI (libgsasr_splat.so) through function pointers this module hands it.

`fast_apply(...)` is installed as `GSCUDA.apply` by the two gswrapper modules when `load()` succeeds; otherwise (extension not
built, `GSASR_AMD_CPP_NODE=0`) the Python Functions stay -- same kernels, same results, a little more host time.  This is
an accelerator of the HOST path only: there is no compute in it.
"""
import ctypes
import importlib.util
import os
from typing import Optional

import torch

PKG = os.path.dirname(os.path.abspath(__file__))
EXT_PATH = os.path.join(PKG, "lib", "_gsasr_autograd.so")
_ext = None
_tried = False
_AUTOTUNE = os.environ.get("GSASR_AMD_AUTOTUNE", "0") not in ("", "0")


def load():
    """the extension module, bound to libgsasr_splat.so's entry points, or None"""
    global _ext, _tried
    if _tried:
        return _ext
    _tried = True
    if os.environ.get("GSASR_AMD_CPP_NODE", "1") == "0" or not os.path.exists(EXT_PATH):
        return None
    try:
        from . import _cabi
        L = _cabi.lib()
        spec = importlib.util.spec_from_file_location("_gsasr_autograd", EXT_PATH)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)

        def addr(f):
            return ctypes.cast(f, ctypes.c_void_p).value
        m.bind(addr(L.gsasr_splat_plan), addr(L.gsasr_splat_forward), addr(L.gsasr_splat_backward),
               addr(L.gsasr_splat_workspace_bytes), addr(L.gsasr_last_error))
        m.bind_step(addr(L.gsasr_step_workspace_bytes), addr(L.gsasr_step_forward), addr(L.gsasr_step_forward_sm),
                    addr(L.gsasr_step_backward))
        _ext = m
    except Exception as e:      # an extension built against another torch, a missing symbol: the Python node serves
        import warnings
        warnings.warn(f"gsasr_amd: C++ autograd node not loaded ({e!r}); using the Python torch.autograd.Function")
        _ext = None
    return _ext


def fast_apply(sigmas, coords, colors, rendered_img, dmax: Optional[float]):
    """`GSCUDA.apply(sigmas, coords, colors, rendered_img[, dmax])` through the C++ node (dmax None: gs_cuda, the unbounded op)"""
    if dmax is not None and not (float(dmax) >= 0.0):
        raise RuntimeError("dmax must be >= 0")
    if torch.is_autocast_enabled("cuda"):     # the autocast-safe boundary of _amp.py: fp32 in, autocast off inside
        sigmas, coords, colors = (t.float() if t.is_floating_point() and t.dtype is not torch.float32 else t
                                  for t in (sigmas, coords, colors))
        with torch.autocast("cuda", enabled=False):
            return _call(sigmas, coords, colors, rendered_img, dmax)
    return _call(sigmas, coords, colors, rendered_img, dmax)


def _call(sigmas, coords, colors, rendered_img, dmax):
    if not (isinstance(sigmas, torch.Tensor) and sigmas.is_cuda):
        raise RuntimeError("sigmas must be a CUDA tensor")
    dev = sigmas.device
    d = -1.0 if dmax is None else float(dmax)
    if _AUTOTUNE and rendered_img.dim() == 3:      # GSASR_AMD_AUTOTUNE=1 (gsasr_amd/tune.py)
        from . import tune
        tune.autotune_hook(sigmas, coords, colors, rendered_img.shape[0], rendered_img.shape[1], dmax,
                           torch.is_grad_enabled() and (sigmas.requires_grad or coords.requires_grad or colors.requires_grad))
    if dev.index is not None and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):
            return _ext.gscuda_apply(sigmas, coords, colors, rendered_img, d, torch.cuda.current_stream(dev).cuda_stream,
                                     torch.cuda.is_current_stream_capturing())
    return _ext.gscuda_apply(sigmas, coords, colors, rendered_img, d, torch.cuda.current_stream(dev).cuda_stream,
                             torch.cuda.is_current_stream_capturing())


def _f32(t):
    return t.float() if (t is not None and t.is_floating_point() and t.dtype is not torch.float32) else t


def fused_step_apply(gs_parameters, step, H: int, W: int, dmax: Optional[float], flags: int, scale_modify=None,
                     default_step: float = 1.2, sizes=None):
    """`gsasr_amd.gaussian_splatting._FusedStep.apply` / `_FusedBatch.apply` through the C++ node: raw decoder output
    `[N,9]` -> `[3,H,W]`, or with `sizes` = [(h_b, w_b)] the batched canvas `[B,N,9]` -> `[B,3,Hmax,Wmax]`.  `flags` = the
    plan flags the caller chose (backward kernel, forward-only)."""
    from . import _cabi
    if dmax is not None and not (float(dmax) >= 0.0):
        raise RuntimeError("dmax must be >= 0")
    if torch.is_autocast_enabled("cuda"):
        gs_parameters, step = _f32(gs_parameters), _f32(step)
        with torch.autocast("cuda", enabled=False):
            return fused_step_apply(gs_parameters, step, H, W, dmax, flags, scale_modify, default_step, sizes)
    if not (isinstance(gs_parameters, torch.Tensor) and gs_parameters.is_cuda):
        raise RuntimeError("gs_parameters must be a CUDA tensor")
    dev = gs_parameters.device
    batch = 0 if sizes is None else len(sizes)
    flat, slot, h, w, h_max = [], 0, int(H), int(W), int(H)
    if batch:
        if not (1 < batch <= _cabi.MAX_BATCH) or gs_parameters.dim() != 3 or gs_parameters.shape[0] != batch:
            raise RuntimeError("gs_parameters must be [B,N,9] with one (h,w) per sample")
        h_max, w = max(int(a) for a, _ in sizes), max(int(b) for _, b in sizes)
        slot = (h_max + 15) // 16 * 16
        h = slot * batch
        flat = [int(v) for hw in sizes for v in hw]
        if step is not None and step.numel() != batch:
            raise RuntimeError("gs_parameters must be [B,N,9] with one step size and one (h,w) per sample")
    sm, stride, mm = None, 0, 0
    if scale_modify is not None:
        _, stride = _cabi._sm_ptr(scale_modify, max(batch, 1))
        sm, mm = scale_modify, _cabi.mismatch_flag(dev).data_ptr()
    elif step is None:
        raise RuntimeError("step size missing")
    d = -1.0 if dmax is None else float(dmax)
    if dev.index is not None and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):
            return _ext.step_apply(gs_parameters, step, h, w, d, int(flags), sm, stride, float(default_step), mm, flat, slot, h_max,
                                   torch.cuda.current_stream(dev).cuda_stream, torch.cuda.is_current_stream_capturing())
    return _ext.step_apply(gs_parameters, step, h, w, d, int(flags), sm, stride, float(default_step), mm, flat, slot, h_max,
                           torch.cuda.current_stream(dev).cuda_stream, torch.cuda.is_current_stream_capturing())


def clear_pool() -> None:
    if _ext is not None:
        _ext.clear_pool()
