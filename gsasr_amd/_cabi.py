"""ctypes binding of libgsasr_splat.so (include/gsasr_splat.h) for torch tensors.

PyTorch is plumbing here: it owns device memory and the stream; every call below passes raw
`data_ptr()`s, sizes and `torch.cuda.current_stream().cuda_stream` through the C ABI.  There is NO
fallback: if the library is missing or a tensor is not a contiguous fp32 CUDA tensor this raises.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# (GSASR_SPLAT_LIB: development override, e.g. to A/B two builds of the library on the same GPU box)
LIB_PATH = os.environ.get("GSASR_SPLAT_LIB") or os.path.join(_PKG, "lib", "libgsasr_splat.so")

EXPORTS = (
    "gsasr_abi_version", "gsasr_last_error", "gsasr_splat_workspace_bytes", "gsasr_splat_plan",
    "gsasr_splat_forward", "gsasr_splat_backward", "gsasr_gs_render", "gsasr_gs_render_backward",
    "gsasr_gs_render_dmax", "gsasr_gs_render_backward_dmax", "gsasr_set_default_cutoff",
    "gsasr_get_default_cutoff", "gsasr_prologue_forward", "gsasr_prologue_backward",
    "gsasr_step_workspace_bytes", "gsasr_step_forward", "gsasr_step_backward",
    "gsasr_band_select", "gsasr_band_merge", "gsasr_resolve_cutoff",
    "gsasr_sample_workspace_bytes", "gsasr_splat_sample_forward", "gsasr_splat_sample_backward",
    "gsasr_step_sample_forward", "gsasr_step_sample_backward",
    "gsasr_step_forward_sm", "gsasr_step_sample_forward_sm", "gsasr_plan_cutoff", "gsasr_release_launcher_scratch", "gsasr_forward_subtile_width",
    "gsasr_set_kernel_choice", "gsasr_get_kernel_choice", "gsasr_clear_kernel_choices",
)

FLAG_OVERWRITE_IMAGE = 2   # GSASR_FLAG_OVERWRITE_IMAGE
FLAG_OVERWRITE_GRADS = 4   # GSASR_FLAG_OVERWRITE_GRADS
FLAG_CHW_IMAGE = 8         # GSASR_FLAG_CHW_IMAGE
FLAG_STRIDE8 = 16          # GSASR_FLAG_STRIDE8
FLAG_CHW_GRAD = 32         # GSASR_FLAG_CHW_GRAD
FLAG_FORWARD_ONLY = 64     # GSASR_FLAG_FORWARD_ONLY
FLAG_BWD_GAUSSIAN = 128    # GSASR_FLAG_BWD_GAUSSIAN
FLAG_BWD_TILE = 256        # GSASR_FLAG_BWD_TILE
FLAG_BWD_ATOMIC = 512      # GSASR_FLAG_BWD_ATOMIC
FLAG_BWD_HOME = 32768      # GSASR_FLAG_BWD_HOME (home-tile backward, ABI 7)
FLAG_COUNTERS_CLEAN = 1024 # GSASR_FLAG_COUNTERS_CLEAN
FLAG_PARITY = 2048         # GSASR_FLAG_PARITY
FLAG_CUTOFF_CAP = 4096     # GSASR_FLAG_CUTOFF_CAP
FLAG_FWD_WIDE, FLAG_FWD_NARROW = 8192, 16384      # forward kernel choice (development A/B, tests): 16x16 / 8x16 sub-tiles
EXACT_CUTOFF = 104.0    # GSASR_SPLAT_EXACT_CUTOFF
NO_CUTOFF = -1.0


class Dims(ctypes.Structure):
    """struct gsasr_dims"""
    _fields_ = [("s", ctypes.c_int), ("h", ctypes.c_int), ("w", ctypes.c_int), ("c", ctypes.c_int),
                ("dmax", ctypes.c_float), ("row0", ctypes.c_int), ("row1", ctypes.c_int),
                ("cutoff", ctypes.c_float), ("flags", ctypes.c_uint),
                ("batch", ctypes.c_int), ("slot", ctypes.c_int), ("sample_hw", ctypes.POINTER(ctypes.c_int)),
                ("grad_rows", ctypes.c_int), ("list_cap", ctypes.c_int)]


_lib = None


def lib():
    """Load the shared library (once). Raises if it has not been built -- there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP rasterizer is not built. Run `python -m gsasr_amd.build` "
                "(or __graft_entry__.build()). gsasr_amd has no CPU/eager fallback by design.")
        L = ctypes.CDLL(LIB_PATH)
        vp, f, i, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
        dp = ctypes.POINTER(Dims)
        L.gsasr_abi_version.restype = i
        L.gsasr_last_error.restype = ctypes.c_char_p
        L.gsasr_splat_workspace_bytes.restype = sz
        L.gsasr_splat_workspace_bytes.argtypes = [dp]
        L.gsasr_splat_plan.restype = i
        L.gsasr_splat_plan.argtypes = [vp, vp, vp, dp, vp, sz, vp]
        L.gsasr_splat_forward.restype = i
        L.gsasr_splat_forward.argtypes = [dp, vp, sz, vp, vp]
        L.gsasr_splat_backward.restype = i
        L.gsasr_splat_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, dp, vp, sz, vp]
        L.gsasr_gs_render.restype = i
        L.gsasr_gs_render.argtypes = [vp, vp, vp, vp, i, i, i, i, vp]
        L.gsasr_gs_render_backward.restype = i
        L.gsasr_gs_render_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, vp]
        L.gsasr_gs_render_dmax.restype = i
        L.gsasr_gs_render_dmax.argtypes = [vp, vp, vp, vp, i, i, i, i, f, vp]
        L.gsasr_gs_render_backward_dmax.restype = i
        L.gsasr_gs_render_backward_dmax.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, f, vp]
        L.gsasr_prologue_forward.restype = i
        L.gsasr_prologue_forward.argtypes = [vp, vp, i, i, i, vp, vp, vp, vp]
        L.gsasr_prologue_backward.restype = i
        L.gsasr_prologue_backward.argtypes = [vp, vp, i, i, i, vp, vp, vp, vp, vp]
        L.gsasr_step_workspace_bytes.restype = sz
        L.gsasr_step_workspace_bytes.argtypes = [dp]
        L.gsasr_step_forward.restype = i
        L.gsasr_step_forward.argtypes = [vp, vp, dp, vp, sz, vp, vp]
        L.gsasr_step_backward.restype = i
        L.gsasr_step_backward.argtypes = [vp, vp, vp, vp, dp, vp, sz, vp]
        L.gsasr_step_forward_sm.restype = i
        L.gsasr_step_forward_sm.argtypes = [vp, vp, i, f, vp, dp, vp, sz, vp, vp]
        L.gsasr_step_sample_forward_sm.restype = i
        L.gsasr_step_sample_forward_sm.argtypes = [vp, vp, i, f, vp, dp, vp, sz, vp, i, vp, vp, sz, vp]
        L.gsasr_band_select.restype = i
        L.gsasr_band_select.argtypes = [vp, dp, i, i, i, vp, vp, vp, vp, vp, vp]
        L.gsasr_band_merge.restype = i
        L.gsasr_band_merge.argtypes = [vp, i, vp, vp, vp, vp, vp, i, vp]
        L.gsasr_sample_workspace_bytes.restype = sz
        L.gsasr_sample_workspace_bytes.argtypes = [dp, i]
        L.gsasr_splat_sample_forward.restype = i
        L.gsasr_splat_sample_forward.argtypes = [dp, vp, sz, vp, i, vp, vp, sz, vp]
        L.gsasr_splat_sample_backward.restype = i
        L.gsasr_splat_sample_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, dp, vp, sz, vp, i, vp, sz, vp]
        L.gsasr_step_sample_forward.restype = i
        L.gsasr_step_sample_forward.argtypes = [vp, vp, dp, vp, sz, vp, i, vp, vp, sz, vp]
        L.gsasr_step_sample_backward.restype = i
        L.gsasr_step_sample_backward.argtypes = [vp, vp, vp, vp, dp, vp, sz, vp, i, vp, sz, vp]
        L.gsasr_set_default_cutoff.restype = None
        L.gsasr_set_default_cutoff.argtypes = [f]
        L.gsasr_get_default_cutoff.restype = f
        L.gsasr_resolve_cutoff.restype = f
        L.gsasr_resolve_cutoff.argtypes = [f, i]
        L.gsasr_forward_subtile_width.restype = i
        L.gsasr_forward_subtile_width.argtypes = [dp]
        L.gsasr_plan_cutoff.restype = i
        L.gsasr_plan_cutoff.argtypes = [dp, vp, sz, vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint)]
        L.gsasr_set_kernel_choice.restype = i
        L.gsasr_set_kernel_choice.argtypes = [dp, ctypes.c_uint, i]
        L.gsasr_get_kernel_choice.restype = i
        L.gsasr_get_kernel_choice.argtypes = [dp, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_int)]
        L.gsasr_clear_kernel_choices.restype = None
        if L.gsasr_abi_version() != 7:
            raise RuntimeError("libgsasr_splat.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().gsasr_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (status {rc}): {msg}")


def _chk(t: torch.Tensor, name: str, shape_tail: Optional[Tuple[int, ...]] = None) -> int:
    # same failure mode as the reference's CHECK_INPUT (gswrapper.cpp:5-7): RuntimeError
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {t.dtype}); the kernels read raw fp32")
    if shape_tail is not None and tuple(t.shape[-len(shape_tail):]) != shape_tail:
        raise RuntimeError(f"{name} has shape {tuple(t.shape)}, expected [..., {shape_tail}]")
    return t.data_ptr()


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _ptr3(t: torch.Tensor, name: str, last: int) -> int:
    """`_chk(t, name, (last,))` for the hot path: one combined test first, the explanatory ones only when it fails"""
    if t.__class__ is torch.Tensor and t.is_cuda and t.dtype is torch.float32 and t.is_contiguous() and t.shape[-1] == last:
        return t.data_ptr()
    return _chk(t, name, (last,))


class _Nop:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOP = _Nop()


def _on(device: torch.device):
    """`torch.cuda.device(device)` only when it is not the current device already (the context manager costs ~10 us,
    a third of a call's host time on the single-GPU-per-process layout this package is meant for)"""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return _NOP if idx == torch.cuda.current_device() else torch.cuda.device(device)


class _WorkspacePool:
    """Workspaces of the step entry points, kept between calls so that a plan can skip its counter memset: every plan
    zeroes the per-cell counters of the OTHER parity on the side (GSASR_FLAG_COUNTERS_CLEAN / GSASR_FLAG_PARITY), so a
    workspace that comes back from a finished step is clean for the flipped parity.  Keyed by (device, stream, size):
    reuse is ordered by the stream, exactly like the caching allocator's own reuse.  Bounded: a few workspaces per
    key, MAX_BYTES in all (least recently used keys go first -- training on ragged sizes meets many sizes)."""
    KEEP = 4                    # free workspaces kept per key (forward and backward of a few steps in flight)
    MAX_BYTES = 2 << 30

    def __init__(self):
        self.free = {}          # key -> [(tensor, parity)]; dict order = least recently used first
        self.bytes = 0

    def take(self, key, nbytes, dev):
        lst = self.free.get(key)
        if lst:
            ws, parity = lst.pop()
            self.bytes -= ws.numel()
            if lst:
                self.free[key] = self.free.pop(key)     # most recently used
            else:
                del self.free[key]
            return ws, parity, True
        return torch.empty(nbytes, dtype=torch.uint8, device=dev), 0, False

    def give(self, key, ws, parity):
        lst = self.free.pop(key, [])
        if len(lst) < self.KEEP and ws.numel() <= self.MAX_BYTES:
            lst.append((ws, parity))
            self.bytes += ws.numel()
        if lst:
            self.free[key] = lst
        while self.bytes > self.MAX_BYTES and self.free:
            old = next(iter(self.free))
            for t, _ in self.free.pop(old):
                self.bytes -= t.numel()

    def clear(self):
        self.free.clear()
        self.bytes = 0


_POOL = _WorkspacePool()


def clear_workspace_pool() -> None:
    """Drop the pooled plan workspaces (up to `_WorkspacePool.MAX_BYTES` of device memory that `torch.cuda.empty_cache()`
    cannot see as free while the pool holds it): call it next to `empty_cache()` when memory is tight.  Also frees the
    scratch the reference-shaped C launchers (module `gscuda`) keep per stream."""
    _POOL.clear()
    if _lib is not None:
        _lib.gsasr_release_launcher_scratch()
    from . import _cpp_node
    _cpp_node.clear_pool()


@dataclass
class Plan:
    """Binning workspace of one (sigmas, coords, colors, dims): shared by forward and backward."""
    dims: Dims
    workspace: torch.Tensor
    device: torch.device
    pool_key: Optional[tuple] = None      # set for pooled workspaces: returned (with the parity flipped) when the plan dies.
    #                                       A caller that re-plans on `workspace` itself (C entry points with its own flags)
    #                                       must set this to None: the pool's "counters clean" bookkeeping no longer holds
    parity: int = 0

    def __del__(self):
        if self.pool_key is not None:
            try:
                _POOL.give(self.pool_key, self.workspace, self.parity ^ 1)
            except Exception:      # interpreter shutdown
                pass


def make_dims(s: int, h: int, w: int, dmax: Optional[float], rows: Optional[Tuple[int, int]] = None,
              cutoff: float = 0.0, flags: int = 0, list_cap: int = 0) -> Dims:
    r0, r1 = (0, h) if rows is None else rows
    d = Dims(int(s), int(h), int(w), 3, -1.0 if dmax is None else float(dmax), int(r0), int(r1),
             float(cutoff), int(flags))
    d.list_cap = int(list_cap)      # tile lists: 0 = the library's capacity estimate, > 0 entries per tile, < 0 none
    return d


_PLAN_DIMS = {}     # (s, h, w, dmax, rows, cutoff, flags) -> ([Dims fresh, pooled parity 0, pooled parity 1], workspace bytes)


def _plan_dims(s: int, h: int, w: int, dmax, rows, cutoff: float, flags: int, list_cap: int = 0):
    key = (s, h, w, dmax, rows, cutoff, flags, list_cap)
    hit = _PLAN_DIMS.get(key)
    if hit is None:      # (dims structs + workspace size per shape: built once, not per call)
        if dmax is not None and not (float(dmax) >= 0.0):
            raise RuntimeError("dmax must be >= 0")
        variants = [make_dims(s, h, w, dmax, rows, cutoff, int(flags) | f, list_cap)
                    for f in (0, FLAG_COUNTERS_CLEAN, FLAG_COUNTERS_CLEAN | FLAG_PARITY)]
        nbytes = lib().gsasr_splat_workspace_bytes(ctypes.byref(variants[0]))
        if nbytes == 0:
            check(-1, "gsasr_splat_workspace_bytes")
        if len(_PLAN_DIMS) > 512:
            _PLAN_DIMS.clear()
        hit = _PLAN_DIMS[key] = (variants, nbytes)
    return hit


def plan(sigmas: torch.Tensor, coords: torch.Tensor, colors: torch.Tensor, h: int, w: int,
         dmax: Optional[float], rows: Optional[Tuple[int, int]] = None, cutoff: float = 0.0,
         flags: int = 0, list_cap: int = 0) -> Plan:
    ps = _ptr3(sigmas, "sigmas", 3)
    pc = _ptr3(coords, "coords", 2)
    pk = _ptr3(colors, "colors", 3)
    s = sigmas.shape[0]
    if coords.shape[0] != s or colors.shape[0] != s:
        raise RuntimeError("sigmas, coords, colors disagree on the number of Gaussians")
    variants, nbytes = _plan_dims(s, int(h), int(w), dmax, rows, cutoff, flags, list_cap)
    dev = sigmas.device
    with _on(dev):
        stream = _stream(dev)
        if torch.cuda.is_current_stream_capturing():
            # a captured plan is replayed on the same workspace with the same parity: it must zero its own counters
            ws, parity, clean, pool_key = torch.empty(nbytes, dtype=torch.uint8, device=dev), 0, False, None
        else:
            pool_key = (dev.index, stream, nbytes, s, h, w, 0, 0, flags & _LAYOUT_FLAGS)
            ws, parity, clean = _POOL.take(pool_key, nbytes, dev)
        d = variants[1 + parity] if clean else variants[0]
        check(lib().gsasr_splat_plan(ps, pc, pk, ctypes.byref(d), ws.data_ptr(), nbytes, stream), "gsasr_splat_plan")
    return Plan(d, ws, dev, pool_key, parity)


_LAYOUT_FLAGS = FLAG_FORWARD_ONLY | FLAG_BWD_TILE | FLAG_BWD_GAUSSIAN | FLAG_BWD_ATOMIC | FLAG_BWD_HOME | FLAG_CHW_GRAD | FLAG_STRIDE8


def _pool_key(d: Dims, nbytes: int, dev):
    """Workspaces are interchangeable only between plans of the SAME layout: "the counters of parity p are zero" is a
    statement about where the counter arrays lie and how long they are (grid size), so the key carries everything the
    layout depends on, not just the byte count (two small shapes easily round to the same size)."""
    return (dev.index, _stream(dev), nbytes, d.s, d.h, d.w, d.batch, d.slot, d.flags & _LAYOUT_FLAGS)


def _pooled_workspace(d: Dims, nbytes: int, dev):
    """a workspace for a plan with dims `d`: from the pool, with the flags that tell the plan its counters are clean
    (set on `d`), or fresh.  Returns (workspace, pool key or None, parity)."""
    if torch.cuda.is_current_stream_capturing():
        # a captured plan is replayed on the same workspace with the same parity: it must zero its own counters
        return torch.empty(nbytes, dtype=torch.uint8, device=dev), None, 0
    pool_key = _pool_key(d, nbytes, dev)
    ws, parity, clean = _POOL.take(pool_key, nbytes, dev)
    if clean:      # (these two bits do not change the layout)
        d.flags |= FLAG_COUNTERS_CLEAN | (FLAG_PARITY if parity else 0)
    return ws, pool_key, parity


def _dims_with(p: Plan, extra_flags: int) -> Dims:
    """the plan's dims with `extra_flags` added (copies are cached on the Dims object, which plans of one shape share)"""
    d0 = p.dims
    if not extra_flags or (d0.flags & extra_flags) == extra_flags:
        return d0
    cache = d0.__dict__.get("_with")
    if cache is None:
        cache = d0.__dict__["_with"] = {}
    d = cache.get(extra_flags)
    if d is None or d.flags != (d0.flags | extra_flags):
        d = Dims.from_buffer_copy(d0)
        d.flags |= extra_flags
        if hasattr(d0, "_keepalive"):
            d._keepalive = d0._keepalive
        cache[extra_flags] = d
    return d


def forward(p: Plan, img: torch.Tensor, overwrite: bool = False, chw: bool = False, flags: int = 0) -> torch.Tensor:
    """img += splat (reference contract), or img = splat when `overwrite` (img may be torch.empty).
    `chw`: img is planar [3, rows, W] instead of [rows, W, 3].  `flags`: FLAG_FWD_WIDE / FLAG_FWD_NARROW (kernel choice)."""
    d0 = p.dims
    rows = d0.row1 - d0.row0
    if chw:
        pi = _chk(img, "rendered_img", (rows, d0.w))
        if img.shape[0] != 3 or img.dim() != 3 or img.device != p.device:
            raise RuntimeError("rendered_img does not match the plan (shape / device)")
    else:
        pi = _ptr3(img, "rendered_img", 3)
        if img.dim() != 3 or img.shape[0] != rows or img.shape[1] != d0.w or img.device != p.device:
            raise RuntimeError("rendered_img does not match the plan (shape / device)")
    d = _dims_with(p, (FLAG_OVERWRITE_IMAGE if overwrite else 0) | (FLAG_CHW_IMAGE if chw else 0) |
                   (flags & (FLAG_FWD_WIDE | FLAG_FWD_NARROW)))
    with _on(p.device):
        check(lib().gsasr_splat_forward(ctypes.byref(d), p.workspace.data_ptr(), p.workspace.numel(), pi,
                                        _stream(p.device)), "gsasr_splat_forward")
    return img


def forward_subtile_width(p: Plan, flags: int = 0) -> int:
    """16 when `forward(p, ..., flags=flags)` runs the wide forward (16 x 16 sub-tiles), 8 for the 8 x 16 kernels"""
    return int(lib().gsasr_forward_subtile_width(ctypes.byref(_dims_with(p, flags & (FLAG_FWD_WIDE | FLAG_FWD_NARROW)))))


def backward(p: Plan, sigmas, coords, colors, grad_img, g_sigmas, g_coords, g_colors, overwrite: bool = False) -> None:
    """g_* += gradients (reference contract: caller zero-fills), or g_* = gradients when `overwrite`."""
    d0 = p.dims
    pg = _ptr3(grad_img, "grads", 3)
    if grad_img.dim() != 3 or grad_img.shape[1] != d0.w:
        raise RuntimeError(f"grads has shape {tuple(grad_img.shape)}, expected [rows, {d0.w}, 3]")
    if grad_img.shape[0] != d0.row1 - d0.row0:
        raise RuntimeError("grads does not match the plan's row band")
    d = _dims_with(p, FLAG_OVERWRITE_GRADS if overwrite else 0)
    with _on(p.device):
        check(lib().gsasr_splat_backward(_ptr3(sigmas, "sigmas", 3), _ptr3(coords, "coords", 2), _ptr3(colors, "colors", 3), pg,
                                         _ptr3(g_sigmas, "grads_sigmas", 3), _ptr3(g_coords, "grads_coords", 2),
                                         _ptr3(g_colors, "grads_colors", 3), ctypes.byref(d), p.workspace.data_ptr(),
                                         p.workspace.numel(), _stream(p.device)), "gsasr_splat_backward")


_AUTOTUNE = os.environ.get("GSASR_AMD_AUTOTUNE", "0") not in ("", "0")


def _tune_on() -> bool:
    return _AUTOTUNE


def plan_forward(sigmas: torch.Tensor, coords: torch.Tensor, colors: torch.Tensor, img: torch.Tensor,
                 dmax: Optional[float]) -> Plan:
    """`plan` + `forward` (accumulating into `img[H,W,3]`) as one host call: what `GSCUDA.forward` does, with one device /
    stream lookup and one set of argument checks for both C calls"""
    ps = _ptr3(sigmas, "sigmas", 3)
    pc = _ptr3(coords, "coords", 2)
    pk = _ptr3(colors, "colors", 3)
    pi = _ptr3(img, "rendered_img", 3)
    if img.dim() != 3:
        raise RuntimeError("rendered_img must be [H,W,3]")
    s, h, w = sigmas.shape[0], img.shape[0], img.shape[1]
    dev = sigmas.device
    if coords.shape[0] != s or colors.shape[0] != s:
        raise RuntimeError("sigmas, coords, colors disagree on the number of Gaussians")
    if img.device != dev:
        raise RuntimeError("rendered_img does not match the plan (shape / device)")
    if _tune_on():      # GSASR_AMD_AUTOTUNE=1: a shape's first call measures the kernel combinations (gsasr_amd/tune.py)
        from . import tune
        tune.autotune_hook(sigmas, coords, colors, h, w, dmax,
                           torch.is_grad_enabled() and (sigmas.requires_grad or coords.requires_grad or colors.requires_grad))
    variants, nbytes = _plan_dims(s, h, w, dmax, None, 0.0, 0)
    L = lib()
    with _on(dev):
        stream = _stream(dev)
        if torch.cuda.is_current_stream_capturing():
            ws, parity, clean, pool_key = torch.empty(nbytes, dtype=torch.uint8, device=dev), 0, False, None
        else:
            pool_key = (dev.index, stream, nbytes, s, h, w, 0, 0, 0)
            ws, parity, clean = _POOL.take(pool_key, nbytes, dev)
        d = variants[1 + parity] if clean else variants[0]
        pw = ws.data_ptr()
        check(L.gsasr_splat_plan(ps, pc, pk, ctypes.byref(d), pw, nbytes, stream), "gsasr_splat_plan")
        check(L.gsasr_splat_forward(ctypes.byref(d), pw, nbytes, pi, stream), "gsasr_splat_forward")
    return Plan(d, ws, dev, pool_key, parity)


def backward_new(p: Plan, sigmas: torch.Tensor, coords: torch.Tensor, colors: torch.Tensor, grad_img: torch.Tensor):
    """`backward` into three fresh gradient tensors (stored, not accumulated): what `GSCUDA.backward` returns"""
    g_sigmas, g_coords, g_colors = torch.empty_like(sigmas), torch.empty_like(coords), torch.empty_like(colors)
    backward(p, sigmas, coords, colors, grad_img if grad_img.is_contiguous() else grad_img.contiguous(), g_sigmas, g_coords,
             g_colors, overwrite=True)
    return g_sigmas, g_coords, g_colors


# ---- packed [N,8] records (GSASR_FLAG_STRIDE8): the wire format of the multi-GPU exchange --------------
def _cols(packed: torch.Tensor, name: str):
    base = _chk(packed, name, (8,))
    if packed.dim() != 2:
        raise RuntimeError(f"{name} must be [N,8]")
    return base, base + 12, base + 20      # sigmas, coords, colors columns of the same records


def plan_packed(packed: torch.Tensor, h: int, w: int, dmax: Optional[float],
                rows: Optional[Tuple[int, int]] = None, cutoff: float = 0.0, workspace: Optional[torch.Tensor] = None,
                flags: int = 0) -> Plan:
    """`plan` for Gaussians held as one `[N,8]` tensor {sx,sy,rho,x,y,r,g,b}; no unpacking copies."""
    ps, pc, pk = _cols(packed, "packed")
    if dmax is not None and not (float(dmax) >= 0.0):
        raise RuntimeError("dmax must be >= 0")
    d = make_dims(packed.shape[0], h, w, dmax, rows, cutoff, FLAG_STRIDE8 | int(flags))
    L = lib()
    nbytes = L.gsasr_splat_workspace_bytes(ctypes.byref(d))
    if nbytes == 0:
        check(-1, "gsasr_splat_workspace_bytes")
    dev = packed.device
    with _on(dev):
        pool_key, parity = None, 0
        if workspace is not None:
            ws = workspace
        else:
            ws, pool_key, parity = _pooled_workspace(d, nbytes, dev)
        if ws.numel() < nbytes:
            raise RuntimeError("workspace smaller than gsasr_splat_workspace_bytes()")
        check(L.gsasr_splat_plan(ps, pc, pk, ctypes.byref(d), ws.data_ptr(), ws.numel(), _stream(dev)),
              "gsasr_splat_plan")
    return Plan(d, ws, dev, pool_key, parity)


def backward_packed(p: Plan, packed: torch.Tensor, grad_img: torch.Tensor, g_packed: torch.Tensor,
                    overwrite: bool = False) -> None:
    """`backward` with inputs and gradients as `[N,8]` records (the plan must come from `plan_packed`)."""
    if not (p.dims.flags & FLAG_STRIDE8):
        raise RuntimeError("backward_packed needs a plan made by plan_packed")
    ps, pc, pk = _cols(packed, "packed")
    gs, gc, gk = _cols(g_packed, "g_packed")
    pg = _chk(grad_img, "grads", (p.dims.w, 3))
    if grad_img.shape[0] != p.dims.row1 - p.dims.row0 or g_packed.shape[0] != p.dims.s or packed.shape[0] != p.dims.s:
        raise RuntimeError("grads / g_packed do not match the plan")
    d = _dims_with(p, FLAG_OVERWRITE_GRADS if overwrite else 0)
    with _on(p.device):
        check(lib().gsasr_splat_backward(ps, pc, pk, pg, gs, gc, gk, ctypes.byref(d), p.workspace.data_ptr(),
                                         p.workspace.numel(), _stream(p.device)), "gsasr_splat_backward")


def backward_to_packed(p: Plan, sigmas: torch.Tensor, coords: torch.Tensor, colors: torch.Tensor, grad_img: torch.Tensor,
                       g_packed: torch.Tensor, overwrite: bool = True) -> None:
    """`backward` of a plan made from three separate arrays, with the gradient written as ONE `[N,8]` array
    {d sx, d sy, d rho, d x, d y, d r, d g, d b} -- the buffer a collective then runs on in place (gsasr_amd/shard.py).
    The backward reads the plan, not the input arrays, so only the OUTPUT stride changes (GSASR_FLAG_STRIDE8 at backward time)."""
    if p.dims.flags & FLAG_STRIDE8:
        raise RuntimeError("the plan is packed already: use backward_packed")
    ps, pc, pk = _ptr3(sigmas, "sigmas", 3), _ptr3(coords, "coords", 2), _ptr3(colors, "colors", 3)
    gs, gc, gk = _cols(g_packed, "g_packed")
    pg = _chk(grad_img, "grads", (p.dims.w, 3))
    if grad_img.shape[0] != p.dims.row1 - p.dims.row0 or g_packed.shape[0] != p.dims.s or sigmas.shape[0] != p.dims.s:
        raise RuntimeError("grads / g_packed do not match the plan")
    d = _dims_with(p, FLAG_STRIDE8 | (FLAG_OVERWRITE_GRADS if overwrite else 0))
    with _on(p.device):
        check(lib().gsasr_splat_backward(ps, pc, pk, pg, gs, gc, gk, ctypes.byref(d), p.workspace.data_ptr(),
                                         p.workspace.numel(), _stream(p.device)), "gsasr_splat_backward")


def _chk_i32(t: torch.Tensor, name: str, n: int) -> int:
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and t.dtype == torch.int32 and t.numel() >= n):
        raise RuntimeError(f"{name} must be a contiguous int32 CUDA tensor with >= {n} elements")
    return t.data_ptr()


def band_select(packed: torch.Tensor, h: int, w: int, dmax: Optional[float], rows: Tuple[int, int],
                rows_above: int, rows_below: int, up: torch.Tensor, down: torch.Tensor, up_index: torch.Tensor,
                down_index: torch.Tensor, counts: torch.Tensor, cutoff: float = 0.0) -> None:
    """gsasr_band_select: fill `up`/`down` `[cap,8]` with this band's Gaussians that reach the neighbour bands."""
    base = _chk(packed, "packed", (8,))
    cap = up.shape[0]
    if down.shape[0] != cap:
        raise RuntimeError("up and down must have the same capacity")
    d = make_dims(packed.shape[0], h, w, dmax, rows, cutoff, FLAG_STRIDE8)
    with _on(packed.device):
        check(lib().gsasr_band_select(base, ctypes.byref(d), int(rows_above), int(rows_below), cap,
                                      _chk(up, "up", (8,)), _chk(down, "down", (8,)), _chk_i32(up_index, "up_index", cap),
                                      _chk_i32(down_index, "down_index", cap), _chk_i32(counts, "counts", 4),
                                      _stream(packed.device)), "gsasr_band_select")


def band_merge(g_packed: torch.Tensor, g_up: torch.Tensor, g_down: torch.Tensor, up_index: torch.Tensor,
               down_index: torch.Tensor, counts: torch.Tensor) -> None:
    """gsasr_band_merge: g_packed[index] += the gradients the neighbours returned for the selected records."""
    cap = g_up.shape[0]
    with _on(g_packed.device):
        check(lib().gsasr_band_merge(_chk(g_packed, "g_packed", (8,)), g_packed.shape[0], _chk(g_up, "g_up", (8,)),
                                     _chk(g_down, "g_down", (8,)), _chk_i32(up_index, "up_index", cap),
                                     _chk_i32(down_index, "down_index", cap), _chk_i32(counts, "counts", 4), cap,
                                     _stream(g_packed.device)), "gsasr_band_merge")


def prologue_forward(gs_parameters: torch.Tensor, step: torch.Tensor, h: int, w: int):
    """gs_parameters[N,9] (raw decoder output) -> kernel-frame (sigmas[N,3], coords[N,2], colors[N,3])."""
    pp = _chk(gs_parameters, "gs_parameters", (9,))
    ps = _chk(step, "step_size")
    n, dev = gs_parameters.shape[0], gs_parameters.device
    out = (torch.empty(n, 3, device=dev), torch.empty(n, 2, device=dev), torch.empty(n, 3, device=dev))
    with _on(dev):
        check(lib().gsasr_prologue_forward(pp, ps, n, int(h), int(w), out[0].data_ptr(), out[1].data_ptr(),
                                           out[2].data_ptr(), _stream(dev)), "gsasr_prologue_forward")
    return out


def prologue_backward(gs_parameters, step, h: int, w: int, g_sigmas, g_coords, g_colors) -> torch.Tensor:
    pp = _chk(gs_parameters, "gs_parameters", (9,))
    ps = _chk(step, "step_size")
    ptrs = [_chk(g_sigmas, "g_sigmas", (3,)), _chk(g_coords, "g_coords", (2,)), _chk(g_colors, "g_colors", (3,))]
    n, dev = gs_parameters.shape[0], gs_parameters.device
    gp = torch.empty(n, 9, device=dev)
    with _on(dev):
        check(lib().gsasr_prologue_backward(pp, ps, n, int(h), int(w), *ptrs, gp.data_ptr(), _stream(dev)),
              "gsasr_prologue_backward")
    return gp


_STEP_DIMS = {}     # (n, h, w, dmax, flags) -> (Dims, workspace bytes) of the single-image step entry points


_MISMATCH = {}      # device index -> int32[2] device tensor: the sticky "scale_modify pair differs" word of the _sm entry points


def mismatch_flag(dev: torch.device) -> torch.Tensor:
    t = _MISMATCH.get(dev.index)
    if t is None:
        t = _MISMATCH[dev.index] = torch.zeros(2, dtype=torch.int32, device=dev)
    return t


def _sm_ptr(sm: torch.Tensor, batch: int):
    """pointer + element stride of `scale_modify` pairs: a float32 CUDA tensor `[2]` (or longer) for one image, `[B, >=2]`
    rows for a batched canvas"""
    if not (sm.__class__ is torch.Tensor and sm.is_cuda and sm.dtype is torch.float32):
        raise RuntimeError("scale_modify must be a float32 CUDA tensor")
    if batch <= 1:
        if sm.dim() != 1 or sm.shape[0] < 2 or sm.stride(0) != 1:
            raise RuntimeError("scale_modify must be a contiguous [2] tensor")
        return sm.data_ptr(), 2
    if sm.dim() != 2 or sm.shape[0] != batch or sm.shape[1] < 2 or sm.stride(1) != 1 or sm.stride(0) < 2:
        raise RuntimeError(f"scale_modify must be [{batch}, 2] with unit inner stride")
    return sm.data_ptr(), int(sm.stride(0))


def step_forward(gs_parameters: torch.Tensor, step: Optional[torch.Tensor], h: int, w: int, dmax: Optional[float],
                 extra_flags: int = 0, scale_modify: Optional[torch.Tensor] = None, default_step_size: float = 1.2):
    """prologue + plan + forward in ONE call: raw `gs_parameters[N,9]` -> planar image `[3,h,w]` (fresh).
    `extra_flags`: FLAG_FORWARD_ONLY (no backward will follow), FLAG_BWD_TILE (plan for the tile-stationary backward).
    The step size is `step` (a `[1]` device tensor), or with `scale_modify` (a `[2]` float32 CUDA tensor) the reference's
    `default_step_size / scale_modify[0]` formed on the device, its `[0] == [1]` assert reported through `mismatch_flag`."""
    pp = _ptr3(gs_parameters, "gs_parameters", 9)
    dev = gs_parameters.device
    if scale_modify is None:
        ps = _chk(step, "step_size")
    else:
        psm, stride = _sm_ptr(scale_modify, 1)
    L = lib()
    key = (gs_parameters.shape[0], int(h), int(w), dmax, int(extra_flags))
    hit = _STEP_DIMS.get(key)
    if hit is None:      # (dims structs + workspace size per shape: built once, not per call)
        if dmax is not None and not (float(dmax) >= 0.0):
            raise RuntimeError("dmax must be >= 0")
        base = FLAG_OVERWRITE_IMAGE | FLAG_CHW_IMAGE | int(extra_flags)
        variants = [make_dims(gs_parameters.shape[0], h, w, dmax, flags=base | f)
                    for f in (0, FLAG_COUNTERS_CLEAN, FLAG_COUNTERS_CLEAN | FLAG_PARITY)]   # fresh / pooled parity 0 / 1
        nbytes = L.gsasr_step_workspace_bytes(ctypes.byref(variants[0]))
        if nbytes == 0:
            check(-1, "gsasr_step_workspace_bytes")
        if len(_STEP_DIMS) > 256:
            _STEP_DIMS.clear()
        hit = _STEP_DIMS[key] = (variants, nbytes)
    variants, nbytes = hit
    with _on(dev):
        stream = _stream(dev)
        if torch.cuda.is_current_stream_capturing():
            # a captured plan is replayed on the same workspace with the same parity: it must zero its own counters
            ws, parity, clean, pool_key = torch.empty(nbytes, dtype=torch.uint8, device=dev), 0, False, None
        else:
            d0 = variants[0]
            pool_key = (dev.index, stream, nbytes, d0.s, d0.h, d0.w, 0, 0, d0.flags & _LAYOUT_FLAGS)
            ws, parity, clean = _POOL.take(pool_key, nbytes, dev)
        d = variants[1 + parity] if clean else variants[0]
        img = torch.empty(3, int(h), int(w), dtype=torch.float32, device=dev)
        if scale_modify is None:
            check(L.gsasr_step_forward(pp, ps, ctypes.byref(d), ws.data_ptr(), nbytes, img.data_ptr(), stream),
                  "gsasr_step_forward")
        else:
            check(L.gsasr_step_forward_sm(pp, psm, stride, float(default_step_size), mismatch_flag(dev).data_ptr(),
                                          ctypes.byref(d), ws.data_ptr(), nbytes, img.data_ptr(), stream),
                  "gsasr_step_forward_sm")
    return img, Plan(d, ws, dev, pool_key, parity)


def step_backward(p: Plan, gs_parameters: torch.Tensor, step: Optional[torch.Tensor], grad: torch.Tensor, chw: bool = False) -> torch.Tensor:
    """splat backward + prologue backward in ONE call; `grad` is `[h,w,3]`, or with `chw` the planar `[3,h,w]` autograd
    hands back (tile-stationary backward, GSASR_FLAG_CHW_GRAD); returns d/d gs_parameters `[N,9]`.  `step=None`: the step
    size the forward's prologue used (kept in the workspace)."""
    pp = _ptr3(gs_parameters, "gs_parameters", 9)
    ps = None if step is None else _chk(step, "step_size")
    pg = _chk(grad, "grads", (3, p.dims.h, p.dims.w) if chw else (p.dims.h, p.dims.w, 3))
    d = _dims_with(p, FLAG_CHW_GRAD if chw else 0)
    with _on(p.device):
        gp = torch.empty_like(gs_parameters)
        check(lib().gsasr_step_backward(pp, ps, pg, gp.data_ptr(), ctypes.byref(d), p.workspace.data_ptr(),
                                        p.workspace.numel(), _stream(p.device)), "gsasr_step_backward")
    return gp


# ---- batched canvas --------------------------------------------------------------
MAX_BATCH = 64   # GSASR_MAX_BATCH


def make_batch_dims(n_per: int, sizes, w_max: int, h_max: int, dmax: Optional[float], cutoff: float = 0.0,
                    flags: int = 0) -> Dims:
    """dims of a canvas of len(sizes) slots; `sizes` = [(h_b, w_b)].  The returned struct owns the host array."""
    B = len(sizes)
    if not (1 < B <= MAX_BATCH):
        raise RuntimeError(f"batch size must be in 2..{MAX_BATCH}")
    slot = (int(h_max) + 15) // 16 * 16
    hw = (ctypes.c_int * (2 * B))(*[int(v) for hw_ in sizes for v in hw_])
    d = Dims(int(n_per) * B, slot * B, int(w_max), 3, -1.0 if dmax is None else float(dmax), 0, slot * B,
             float(cutoff), int(flags), B, slot, ctypes.cast(hw, ctypes.POINTER(ctypes.c_int)))
    d._keepalive = hw
    return d


def batch_forward(gs_parameters: torch.Tensor, steps: Optional[torch.Tensor], sizes, dmax: Optional[float], extra_flags: int = 0,
                  scale_modify: Optional[torch.Tensor] = None, default_step_size: float = 1.2):
    """prologue + plan + forward of a whole batch in ONE set of launches.
    `gs_parameters` [B,N,9], `steps` [B] (device), `sizes` [(h_b, w_b)] -> planar images `[B,3,slot,w_max]`
    (sample b in `[:, :, :h_b, :w_b]`, zero elsewhere) and the plan for `batch_backward`."""
    pp = _chk(gs_parameters, "gs_parameters", (9,))
    if gs_parameters.dim() != 3 or len(sizes) != gs_parameters.shape[0]:
        raise RuntimeError("gs_parameters must be [B,N,9] with one step size and one (h,w) per sample")
    if scale_modify is None:
        ps = _chk(steps, "step_sizes")
        if steps.numel() != gs_parameters.shape[0]:
            raise RuntimeError("gs_parameters must be [B,N,9] with one step size and one (h,w) per sample")
    else:
        psm, stride = _sm_ptr(scale_modify, gs_parameters.shape[0])
    if dmax is not None and not (float(dmax) >= 0.0):
        raise RuntimeError("dmax must be >= 0")
    B, n = gs_parameters.shape[0], gs_parameters.shape[1]
    h_max, w_max = max(h for h, _ in sizes), max(w for _, w in sizes)
    d = make_batch_dims(n, sizes, w_max, h_max, dmax, flags=FLAG_OVERWRITE_IMAGE | FLAG_CHW_IMAGE | int(extra_flags))
    L = lib()
    nbytes = L.gsasr_step_workspace_bytes(ctypes.byref(d))
    if nbytes == 0:
        check(-1, "gsasr_step_workspace_bytes")
    dev = gs_parameters.device
    with _on(dev):
        stream = _stream(dev)
        ws, pool_key, parity = _pooled_workspace(d, nbytes, dev)
        img = torch.empty(B, 3, d.slot, w_max, dtype=torch.float32, device=dev)
        if scale_modify is None:
            check(L.gsasr_step_forward(pp, ps, ctypes.byref(d), ws.data_ptr(), nbytes, img.data_ptr(), stream),
                  "gsasr_step_forward")
        else:
            check(L.gsasr_step_forward_sm(pp, psm, stride, float(default_step_size), mismatch_flag(dev).data_ptr(),
                                          ctypes.byref(d), ws.data_ptr(), nbytes, img.data_ptr(), stream),
                  "gsasr_step_forward_sm")
    return img, Plan(d, ws, dev, pool_key, parity)


def batch_backward(p: Plan, gs_parameters: torch.Tensor, steps: Optional[torch.Tensor], grad: torch.Tensor, chw: bool = False) -> torch.Tensor:
    """`grad` is `[B, slot, w_max, 3]`, or with `chw` the planar `[B, 3, rows, w_max]` autograd hands back (any
    `rows` >= every sample's height; tile-stationary backward); returns d/d gs_parameters `[B,N,9]`.  `steps=None`: the
    step sizes the forward's prologue used."""
    pp = _chk(gs_parameters, "gs_parameters", (9,))
    ps = None if steps is None else _chk(steps, "step_sizes")
    d = p.dims
    if chw:
        if grad.dim() != 4 or grad.shape[0] != d.batch or grad.shape[1] != 3 or grad.shape[3] != d.w:
            raise RuntimeError(f"grads has shape {tuple(grad.shape)}, expected [{d.batch}, 3, rows, {d.w}]")
        hmax = max(d.sample_hw[2 * b] for b in range(d.batch))
        if not (hmax <= grad.shape[2]):
            raise RuntimeError(f"grads has {grad.shape[2]} rows per plane, a sample has {hmax}")
        pg = _chk(grad, "grads")
        d = Dims.from_buffer_copy(p.dims)
        d.flags |= FLAG_CHW_GRAD
        d.grad_rows = int(grad.shape[2])
    else:
        pg = _chk(grad, "grads", (p.dims.batch, p.dims.slot, p.dims.w, 3))
    with _on(p.device):
        gp = torch.empty_like(gs_parameters)
        check(lib().gsasr_step_backward(pp, ps, pg, gp.data_ptr(), ctypes.byref(d), p.workspace.data_ptr(),
                                        p.workspace.numel(), _stream(p.device)), "gsasr_step_backward")
    return gp


# ---- sampled pixels ----------------------------------------------------------------
def _points(points: torch.Tensor, batch: int, device) -> Tuple[torch.Tensor, int]:
    """`[S,2]` (one image) or `[B,S,2]` (batched canvas) integer (row, column) pairs -> contiguous int32 on `device`."""
    if not (isinstance(points, torch.Tensor) and not points.dtype.is_floating_point and points.dtype != torch.bool
            and points.shape[-1:] == (2,) and points.dim() == (3 if batch > 1 else 2)
            and (batch <= 1 or points.shape[0] == batch)):
        raise RuntimeError("points must be an integer tensor [S,2] (or [B,S,2] for a batched canvas)")
    return points.to(device=device, dtype=torch.int32).contiguous(), int(points.shape[-2])


def _sample_ws(d: Dims, n_points: int, dev) -> torch.Tensor:
    nbytes = lib().gsasr_sample_workspace_bytes(ctypes.byref(d), n_points)
    if nbytes == 0:
        check(-1, "gsasr_sample_workspace_bytes")
    return torch.empty(nbytes, dtype=torch.uint8, device=dev)


def sample_forward(p: Plan, points: torch.Tensor):
    """values of the splat at `points` only: `[3,S]` (`[B,3,S]` on a batched canvas) + the state for `sample_backward`."""
    B = max(int(p.dims.batch), 1)
    pts, n = _points(points, B, p.device)
    with _on(p.device):
        sws = _sample_ws(p.dims, n, p.device)
        out = torch.empty((B, 3, n) if B > 1 else (3, n), dtype=torch.float32, device=p.device)
        check(lib().gsasr_splat_sample_forward(ctypes.byref(p.dims), p.workspace.data_ptr(), p.workspace.numel(),
                                               pts.data_ptr(), n, out.data_ptr(), sws.data_ptr(), sws.numel(),
                                               _stream(p.device)), "gsasr_splat_sample_forward")
    return out, (pts, n, sws)


def sample_backward(p: Plan, state, sigmas, coords, colors, grad_out, g_sigmas, g_coords, g_colors,
                    overwrite: bool = False, resort: bool = False) -> None:
    """g_* (+)= gradient of sum(grad_out * sample_forward(...)); `state` is what `sample_forward` returned."""
    pts, n, sws = state
    B = max(int(p.dims.batch), 1)
    ptrs = [_chk(sigmas, "sigmas", (3,)), _chk(coords, "coords", (2,)), _chk(colors, "colors", (3,)),
            _chk(grad_out, "grad_out", (3, n)), _chk(g_sigmas, "grads_sigmas", (3,)),
            _chk(g_coords, "grads_coords", (2,)), _chk(g_colors, "grads_colors", (3,))]
    if grad_out.numel() != B * 3 * n:
        raise RuntimeError("grad_out does not match the points")
    d = _dims_with(p, FLAG_OVERWRITE_GRADS if overwrite else 0)
    with _on(p.device):
        check(lib().gsasr_splat_sample_backward(*ptrs, ctypes.byref(d), p.workspace.data_ptr(), p.workspace.numel(),
                                                pts.data_ptr() if resort else None, n, sws.data_ptr(), sws.numel(),
                                                _stream(p.device)), "gsasr_splat_sample_backward")


def step_sample_forward(gs_parameters: torch.Tensor, step: Optional[torch.Tensor], h: int, w: int, dmax: Optional[float],
                        points: torch.Tensor, scale_modify: Optional[torch.Tensor] = None, default_step_size: float = 1.2):
    """prologue + plan + sampled forward in ONE call: raw `gs_parameters[N,9]` -> `[3,S]` (step size as in `step_forward`)."""
    pp = _chk(gs_parameters, "gs_parameters", (9,))
    ps = None if scale_modify is not None else _chk(step, "step_size")
    if dmax is not None and not (float(dmax) >= 0.0):
        raise RuntimeError("dmax must be >= 0")
    dev = gs_parameters.device
    d = make_dims(gs_parameters.shape[0], h, w, dmax)
    return _step_sample_forward(d, pp, ps, points, dev, scale_modify, default_step_size)


def batch_sample_forward(gs_parameters: torch.Tensor, steps: torch.Tensor, sizes, dmax: Optional[float],
                         points: torch.Tensor):
    """the same for a whole batch: `gs_parameters` [B,N,9], `points` [B,S,2] on each sample's own grid -> `[B,3,S]`."""
    pp = _chk(gs_parameters, "gs_parameters", (9,))
    ps = _chk(steps, "step_sizes")
    if gs_parameters.dim() != 3 or steps.numel() != gs_parameters.shape[0] or len(sizes) != gs_parameters.shape[0]:
        raise RuntimeError("gs_parameters must be [B,N,9] with one step size and one (h,w) per sample")
    if dmax is not None and not (float(dmax) >= 0.0):
        raise RuntimeError("dmax must be >= 0")
    h_max, w_max = max(h for h, _ in sizes), max(w for _, w in sizes)
    d = make_batch_dims(gs_parameters.shape[1], sizes, w_max, h_max, dmax)
    return _step_sample_forward(d, pp, ps, points, gs_parameters.device)


def _step_sample_forward(d: Dims, pp: int, ps, points: torch.Tensor, dev, scale_modify=None, default_step_size: float = 1.2):
    L = lib()
    B = max(int(d.batch), 1)
    if scale_modify is not None:
        psm, stride = _sm_ptr(scale_modify, B)
    pts, n = _points(points, B, dev)
    nbytes = L.gsasr_step_workspace_bytes(ctypes.byref(d))
    if nbytes == 0:
        check(-1, "gsasr_step_workspace_bytes")
    with _on(dev):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        sws = _sample_ws(d, n, dev)
        out = torch.empty((B, 3, n) if B > 1 else (3, n), dtype=torch.float32, device=dev)
        if scale_modify is None:
            check(L.gsasr_step_sample_forward(pp, ps, ctypes.byref(d), ws.data_ptr(), nbytes, pts.data_ptr(), n,
                                              out.data_ptr(), sws.data_ptr(), sws.numel(), _stream(dev)),
                  "gsasr_step_sample_forward")
        else:
            check(L.gsasr_step_sample_forward_sm(pp, psm, stride, float(default_step_size), mismatch_flag(dev).data_ptr(),
                                                 ctypes.byref(d), ws.data_ptr(), nbytes, pts.data_ptr(), n, out.data_ptr(),
                                                 sws.data_ptr(), sws.numel(), _stream(dev)), "gsasr_step_sample_forward_sm")
    return out, Plan(d, ws, dev), (pts, n, sws)


def step_sample_backward(p: Plan, state, gs_parameters: torch.Tensor, step: torch.Tensor,
                         grad_out: torch.Tensor) -> torch.Tensor:
    """sampled backward + prologue backward in ONE call; returns d/d gs_parameters (`[N,9]` or `[B,N,9]`)."""
    pts, n, sws = state
    pp = _chk(gs_parameters, "gs_parameters", (9,))
    ps = None if step is None else _chk(step, "step_size")      # None: the step sizes the forward's prologue used
    pg = _chk(grad_out, "grad_out", (3, n))
    if grad_out.numel() != max(int(p.dims.batch), 1) * 3 * n:
        raise RuntimeError("grad_out does not match the points")
    with _on(p.device):
        gp = torch.empty_like(gs_parameters)
        check(lib().gsasr_step_sample_backward(pp, ps, pg, gp.data_ptr(), ctypes.byref(p.dims), p.workspace.data_ptr(),
                                               p.workspace.numel(), None, n, sws.data_ptr(), sws.numel(),
                                               _stream(p.device)), "gsasr_step_sample_backward")
    return gp


def set_default_cutoff(tau: float) -> None:
    """tau > 0: skip exponent < -tau; tau < 0: never skip; 0 restores the adaptive default ln(N/1e-5)."""
    lib().gsasr_set_default_cutoff(float(tau))


def get_default_cutoff() -> float:
    return float(lib().gsasr_get_default_cutoff())


def plan_cutoff(p: Plan) -> Tuple[float, int]:
    """(tau, K) the windows of plan `p` were built with: for the bounded op under the adaptive default tau is data-derived,
    ln(K / 1e-5) with K = the plan's own bound on how many dmax boxes cover one pixel (include/gsasr_splat.h); K = 0
    otherwise.  Synchronises the stream: for reports and tests."""
    tau, k = ctypes.c_float(0.0), ctypes.c_uint(0)
    with _on(p.device):
        check(lib().gsasr_plan_cutoff(ctypes.byref(p.dims), p.workspace.data_ptr(), p.workspace.numel(), _stream(p.device),
                                      ctypes.byref(tau), ctypes.byref(k)), "gsasr_plan_cutoff")
    return float(tau.value), int(k.value)


def resolve_cutoff(cutoff: float, s: int) -> float:
    """tau a plan with `cutoff` (0 = process default) over `s` Gaussians uses."""
    return float(lib().gsasr_resolve_cutoff(float(cutoff), int(s)))


# ---- kernel choices registered per shape (include/gsasr_splat.h: gsasr_set_kernel_choice; gsasr_amd/tune.py measures them) ----
CHOICE_FLAGS = FLAG_FWD_WIDE | FLAG_FWD_NARROW | FLAG_BWD_TILE | FLAG_BWD_GAUSSIAN | FLAG_BWD_HOME


_N_CHOICES = 0      # registrations made through this module since the last clear (0 = the fused host path skips its lookup)


def kernel_choices_registered() -> bool:
    return _N_CHOICES > 0


def set_kernel_choice(shape: Dims, flags: int, list_cap: int = 0) -> None:
    """Register the kernel choice for plans of `shape` (a Dims as `make_dims` / `make_batch_dims` builds them; only its shape
    fields and FLAG_FORWARD_ONLY are read).  Later calls without an explicit choice of their own follow it."""
    global _N_CHOICES
    check(lib().gsasr_set_kernel_choice(ctypes.byref(shape), int(flags), int(list_cap)), "gsasr_set_kernel_choice")
    _N_CHOICES += 1
    _PLAN_DIMS.clear()      # workspace sizes follow the registered choice
    _STEP_DIMS.clear()


def get_kernel_choice(shape: Dims) -> Optional[Tuple[int, int]]:
    f, c = ctypes.c_uint(0), ctypes.c_int(0)
    if not lib().gsasr_get_kernel_choice(ctypes.byref(shape), ctypes.byref(f), ctypes.byref(c)):
        return None
    return int(f.value), int(c.value)


def clear_kernel_choices() -> None:
    global _N_CHOICES
    lib().gsasr_clear_kernel_choices()
    _N_CHOICES = 0
    _PLAN_DIMS.clear()
    _STEP_DIMS.clear()
