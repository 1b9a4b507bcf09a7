"""numpy front-end of the CPU oracle (oracle/gs_ref.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (gsasr_amd/) never does.  Parity status: pinned against tests/golden/*.npz, which were
produced from the reference's own `torch_version` (utils/gs_cuda/check.py:4-27,
utils/gs_cuda_dmax/check.py:4-31) -- see tests/golden/make_golden.py and tests/test_oracle.py.

Functions mirror the launcher prototypes of the reference (utils/gs_cuda/gs.h:1-24,
utils/gs_cuda_dmax/gs.h:1-26): raw fp32 arrays `sigmas[s,3]`, `coords[s,2]`, `colors[s,3]`,
image `[h,w,3]`.  `dmax=None` selects the unbounded gs_cuda variant, a float the gs_cuda_dmax one.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgs_ref.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)
_D = ctypes.POINTER(ctypes.c_double)


def build(force: bool = False) -> str:
    """Compile oracle/gs_ref.c with gcc (see oracle/Makefile). Returns the library path."""
    src = os.path.join(_HERE, "gs_ref.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgs_ref.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_LIB_PATH)
        i, f = ctypes.c_int, ctypes.c_float
        lib.gsref_forward_f32.argtypes = [_F, _F, _F, _F, i, i, i, i, f, i, i, i, i]
        lib.gsref_backward_f32.argtypes = [_F, _F, _F, _F, _F, _F, _F, i, i, i, i, f, i, i, i, i]
        lib.gsref_forward_f64.argtypes = [_F, _F, _F, _D, i, i, i, f, i, i]
        lib.gsref_backward_f64.argtypes = [_F, _F, _F, _F, _D, _D, _D, i, i, i, f, i, i]
        lib.gsref_num_threads.restype = i
        lib.gsref_set_num_threads.argtypes = [i]
        for fn in (lib.gsref_forward_f32, lib.gsref_backward_f32, lib.gsref_forward_f64,
                   lib.gsref_backward_f64, lib.gsref_set_num_threads):
            fn.restype = None
        _lib = lib
    return _lib


def num_threads() -> int:
    return int(_load().gsref_num_threads())


def set_num_threads(n: int) -> None:
    _load().gsref_set_num_threads(int(n))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_F)


def _dp(a: np.ndarray):
    return a.ctypes.data_as(_D)


def _rows(h: int, rows: Optional[Tuple[int, int]]) -> Tuple[int, int]:
    r0, r1 = (0, h) if rows is None else rows
    assert 0 <= r0 <= r1 <= h
    return int(r0), int(r1)


def forward_f32(sigmas, coords, colors, h: int, w: int, dmax: Optional[float] = None,
                img: Optional[np.ndarray] = None, use_fma: bool = False,
                rows: Optional[Tuple[int, int]] = None) -> np.ndarray:
    """fp32 restatement of `_gs_render` (accumulates into `img` if given, like the kernels)."""
    sigmas, coords, colors = _f32(sigmas), _f32(coords), _f32(colors)
    s = sigmas.shape[0]
    r0, r1 = _rows(h, rows)
    if img is None:
        img = np.zeros((r1 - r0, w, 3), np.float32)
    assert img.dtype == np.float32 and img.flags.c_contiguous and img.shape == (r1 - r0, w, 3)
    variant = 0 if dmax is None else 1
    _load().gsref_forward_f32(_fp(sigmas), _fp(coords), _fp(colors), _fp(img), s, h, w, 3,
                              0.0 if dmax is None else float(dmax), variant, int(use_fma), r0, r1)
    return img


def backward_f32(sigmas, coords, colors, grads, dmax: Optional[float] = None, use_fma: bool = False,
                 h: Optional[int] = None, rows: Optional[Tuple[int, int]] = None):
    """fp32 restatement of `_gs_render_backward`; returns (g_sigmas, g_coords, g_colors)."""
    sigmas, coords, colors, grads = _f32(sigmas), _f32(coords), _f32(colors), _f32(grads)
    s = sigmas.shape[0]
    hh, w, c = grads.shape
    h = hh if h is None else h
    r0, r1 = _rows(h, rows)
    assert r1 - r0 == hh and c == 3
    gs, gc, gk = (np.zeros((s, 3), np.float32), np.zeros((s, 2), np.float32),
                  np.zeros((s, 3), np.float32))
    variant = 0 if dmax is None else 1
    _load().gsref_backward_f32(_fp(sigmas), _fp(coords), _fp(colors), _fp(grads), _fp(gs), _fp(gc),
                               _fp(gk), s, h, w, 3, 0.0 if dmax is None else float(dmax), variant,
                               int(use_fma), r0, r1)
    return gs, gc, gk


def forward_f64(sigmas, coords, colors, h: int, w: int, dmax: Optional[float] = None,
                rows: Optional[Tuple[int, int]] = None) -> np.ndarray:
    """Double-precision truth with reference-exact box decisions; returns float64 [rows,w,3]."""
    sigmas, coords, colors = _f32(sigmas), _f32(coords), _f32(colors)
    r0, r1 = _rows(h, rows)
    img = np.zeros((r1 - r0, w, 3), np.float64)
    _load().gsref_forward_f64(_fp(sigmas), _fp(coords), _fp(colors), _dp(img), sigmas.shape[0], h, w,
                              -1.0 if dmax is None else float(dmax), r0, r1)
    return img


def backward_f64(sigmas, coords, colors, grads, dmax: Optional[float] = None,
                 h: Optional[int] = None, rows: Optional[Tuple[int, int]] = None):
    sigmas, coords, colors, grads = _f32(sigmas), _f32(coords), _f32(colors), _f32(grads)
    s = sigmas.shape[0]
    hh, w, c = grads.shape
    h = hh if h is None else h
    r0, r1 = _rows(h, rows)
    assert r1 - r0 == hh and c == 3
    gs, gc, gk = np.zeros((s, 3)), np.zeros((s, 2)), np.zeros((s, 3))
    _load().gsref_backward_f64(_fp(sigmas), _fp(coords), _fp(colors), _fp(grads), _dp(gs), _dp(gc),
                               _dp(gk), s, h, w, -1.0 if dmax is None else float(dmax), r0, r1)
    return gs, gc, gk
