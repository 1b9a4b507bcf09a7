"""Drop-in for the reference's pybind11 module `gscuda`.

The reference builds one module named `gscuda` per variant (utils/gs_cuda/gswrapper.cpp:73-80 via
`load(name="gscuda")`, utils/gs_cuda_dmax/gswrapper.cpp:75-82 via setup_gscuda.py:6-21) exporting

    gs_render(sigmas, coords, colors, rendered_img, s, h, w, c[, dmax]) -> None
    gs_render_backward(sigmas, coords, colors, grads, grads_sigmas, grads_coords, grads_colors,
                       s, h, w, c[, dmax]) -> None

This module accepts both arities (no `dmax` = the unbounded gs_cuda op) and forwards to the
reference-shaped C entry points of libgsasr_splat.so on the CURRENT torch stream (the reference
launches on the null stream, gs.cu:82).  Same contracts: fp32 contiguous CUDA tensors or
RuntimeError; `rendered_img` is accumulated into; the dmax backward adds into the caller's
(zero-initialised) outputs, the unbounded backward overwrites them.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _cabi


def gs_render(sigmas, coords, colors, rendered_img, s: int, h: int, w: int, c: int,
              dmax: Optional[float] = None) -> None:
    ps, pc, pk = _cabi._chk(sigmas, "sigmas"), _cabi._chk(coords, "coords"), _cabi._chk(colors, "colors")
    pi = _cabi._chk(rendered_img, "rendered_img")
    dev = sigmas.device
    L = _cabi.lib()
    with torch.cuda.device(dev):
        st = _cabi._stream(dev)
        if dmax is None:
            rc = L.gsasr_gs_render(ps, pc, pk, pi, int(s), int(h), int(w), int(c), st)
        else:
            rc = L.gsasr_gs_render_dmax(ps, pc, pk, pi, int(s), int(h), int(w), int(c), float(dmax), st)
    _cabi.check(rc, "gs_render")


def gs_render_backward(sigmas, coords, colors, grads, grads_sigmas, grads_coords, grads_colors, s: int, h: int,
                       w: int, c: int, dmax: Optional[float] = None) -> None:
    ptrs = [_cabi._chk(t, n) for t, n in ((sigmas, "sigmas"), (coords, "coords"), (colors, "colors"),
                                          (grads, "grads"), (grads_sigmas, "grads_sigmas"),
                                          (grads_coords, "grads_coords"), (grads_colors, "grads_colors"))]
    dev = sigmas.device
    L = _cabi.lib()
    with torch.cuda.device(dev):
        st = _cabi._stream(dev)
        if dmax is None:
            rc = L.gsasr_gs_render_backward(*ptrs, int(s), int(h), int(w), int(c), st)
        else:
            rc = L.gsasr_gs_render_backward_dmax(*ptrs, int(s), int(h), int(w), int(c), float(dmax), st)
    _cabi.check(rc, "gs_render_backward")
