"""Kernel choice measured on the caller's own Gaussians (opt-in).

The library picks its kernels from the SHAPE of a problem -- pixels per Gaussian, image size -- because the window sizes
that really decide live on the device and no entry point synchronises to read them.  Those rules were drawn on GSASR-shaped
synthetic Gaussians ("about one LR pixel": gsasr_amd/synthetic.py).  On other size distributions another combination of

    forward  : 8 x 16 or 16 x 16-px sub-tiles           (FLAG_FWD_NARROW / FLAG_FWD_WIDE)
    backward : Gaussian-, tile-stationary or home-tile   (FLAG_BWD_GAUSSIAN / FLAG_BWD_TILE / FLAG_BWD_HOME)
    lists    : the plan's tile lists, or the search     (list_cap > 0 / < 0)

can be 5..45% faster (profiles/history/r05_policy_regret.txt).  `tune()` times the combinations on the tensors it is given --
plan + forward (+ backward), a few repetitions each in three rounds after a 30 ms warm-up -- and registers the winner
for the shape in the C library (`gsasr_set_kernel_choice`, include/gsasr_splat.h), so that every later call of that shape
through any entry point (the GSCUDA drop-in, the C++ autograd node, the C ABI itself) follows it.  All combinations compute
the same sums in another order (tests/test_tune.py).

    from gsasr_amd import tune
    tune.tune(sigmas, coords, colors, H, W, dmax=0.1)          # once per shape, e.g. on the first batch

or `GSASR_AMD_AUTOTUNE=1` in the environment: the GSCUDA drop-in tunes each new shape on its first call (like
`torch.backends.cudnn.benchmark`; that first call synchronises and takes ~10 steps' worth of time; never during stream capture).
Nothing here is on the default path: without a call or the variable the library's own rules apply.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch

from . import _cabi

AUTOTUNE = _cabi._AUTOTUNE
WARM_MS = 30.0          # milliseconds of the default combination run before anything is timed (clock ramp)
MIN_GAIN = 0.03         # a combination replaces the library's own choice only when it is at least this much faster
_SEEN = set()           # shapes the autotune hook has handled (tuned or declined) in this process


@dataclass
class TuneResult:
    name: str                       # winning combination ("default" = the library's own rule stays)
    flags: int
    list_cap: int
    ms: Dict[str, float] = field(default_factory=dict)      # combination -> milliseconds per plan + forward (+ backward)
    registered: bool = False


def default_list_capacity(s: int, w: int, rows: int, wide: bool) -> int:
    """entries per tile the library would give a plan with lists (tl_cap_for, gsasr_amd/csrc/splat_common.h): four times what a
    tile of LR-pixel sized Gaussians collects, + 64"""
    rho = max(float(s) / (float(w) * float(max(rows, 1))), 1e-9)
    e = 5.0 / math.sqrt(rho)
    want = 4.0 * rho * (32.0 + e) * ((32.0 if wide else 16.0) + e) + 64.0
    return int(min(max(want, 64.0), 65536.0))


def _shape_key(s, h, w, dmax, rows, cutoff, forward_only):
    return (int(s), int(h), int(w), None if dmax is None else float(dmax), rows, float(cutoff), bool(forward_only))


def candidates(s: int, w: int, rows: int, backward: bool):
    """(name, flags, list_cap) of every combination, the library's own choice first"""
    out = [("default", 0, 0)]
    for fw, ff in (("narrow", _cabi.FLAG_FWD_NARROW), ("wide", _cabi.FLAG_FWD_WIDE)):
        for bw, bf in ((("gaussian", _cabi.FLAG_BWD_GAUSSIAN), ("tile", _cabi.FLAG_BWD_TILE), ("home", _cabi.FLAG_BWD_HOME)) if backward else (("", 0),)):
            for lists in (False, True):
                cap = default_list_capacity(s, w, rows, fw == "wide") if lists else -1
                out.append(("-".join(x for x in (fw, bw, "lists" if lists else "search") if x), ff | bf, cap))
    return out


def _measure(cands, step, iters: int, rounds: int) -> TuneResult:
    """time `step(flags, list_cap)` for every (name, flags, list_cap) of `cands` (the first one is the default)"""
    res = TuneResult(cands[0][0], cands[0][1], cands[0][2])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # The GPU's clocks follow its load: a burst of a few steps after an idle gap runs slower than the same steps a moment
    # later, which would favour whichever combination is measured last.  So: warm up on the default for WARM_MS first, keep
    # the queue busy from then on (one synchronisation per measurement), go round the combinations `rounds` times and keep
    # each one's fastest round.
    a.record()
    t = 0.0
    while t < WARM_MS:
        for _ in range(4):
            step(cands[0][1], cands[0][2])
        b.record()
        b.synchronize()
        t = a.elapsed_time(b)
    alive = {}
    for name, flags, cap in cands:
        try:
            step(flags, cap)                        # (first use of a layout: workspace allocation)
            alive[name] = (flags, cap)
        except RuntimeError:
            pass                                    # a combination this shape does not admit
    n = max(1, iters)
    for _ in range(max(1, rounds)):
        for name, (flags, cap) in alive.items():
            a.record()
            for _ in range(n):
                step(flags, cap)
            b.record()
            b.synchronize()
            ms = a.elapsed_time(b) / n
            res.ms[name] = min(ms, res.ms.get(name, ms))
    return res


def _pick(res: TuneResult, cands) -> None:
    if cands[0][0] not in res.ms:
        raise RuntimeError("tune: the default combination failed")
    best = min(res.ms, key=res.ms.get)
    if best != cands[0][0] and res.ms[best] <= (1.0 - MIN_GAIN) * res.ms[cands[0][0]]:
        res.name = best
        res.flags, res.list_cap = next((f, c) for nm, f, c in cands if nm == best)


def tune(sigmas: torch.Tensor, coords: torch.Tensor, colors: torch.Tensor, h: int, w: int, dmax: Optional[float],
         *, backward: bool = True, rows: Optional[Tuple[int, int]] = None, cutoff: float = 0.0, iters: int = 3,
         register: bool = True, grad: Optional[torch.Tensor] = None, forward_only_plan: Optional[bool] = None,
         rounds: int = 3) -> TuneResult:
    """Time every kernel combination on these Gaussians and (by default) register the fastest for the shape.

    `backward=False`: only plan + forward are timed, and the result is registered for plans made with FLAG_FORWARD_ONLY
    (`forward_only_plan=False`: for full plans whose backward never runs -- the GSCUDA drop-in under no_grad).  `grad`: the
    upstream gradient to time the backward with ([rows, w, 3]; default: ones).  Synchronises the device; do not call during
    stream capture."""
    if not sigmas.is_cuda:
        raise RuntimeError("tune() measures on the GPU: CUDA tensors required")
    # (the timing events must be recorded on the stream the kernels run on: the tensors' device, not whatever is current)
    with torch.cuda.device(sigmas.device):
        return _tune_on_device(sigmas, coords, colors, h, w, dmax, backward, rows, cutoff, iters, register, grad,
                               forward_only_plan, rounds)


def _tune_on_device(sigmas, coords, colors, h, w, dmax, backward, rows, cutoff, iters, register, grad, forward_only_plan,
                    rounds) -> TuneResult:
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("tune() synchronises: not during stream capture")
    s = int(sigmas.shape[0])
    r0, r1 = (0, int(h)) if rows is None else rows
    nrows = r1 - r0
    if forward_only_plan is None:
        forward_only_plan = not backward
    if backward and forward_only_plan:
        raise RuntimeError("a FLAG_FORWARD_ONLY plan has no backward to time")
    base = _cabi.FLAG_FORWARD_ONLY if forward_only_plan else 0
    shape = _cabi.make_dims(s, h, w, dmax, rows, cutoff, base)
    prev = _cabi.get_kernel_choice(shape)
    if prev is not None:            # measure the library's own rule as "default", not an earlier registration
        _cabi.set_kernel_choice(shape, 0, 0)
    dev = sigmas.device
    cands = candidates(s, w, nrows, backward)
    with torch.no_grad():
        img = torch.empty(nrows, w, 3, device=dev)
        if backward:
            g = grad if grad is not None else torch.ones(nrows, w, 3, device=dev)
            gs, gc, gk = torch.empty_like(sigmas), torch.empty_like(coords), torch.empty_like(colors)

        def step(flags, cap):
            p = _cabi.plan(sigmas, coords, colors, h, w, dmax, rows, cutoff, base | flags, cap)
            _cabi.forward(p, img, overwrite=True)
            if backward:
                _cabi.backward(p, sigmas, coords, colors, g, gs, gc, gk, overwrite=True)

        res = _measure(cands, step, iters, rounds)
    if prev is not None and not register:
        _cabi.set_kernel_choice(shape, *prev)
    _pick(res, cands)
    if register:
        # The library's own rule won: nothing to register (the C table holds 256 shapes; ragged crop sizes reach that soon),
        # an earlier registration of the shape is reset to "no choice".
        if res.name != cands[0][0] or prev is not None:
            _cabi.set_kernel_choice(shape, res.flags, res.list_cap)
        res.registered = True
    _SEEN.add(_shape_key(s, h, w, dmax, rows, cutoff, forward_only_plan))
    return res


def _fused_candidates(s: int, w: int, rows: int, default_tile: bool):
    """backward kernel x lists for the fused entry points (their forward is the 8 x 16 kernel or, on single images, the library's
    own choice): the first entry is what gsasr_amd.gaussian_splatting does untuned"""
    T, G, Hm = _cabi.FLAG_BWD_TILE, _cabi.FLAG_BWD_GAUSSIAN, _cabi.FLAG_BWD_HOME
    cap = default_list_capacity(s, w, rows, False)
    return [("default", T if default_tile else G, 0), ("gaussian-search", G, -1), ("gaussian-lists", G, cap),
            ("tile-search", T, -1), ("tile-lists", T, cap), ("home-search", Hm, -1), ("home-lists", Hm, cap)]


def _tune_fused(shape, cands, run, iters, rounds, register) -> TuneResult:
    prev = _cabi.get_kernel_choice(shape)
    state = {"cap": None}

    def step(flags, cap):
        if cap != state["cap"]:                     # the fused calls carry no list_cap of their own: the registry supplies it
            _cabi.set_kernel_choice(shape, 0, cap)
            state["cap"] = cap
        run(flags)

    try:
        with torch.no_grad():
            res = _measure(cands, step, iters, rounds)
    finally:
        _cabi.set_kernel_choice(shape, *(prev if prev is not None else (0, 0)))
    _pick(res, cands)
    if register:
        _cabi.set_kernel_choice(shape, res.flags if res.name != "default" else 0, res.list_cap)
        res.registered = True
    return res


def tune_batch(gs_parameters: torch.Tensor, steps: torch.Tensor, sizes, dmax: Optional[float], *, iters: int = 3, rounds: int = 3,
               register: bool = True) -> TuneResult:
    """The batched training step (`generate_2D_gaussian_splatting_batch`: `gs_parameters[B,N,9]`, per-sample step sizes and
    `(h, w)`): time prologue + plan + forward + backward for {Gaussian-, tile-stationary backward} x {lists, search} on THIS
    batch and register the winner for the canvas shape; the fused host path follows the registration from then on."""
    from .gaussian_splatting import _tile_backward
    if not gs_parameters.is_cuda or torch.cuda.is_current_stream_capturing():
        raise RuntimeError("tune_batch() measures on the GPU, outside stream capture")
    B, n = gs_parameters.shape[0], gs_parameters.shape[1]
    h_max, w_max = max(h for h, _ in sizes), max(w for _, w in sizes)
    shape = _cabi.make_batch_dims(n, sizes, w_max, h_max, dmax)
    gp = gs_parameters.detach().contiguous()
    grad = torch.ones(B, 3, h_max, w_max, device=gp.device)

    def run(flags):
        _, plan = _cabi.batch_forward(gp, steps, sizes, dmax, _cabi.FLAG_CHW_GRAD | flags)
        _cabi.batch_backward(plan, gp, None, grad, chw=True)

    default_tile = _tile_backward(sum(h * w for h, w in sizes), B * n)
    return _tune_fused(shape, _fused_candidates(B * n, w_max, shape.slot * B, default_tile), run, iters, rounds, register)


def tune_step(gs_parameters: torch.Tensor, step: torch.Tensor, h: int, w: int, dmax: Optional[float], *, iters: int = 3,
              rounds: int = 3, register: bool = True) -> TuneResult:
    """The fused single-image step (`generate_2D_gaussian_splatting_step` on raw decoder output `[N,9]`): as `tune_batch`."""
    from .gaussian_splatting import _tile_backward
    if not gs_parameters.is_cuda or torch.cuda.is_current_stream_capturing():
        raise RuntimeError("tune_step() measures on the GPU, outside stream capture")
    n = gs_parameters.shape[0]
    shape = _cabi.make_dims(n, h, w, dmax)
    gp = gs_parameters.detach().contiguous()
    grad = torch.ones(3, int(h), int(w), device=gp.device)

    def run(flags):
        _, plan = _cabi.step_forward(gp, step, h, w, dmax, _cabi.FLAG_CHW_GRAD | flags)
        _cabi.step_backward(plan, gp, None, grad, chw=True)

    return _tune_fused(shape, _fused_candidates(n, w, h, _tile_backward(h * w, n)), run, iters, rounds, register)


def autotune_hook(sigmas: torch.Tensor, coords: torch.Tensor, colors: torch.Tensor, h: int, w: int, dmax: Optional[float],
                  backward: bool) -> None:
    """called by the drop-in's forward when GSASR_AMD_AUTOTUNE=1: tune a shape the first time it is seen"""
    key = _shape_key(sigmas.shape[0], h, w, dmax, None, 0.0, False)     # (the drop-in's plans are full plans)
    if key in _SEEN:
        return
    if not sigmas.is_cuda or sigmas.shape[0] == 0 or torch.cuda.is_current_stream_capturing():
        return
    _SEEN.add(key)
    try:
        tune(sigmas.detach(), coords.detach(), colors.detach(), h, w, dmax, backward=backward, forward_only_plan=False)
    except RuntimeError as e:      # never raise into the user's forward: the library's rule serves an untuned shape
        import warnings
        warnings.warn(f"gsasr_amd autotune skipped a shape: {e}")


def reset() -> None:
    """forget every registered choice (and what the autotune hook has seen)"""
    _SEEN.clear()
    _cabi.clear_kernel_choices()
