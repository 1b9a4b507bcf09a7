"""Multi-GPU HR row-band shard of the rasterizer (not present in the reference, whose
rasterizer is rank-local -- TrainTestGSASR/basicsr/models/base_model.py:96-99 is plain DDP).

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm).  The path shards
because every output pixel is an independent sum and every Gaussian gradient is a sum over pixels:

  * rank g owns HR rows [g*H/G, (g+1)*H/G) -- a contiguous slab of `img` / `grad_img`;
  * forward : the Gaussians ([N,8] packed: 32 B each) are broadcast ONCE from the rank that ran the
              decoder; each rank renders its slab (no data-path collective);
  * backward: each rank produces partial per-Gaussian gradients from its slab, then ONE collective:
              `reduce_scatter_tensor` (rank g keeps the gradients of Gaussians [g*N/G,(g+1)*N/G)), or
              `all_reduce` when every rank needs all of them (replicated decoder).

Both collectives move <= 32*N bytes (33.5 MB at N = 1M): on 7 x ~153 GB/s xGMI links that is tens of
microseconds, so a single un-bucketed call per step is the right granularity.

The local rasterizer is pluggable only so the collective plumbing can be tested with gloo on CPU
(tests/ inject the oracle); the default and only product backend is the HIP library.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist
from torch.autograd import Function
from torch.autograd.function import once_differentiable


def row_band(h: int, rank: int, world: int) -> Tuple[int, int]:
    """HR rows [r0, r1) owned by `rank` (bands differ by at most one row)."""
    return (rank * h) // world, ((rank + 1) * h) // world


def gaussian_slice(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Gaussians whose reduced gradients land on `rank` after reduce_scatter (equal, padded chunks)."""
    per = (n + world - 1) // world
    return min(n, rank * per), min(n, (rank + 1) * per)


def pack(sigmas: torch.Tensor, coords: torch.Tensor, colors: torch.Tensor) -> torch.Tensor:
    return torch.cat([sigmas, coords, colors], dim=1).contiguous()  # [N,8]


def unpack(p: torch.Tensor):
    return p[:, 0:3].contiguous(), p[:, 3:5].contiguous(), p[:, 5:8].contiguous()


def broadcast_gaussians(sigmas, coords, colors, src: int = 0, group=None):
    """Broadcast {sigmas, coords, colors} from `src` as one [N,8] message. Non-src ranks pass tensors
    of the right shape (contents ignored)."""
    p = pack(sigmas, coords, colors)
    dist.broadcast(p, src=src, group=group)
    return unpack(p)


class HipBackend:
    """Local band rasterizer = libgsasr_splat.so through the C ABI (the product path)."""

    @staticmethod
    def forward(sigmas, coords, colors, h, w, dmax, rows):
        from . import _cabi
        plan = _cabi.plan(sigmas, coords, colors, h, w, dmax, rows=rows)
        slab = torch.empty(rows[1] - rows[0], w, 3, device=sigmas.device, dtype=torch.float32)
        _cabi.forward(plan, slab, overwrite=True)
        return slab, plan

    @staticmethod
    def backward(state, sigmas, coords, colors, grad_slab):
        from . import _cabi
        g = (torch.empty_like(sigmas), torch.empty_like(coords), torch.empty_like(colors))
        _cabi.backward(state, sigmas, coords, colors, grad_slab.contiguous(), *g, overwrite=True)
        return g


def reduce_gaussian_grads(g_sigmas, g_coords, g_colors, mode: str = "reduce_scatter", group=None):
    """Sum the per-rank partial gradients. Returns full-shape tensors; with "reduce_scatter" only this
    rank's `gaussian_slice` rows are the reduced values and all other rows are zero."""
    world = dist.get_world_size(group)
    n = g_sigmas.shape[0]
    packed = pack(g_sigmas, g_coords, g_colors)
    if mode == "none" or world == 1:
        pass
    elif mode == "all_reduce":
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    elif mode == "reduce_scatter":
        rank = dist.get_rank(group)
        per = (n + world - 1) // world
        padded = packed.new_zeros(per * world, 8)
        padded[:n] = packed
        mine = packed.new_empty(per, 8)
        dist.reduce_scatter_tensor(mine, padded, op=dist.ReduceOp.SUM, group=group)
        a, b = gaussian_slice(n, rank, world)
        packed = torch.zeros_like(packed)
        packed[a:b] = mine[: b - a]
    else:
        raise ValueError(f"unknown grad reduction mode {mode!r}")
    return unpack(packed)


class _BandSplat(Function):
    @staticmethod
    def forward(ctx, sigmas, coords, colors, h, w, dmax, rows, group, grad_reduce, backend):
        slab, state = backend.forward(sigmas, coords, colors, h, w, dmax, rows)
        ctx.save_for_backward(sigmas, coords, colors)
        ctx.state, ctx.group, ctx.grad_reduce, ctx.backend = state, group, grad_reduce, backend
        return slab

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_slab):
        sigmas, coords, colors = ctx.saved_tensors
        g = ctx.backend.backward(ctx.state, sigmas, coords, colors, grad_slab)
        gs, gc, gk = reduce_gaussian_grads(*g, mode=ctx.grad_reduce, group=ctx.group)
        return gs, gc, gk, None, None, None, None, None, None, None


def splat_band(sigmas, coords, colors, h: int, w: int, dmax: Optional[float] = None, group=None,
               grad_reduce: str = "reduce_scatter", rows: Optional[Tuple[int, int]] = None, backend=None):
    """Render this rank's row band: returns `[r1-r0, w, 3]` (HWC, like `GSCUDA.apply`).

    All ranks must hold identical `sigmas/coords/colors` (see `broadcast_gaussians`).  Backward
    reduces the per-Gaussian gradients across the group with `grad_reduce` in
    {"reduce_scatter", "all_reduce", "none"}.
    """
    if rows is None:
        if dist.is_available() and dist.is_initialized():
            rows = row_band(h, dist.get_rank(group), dist.get_world_size(group))
        else:
            rows, grad_reduce = (0, h), "none"
    if not (dist.is_available() and dist.is_initialized()):
        grad_reduce = "none"
    return _BandSplat.apply(sigmas, coords, colors, int(h), int(w), dmax, tuple(rows), group, grad_reduce,
                            backend or HipBackend)


def gather_image(slab: torch.Tensor, h: int, group=None) -> torch.Tensor:
    """all_gather the row bands into the full `[h, w, 3]` image (bands may differ by one row)."""
    world = dist.get_world_size(group)
    w = slab.shape[1]
    maxrows = max(row_band(h, r, world)[1] - row_band(h, r, world)[0] for r in range(world))
    buf = slab.new_zeros(maxrows, w, 3)
    buf[: slab.shape[0]] = slab
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat([parts[r][: row_band(h, r, world)[1] - row_band(h, r, world)[0]] for r in range(world)], dim=0)
