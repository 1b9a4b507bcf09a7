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

Both collectives move <= 32*N bytes per rank (33.5 MB at N = 1M) and grow with the group: fine for one
large image, but they dominate a step whose local work is ~0.1 ms.  When every rank PRODUCES the Gaussians
of its own band (a spatially sharded encoder/decoder, or tiled inference), `BandExchange` replaces both by
a nearest-neighbour exchange: only the Gaussians whose footprint crosses a band edge travel (to the rank
above / below, point-to-point -- which is what xGMI links are), and only their partial gradients come back.
Volume and latency are then independent of the group size.

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
    from ._cabi import FLAG_CUTOFF_CAP as CUTOFF_CAP_FLAG      # an explicit cutoff is an upper bound for the plan's windows

    # ---- packed [N,8] records (BandExchange) ----
    @staticmethod
    def forward_packed(records, h, w, dmax, rows, cutoff=0.0, flags=0):
        from . import _cabi
        plan = _cabi.plan_packed(records, h, w, dmax, rows=rows, cutoff=cutoff, flags=flags)
        slab = torch.empty(rows[1] - rows[0], w, 3, device=records.device, dtype=torch.float32)
        _cabi.forward(plan, slab, overwrite=True)
        return slab, plan

    @staticmethod
    def forward_packed_into(records, slab, h, w, dmax, rows, cutoff=0.0, flags=0, accumulate=False):
        """plan `records` and render them INTO `slab` (stored, or added to what is there): the two-render form of the band
        exchange (`BandExchange(overlap=True)`: own Gaussians first, the halos on top once they have arrived)"""
        from . import _cabi
        plan = _cabi.plan_packed(records, h, w, dmax, rows=rows, cutoff=cutoff, flags=flags)
        _cabi.forward(plan, slab, overwrite=not accumulate)
        return plan

    @staticmethod
    def backward_packed(state, records, grad_slab, g_records):
        from . import _cabi
        _cabi.backward_packed(state, records, grad_slab.contiguous(), g_records, overwrite=True)

    @staticmethod
    def select(own, h, w, dmax, rows, rows_above, rows_below, up, down, up_index, down_index, counts, cutoff=0.0):
        from . import _cabi
        _cabi.band_select(own, h, w, dmax, rows, rows_above, rows_below, up, down, up_index, down_index, counts, cutoff)

    @staticmethod
    def merge(g_own, g_up, g_down, up_index, down_index, counts):
        from . import _cabi
        _cabi.band_merge(g_own, g_up, g_down, up_index, down_index, counts)

    @staticmethod
    def resolve_cutoff(cutoff, s):
        from . import _cabi
        return _cabi.resolve_cutoff(cutoff, s)

    @staticmethod
    def forward(sigmas, coords, colors, h, w, dmax, rows):
        from . import _cabi
        # A proper band that is handed EVERY Gaussian of the image (the broadcast flow) is mostly "dead" Gaussians:
        # the Gaussian-stationary backward would spend a wave on each and, worse, its XCD-contiguous slot order
        # would put all the live ones on one XCD (measured on an eighth of config 4: 1.72 ms vs 2.10 ms for the
        # WHOLE image).  The tile-stationary kernel launches over the band's tiles only.
        band = rows[1] - rows[0] < h
        plan = _cabi.plan(sigmas, coords, colors, h, w, dmax, rows=rows, flags=_cabi.FLAG_BWD_TILE if band else 0)
        slab = torch.empty(rows[1] - rows[0], w, 3, device=sigmas.device, dtype=torch.float32)
        _cabi.forward(plan, slab, overwrite=True)
        return slab, plan

    @staticmethod
    def backward(state, sigmas, coords, colors, grad_slab):
        from . import _cabi
        g = (torch.empty_like(sigmas), torch.empty_like(coords), torch.empty_like(colors))
        _cabi.backward(state, sigmas, coords, colors, grad_slab.contiguous(), *g, overwrite=True)
        return g

    @staticmethod
    def backward_to_packed(state, sigmas, coords, colors, grad_slab, g_packed):
        """the same gradient written as ONE `[N,8]` array: the buffer the collective runs on (no pack / unpack copies)"""
        from . import _cabi
        _cabi.backward_to_packed(state, sigmas, coords, colors, grad_slab.contiguous(), g_packed, overwrite=True)


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
        if hasattr(ctx.backend, "backward_to_packed"):
            # ONE [ceil(N / G) * G, 8] buffer: the backward writes it, the collective reduces it in place, the three
            # gradients are column views of it -- the only allocation of this backward (reduce_gaussian_grads packs, pads,
            # scatters and unpacks: five N x 32-byte copies around the same collective)
            n = sigmas.shape[0]
            world = dist.get_world_size(ctx.group) if (dist.is_available() and dist.is_initialized()) else 1
            per = (n + world - 1) // world
            g = torch.empty(per * world, 8, device=sigmas.device, dtype=sigmas.dtype)
            if per * world > n:
                g[n:].zero_()
            ctx.backend.backward_to_packed(ctx.state, sigmas, coords, colors, grad_slab, g[:n])
            gp = reduce_packed_grads_(g, n, ctx.grad_reduce, ctx.group)
            return gp[:, 0:3], gp[:, 3:5], gp[:, 5:8], None, None, None, None, None, None, None
        g = ctx.backend.backward(ctx.state, sigmas, coords, colors, grad_slab)
        gs, gc, gk = reduce_gaussian_grads(*g, mode=ctx.grad_reduce, group=ctx.group)
        return gs, gc, gk, None, None, None, None, None, None, None


def splat_band(sigmas, coords, colors, h: int, w: int, dmax: Optional[float] = None, group=None,
               grad_reduce: str = "all_reduce", rows: Optional[Tuple[int, int]] = None, backend=None):
    """Render this rank's row band: returns `[r1-r0, w, 3]` (HWC, like `GSCUDA.apply`).

    All ranks must hold identical `sigmas/coords/colors` (see `broadcast_gaussians`).  Backward
    reduces the per-Gaussian gradients across the group with `grad_reduce`:
      "all_reduce" (default)  every rank gets the complete gradient -- right for a replicated decoder and for the
                              `broadcast_gaussians(src)` flow, where only the src rank has an autograd graph behind
                              the Gaussians and needs ALL of their gradient;
      "reduce_scatter"        rank g gets the complete gradient of its `gaussian_slice` rows and ZEROS elsewhere --
                              only for a decoder that is itself sharded by Gaussian slice (1/G of the volume);
      "none"                  the rank's partial gradient, no collective.
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


# ---------------------------------------------------------------------------------------------------
# The same pattern on ONE packed [N,8] buffer (BASELINE config 4 as stated: "Gaussians broadcast once, per-Gaussian
# grads reduce-scatter"): the broadcast's receive buffer IS what the plan reads (GSASR_FLAG_STRIDE8) and the packed
# gradient the backward writes IS the collective's buffer -- no cat / slice copies around either collective.
# ---------------------------------------------------------------------------------------------------
def broadcast_packed(packed: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """ONE broadcast of the `[N,8]` records {sx,sy,rho,x,y,r,g,b} from `src`, in place (non-src ranks pass a buffer of
    the right shape)."""
    if packed.dim() != 2 or packed.shape[1] != 8 or not packed.is_contiguous():
        raise RuntimeError("packed must be a contiguous [N,8] tensor")
    dist.broadcast(packed, src=src, group=group)
    return packed


def reduce_packed_grads_(g: torch.Tensor, n: int, mode: str = "reduce_scatter", group=None) -> torch.Tensor:
    """Sum the ranks' partial gradients held in `g[per*world, 8]` (rows [n:] zero) IN PLACE and return `g[:n]`.
    "reduce_scatter": `reduce_scatter_tensor` with this rank's chunk of `g` itself as the output (the in-place form
    RCCL supports: output = input + rank * count), after which the rows of the other ranks' slices are zeroed --
    32 N / G bytes leave with the reduced values instead of 32 N; "all_reduce": every rank gets everything."""
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if mode == "none" or world == 1:
        return g[:n]
    if mode == "all_reduce":
        dist.all_reduce(g[:n], op=dist.ReduceOp.SUM, group=group)
        return g[:n]
    if mode != "reduce_scatter":
        raise ValueError(f"unknown grad reduction mode {mode!r}")
    rank = dist.get_rank(group)
    per = g.shape[0] // world
    if per * world != g.shape[0] or per * world < n:
        raise RuntimeError("g must hold ceil(n / world) * world rows")
    dist.reduce_scatter_tensor(g[rank * per: (rank + 1) * per], g, op=dist.ReduceOp.SUM, group=group)
    a, b = gaussian_slice(n, rank, world)
    if a > 0:
        g[:a].zero_()
    if b < n:
        g[b:n].zero_()
    return g[:n]


class _BandSplatPacked(Function):
    @staticmethod
    def forward(ctx, packed, h, w, dmax, rows, group, grad_reduce, backend, cutoff):
        band = rows[1] - rows[0] < h
        flags = 0
        if band and backend is HipBackend:      # (proper bands of a replicated set: the tile-stationary backward, see HipBackend.forward)
            from . import _cabi
            flags = _cabi.FLAG_BWD_TILE
        slab, state = backend.forward_packed(packed, h, w, dmax, rows, cutoff, flags=flags) if flags else \
            backend.forward_packed(packed, h, w, dmax, rows, cutoff)
        ctx.save_for_backward(packed)
        ctx.state, ctx.group, ctx.grad_reduce, ctx.backend = state, group, grad_reduce, backend
        return slab

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_slab):
        (packed,) = ctx.saved_tensors
        n = packed.shape[0]
        world = dist.get_world_size(ctx.group) if (dist.is_available() and dist.is_initialized()) else 1
        per = (n + world - 1) // world
        g = torch.empty(per * world, 8, device=packed.device, dtype=packed.dtype)   # the ONE allocation of this backward
        if per * world > n:
            g[n:].zero_()
        ctx.backend.backward_packed(ctx.state, packed, grad_slab, g[:n])
        return reduce_packed_grads_(g, n, ctx.grad_reduce, ctx.group), None, None, None, None, None, None, None, None


def splat_band_packed(packed: torch.Tensor, h: int, w: int, dmax: Optional[float] = None, group=None,
                      grad_reduce: str = "all_reduce", rows: Optional[Tuple[int, int]] = None, backend=None,
                      cutoff: float = 0.0) -> torch.Tensor:
    """`splat_band` for Gaussians held as ONE `[N,8]` tensor (what `broadcast_packed` filled): the plan reads the
    records where they are, the backward writes one `[N,8]` gradient and the collective runs on that buffer in place.
    Same `grad_reduce` modes and the same result as `splat_band(*unpack(packed), ...)`, without its five `[N,8]`-sized
    copies per step."""
    if rows is None:
        if dist.is_available() and dist.is_initialized():
            rows = row_band(h, dist.get_rank(group), dist.get_world_size(group))
        else:
            rows, grad_reduce = (0, h), "none"
    if not (dist.is_available() and dist.is_initialized()):
        grad_reduce = "none"
    return _BandSplatPacked.apply(packed, int(h), int(w), dmax, tuple(rows), group, grad_reduce, backend or HipBackend,
                                  float(cutoff))


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


# ---------------------------------------------------------------------------------------------------
# Band-local (nearest-neighbour) exchange
# ---------------------------------------------------------------------------------------------------
class BandExchange:
    """Buffers and the two point-to-point exchanges of a row-band shard with a SHARDED producer.

    Rank g owns HR rows `row_band(h, g, G)` and holds `n_local` Gaussians (those its decoder produced:
    normally the ones centred in its band, but any assignment works).  Per step:

      forward : `select` (one kernel) lists the own Gaussians whose row footprint -- the plan's window,
                dmax box ∩ support -- reaches above / below the band into fixed-capacity `[cap,8]` buffers
                (NaN = dead padding); ONE batched isend/irecv swaps them with ranks g-1 and g+1; the local
                plan then runs over `records = [own | from_above | from_below]`.
      backward: the local backward writes `g_records` for all three parts; the two halo parts are sent
                back (same batched call, reversed) and `merge` adds them into the owners' gradients.

    Nothing here synchronises the host.  Two conditions make a step incomplete: more than `cap` Gaussians selected
    for one side, or a footprint reaching beyond the adjacent band.  `splat_band_local` poisons its image and its
    gradient with a NaN (device-side, from `counts`) when that happens, so it cannot go unnoticed; `check()` reads
    `counts` (host sync) and raises with the numbers.
    """

    def __init__(self, n_local: int, cap: int, h: int, w: int, dmax: Optional[float], cutoff: float = 0.0,
                 device=None, group=None, backend=None, rank: Optional[int] = None, world: Optional[int] = None,
                 transport: str = "alltoall", overlap: bool = False):
        self.group, self.backend = group, backend or HipBackend
        # overlap = True: TWO renders per band instead of one plan over [own | halos] -- the own Gaussians are planned and
        # splatted while the halos travel, the (at most 2 cap) halo records are planned when they have landed and added on
        # top; in the backward the halo part runs first and its gradients fly home under the own part.  One more (small)
        # plan per direction against an exchange that no kernel waits for.
        self.overlap = bool(overlap)
        if self.overlap and not hasattr(self.backend, "forward_packed_into"):
            raise ValueError("BandExchange(overlap=True) needs a backend with forward_packed_into")
        # how the two neighbour swaps of a step are issued: "alltoall" = ONE `all_to_all_single` with split sizes that are
        # zero for every rank but g-1 / g+1 (RCCL turns it into the same grouped send/recv pairs, but the host pays for one
        # collective call instead of four P2P ops and a coalescing context -- the exchange sits between kernels of tens of
        # microseconds); "p2p" = `batch_isend_irecv`
        if transport not in ("alltoall", "p2p"):
            raise ValueError(f"unknown transport {transport!r}")
        self.transport = transport
        # rank/world default to the process group's; explicit values let one process drive several bands
        # (single-GPU tests, or a host that time-multiplexes bands)
        self.rank = dist.get_rank(group) if rank is None else int(rank)
        self.world = dist.get_world_size(group) if world is None else int(world)
        self.h, self.w, self.dmax, self.cutoff = int(h), int(w), dmax, float(cutoff)
        self.n, self.cap = int(n_local), int(cap)
        # one tau for `select` and the local plan: the default is adaptive in the number of Gaussians, which
        # differs between the two calls (n_local vs n_local + 2 cap)
        resolve = getattr(self.backend, "resolve_cutoff", None)
        if resolve is not None:
            self.cutoff = resolve(self.cutoff, self.n + 2 * self.cap)
        self.plan_flags = getattr(self.backend, "CUTOFF_CAP_FLAG", 0)
        self.rows = row_band(self.h, self.rank, self.world)
        span = lambda r: row_band(self.h, r, self.world)[1] - row_band(self.h, r, self.world)[0]
        self.rows_above = span(self.rank - 1) if self.rank > 0 else 0
        self.rows_below = span(self.rank + 1) if self.rank < self.world - 1 else 0
        f = dict(device=device, dtype=torch.float32)
        n, c = self.n, self.cap
        self.records = torch.full((n + 2 * c, 8), float("nan"), **f)      # [own | from_above | from_below]
        self.g_records = torch.zeros(n + 2 * c, 8, **f)
        # (each direction's two buffers are halves of ONE tensor, in rank order of the peer: what all_to_all_single wants)
        self.send, self.ret = torch.empty(2 * c, 8, **f), torch.zeros(2 * c, 8, **f)
        self.send_up, self.send_down = self.send[:c], self.send[c:]
        self.ret_up, self.ret_down = self.ret[:c], self.ret[c:]
        self.up_index = torch.zeros(c, device=device, dtype=torch.int32)
        self.down_index = torch.zeros(c, device=device, dtype=torch.int32)
        self.counts = torch.zeros(4, device=device, dtype=torch.int32)

    @property
    def own(self) -> torch.Tensor:
        """`[n_local,8]` view the producer writes this rank's Gaussians into."""
        return self.records[: self.n]

    @property
    def from_above(self) -> torch.Tensor:
        return self.records[self.n: self.n + self.cap]

    @property
    def from_below(self) -> torch.Tensor:
        return self.records[self.n + self.cap:]

    def select(self) -> None:
        """fill send_up / send_down (+ indices, counts) from `own`"""
        self.backend.select(self.own, self.h, self.w, self.dmax, self.rows, self.rows_above, self.rows_below,
                            self.send_up, self.send_down, self.up_index, self.down_index, self.counts, self.cutoff)

    def merge(self) -> torch.Tensor:
        """add the returned halo gradients (ret_up / ret_down) into the own part of g_records"""
        self.backend.merge(self.g_records[: self.n], self.ret_up, self.ret_down, self.up_index, self.down_index,
                           self.counts)
        return self.g_records[: self.n]

    def _swap(self, key, to_above, from_above, to_below, from_below, async_op=False):
        """one neighbour swap; async_op: returns the handles to `_wait` on (the transfer then runs beside whatever the
        caller enqueues next: RCCL orders it behind the work already on the stream and `wait` makes the stream wait)"""
        if self.transport == "alltoall":
            if self.world <= 1:
                return []
            hit = self._a2a.get(key) if hasattr(self, "_a2a") else None
            if hit is None:
                c, n = self.cap, self.n
                inp, out = (self.send, self.records[n:]) if key == "fwd" else (self.g_records[n:], self.ret)
                assert inp[:c].data_ptr() == to_above.data_ptr() and out[c:].data_ptr() == from_below.data_ptr()
                lo, hi = (0 if self.rank > 0 else c), (2 * c if self.rank < self.world - 1 else c)
                splits = [0] * self.world
                if self.rank > 0:
                    splits[self.rank - 1] = c
                if self.rank < self.world - 1:
                    splits[self.rank + 1] = c
                if not hasattr(self, "_a2a"):
                    self._a2a = {}
                hit = self._a2a[key] = (out[lo:hi], inp[lo:hi], splits)
            w = dist.all_to_all_single(hit[0], hit[1], hit[2], hit[2], group=self.group, async_op=async_op)
            return [w] if async_op else []
        # the four P2POps of a direction always name the same buffers and peers: built once (this runs twice per step
        # on the host, in front of kernels that take tens of microseconds)
        ops = self._ops.get(key) if hasattr(self, "_ops") else None
        if ops is None:
            ops = []
            if self.rank > 0:
                ops += [dist.P2POp(dist.isend, to_above, self._peer(self.rank - 1), self.group),
                        dist.P2POp(dist.irecv, from_above, self._peer(self.rank - 1), self.group)]
            if self.rank < self.world - 1:
                ops += [dist.P2POp(dist.isend, to_below, self._peer(self.rank + 1), self.group),
                        dist.P2POp(dist.irecv, from_below, self._peer(self.rank + 1), self.group)]
            if not hasattr(self, "_ops"):
                self._ops = {}
            self._ops[key] = ops
        reqs = dist.batch_isend_irecv(ops) if ops else []
        if async_op:
            return reqs
        for r in reqs:
            r.wait()
        return []

    @staticmethod
    def _wait(handles) -> None:
        for h in handles or []:
            h.wait()

    def _peer(self, group_rank: int) -> int:
        return group_rank if self.group is None else dist.get_global_rank(self.group, group_rank)

    def exchange_forward(self) -> torch.Tensor:
        """own Gaussians are in `self.own`; returns `records` with the neighbours' halos in place."""
        self.version = getattr(self, "version", 0) + 1
        self.select()
        self._swap("fwd", self.send_up, self.from_above, self.send_down, self.from_below)
        return self.records

    def exchange_backward(self) -> torch.Tensor:
        """`self.g_records` holds the local backward's output; returns the complete gradients of the own
        Gaussians (`[n_local,8]` view of `g_records`)."""
        n, c = self.n, self.cap
        self._swap("bwd", self.g_records[n:n + c], self.ret_up, self.g_records[n + c:], self.ret_down)
        return self.merge()

    def incomplete(self) -> torch.Tensor:
        """device-side bool scalar: the last `select` dropped Gaussians (n_up or n_down > cap, or n_far > 0).
        `splat_band_local` turns it into a NaN in its outputs, so `check()` is a diagnosis, not a duty."""
        c = self.counts
        return ((c[0] > self.cap) | (c[1] > self.cap) | (c[2] > 0)).reshape(1)

    def check(self) -> Tuple[int, int]:
        """Host-synchronising validity check of the last `exchange_forward`; returns (n_up, n_down)."""
        n_up, n_down, n_far, _ = (int(v) for v in self.counts.tolist())
        if n_up > self.cap or n_down > self.cap:
            raise RuntimeError(f"BandExchange: {max(n_up, n_down)} Gaussians cross a band edge but cap = {self.cap}; "
                               "results of this step are incomplete -- enlarge cap")
        if n_far:
            raise RuntimeError(f"BandExchange: {n_far} Gaussians reach beyond the adjacent band (bands too thin for "
                               "this dmax / cutoff); use broadcast_gaussians + splat_band instead")
        return n_up, n_down


def _poison_if(flag: torch.Tensor, t: torch.Tensor) -> None:
    """t.flat[0] = NaN where `flag` (a device scalar) is set: one element, no host synchronisation.  A band exchange
    that dropped Gaussians (capacity overflow, or a footprint beyond the adjacent band) must not produce a
    plausible-looking image or gradient: the NaN reaches the loss of that very step."""
    first = t.view(-1)[:1]
    first.copy_(torch.where(flag, torch.full_like(first, float("nan")), first))


class _BandLocalSplatOverlap(Function):
    """`splat_band_local` with the exchange hidden behind the own Gaussians' render (BandExchange(overlap=True))"""

    @staticmethod
    def forward(ctx, packed_local, ex):
        if packed_local.data_ptr() != ex.own.data_ptr():
            ex.own.copy_(packed_local)
        ex.version = getattr(ex, "version", 0) + 1
        n, be = ex.n, ex.backend
        ex.select()
        flying = ex._swap("fwd", ex.send_up, ex.from_above, ex.send_down, ex.from_below, async_op=True)
        slab = torch.empty(ex.rows[1] - ex.rows[0], ex.w, 3, device=ex.records.device, dtype=torch.float32)
        # own part under the transfer (windows from the data-derived cutoff below the exchange's conservative tau) ...
        own = be.forward_packed_into(ex.records[:n], slab, ex.h, ex.w, ex.dmax, ex.rows, ex.cutoff, ex.plan_flags, False)
        ex._wait(flying)
        # ... the halos on top (a few thousand records: the conservative tau as given)
        halo = be.forward_packed_into(ex.records[n:], slab, ex.h, ex.w, ex.dmax, ex.rows, ex.cutoff, 0, True)
        ctx.ex, ctx.own, ctx.halo, ctx.version = ex, own, halo, ex.version
        if slab.numel():
            _poison_if(ex.incomplete(), slab)
        return slab

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_slab):
        ex = ctx.ex
        if ex.version != ctx.version:
            raise RuntimeError("BandExchange was used by another splat_band_local forward before this backward ran; "
                               "use one BandExchange per forward that is in flight (e.g. per accumulation micro-step)")
        n, c, be = ex.n, ex.cap, ex.backend
        grad_slab = grad_slab.contiguous()
        # the halo part first: its gradients travel home while the own part is differentiated
        be.backward_packed(ctx.halo, ex.records[n:], grad_slab, ex.g_records[n:])
        flying = ex._swap("bwd", ex.g_records[n:n + c], ex.ret_up, ex.g_records[n + c:], ex.ret_down, async_op=True)
        be.backward_packed(ctx.own, ex.records[:n], grad_slab, ex.g_records[:n])
        ex._wait(flying)
        g = ex.merge().clone()
        if g.numel():
            _poison_if(ex.incomplete(), g)
        return g, None


class _BandLocalSplat(Function):
    @staticmethod
    def forward(ctx, packed_local, ex):
        if packed_local.data_ptr() != ex.own.data_ptr():
            ex.own.copy_(packed_local)
        records = ex.exchange_forward()
        # (ex.cutoff is the conservative tau the selection used; the plan may build its windows with the data-derived one below it)
        # (a backend that declares CUTOFF_CAP_FLAG takes `flags` by keyword; one that does not is never handed the argument)
        slab, state = ex.backend.forward_packed(records, ex.h, ex.w, ex.dmax, ex.rows, ex.cutoff, flags=ex.plan_flags) if ex.plan_flags \
            else ex.backend.forward_packed(records, ex.h, ex.w, ex.dmax, ex.rows, ex.cutoff)
        ctx.ex, ctx.state, ctx.version = ex, state, ex.version
        if slab.numel():
            _poison_if(ex.incomplete(), slab)
        return slab

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_slab):
        ex = ctx.ex
        if ex.version != ctx.version:
            # the exchange's buffers (records, indices, counts) now belong to a later forward: gradients computed from
            # them would silently be those of the wrong Gaussians
            raise RuntimeError("BandExchange was used by another splat_band_local forward before this backward ran; "
                               "use one BandExchange per forward that is in flight (e.g. per accumulation micro-step)")
        ex.backend.backward_packed(ctx.state, ex.records, grad_slab, ex.g_records)
        g = ex.exchange_backward().clone()
        if g.numel():
            _poison_if(ex.incomplete(), g)
        return g, None


def splat_band_local(packed_local: torch.Tensor, ex: BandExchange) -> torch.Tensor:
    """Render this rank's row band from the Gaussians it produced (`[n_local,8]` records) plus the
    neighbours' halos; differentiable w.r.t. `packed_local`, whose gradient includes what the
    neighbouring bands contribute.  Returns `[r1-r0, w, 3]`."""
    return (_BandLocalSplatOverlap if ex.overlap else _BandLocalSplat).apply(packed_local, ex)
