"""Multi-process CPU tests (gloo, world_size 2 and 3) of the row-band shard's collective plumbing
(gsasr_amd/shard.py).  The local rasterizer is the CPU oracle injected as a backend -- on GPU ranks the
same code path runs the HIP backend over RCCL (covered by tests/test_hip_parity.py::test_row_band_*)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gsasr_amd import shard, synthetic


class OracleBackend:
    """CPU stand-in for HipBackend: same forward/backward contract, computed by oracle/gs_ref.c (f64 truth)."""

    @staticmethod
    def forward(sigmas, coords, colors, h, w, dmax, rows):
        from oracle import gs_oracle
        img = gs_oracle.forward_f64(sigmas.numpy(), coords.numpy(), colors.numpy(), h, w, dmax, rows=rows)
        return torch.from_numpy(img).float(), (h, w, dmax, rows)

    @staticmethod
    def backward(state, sigmas, coords, colors, grad_slab):
        from oracle import gs_oracle
        h, w, dmax, rows = state
        g = gs_oracle.backward_f64(sigmas.numpy(), coords.numpy(), colors.numpy(), grad_slab.contiguous().numpy(),
                                   dmax, h=h, rows=rows)
        return tuple(torch.from_numpy(a).float() for a in g)


class OracleBackendPackedOut(OracleBackend):
    """the same with the gradient written as one [N,8] array (HipBackend.backward_to_packed): `splat_band` then reduces that
    buffer in place and returns column views of it instead of packing / padding / unpacking around the collective"""

    @staticmethod
    def backward_to_packed(state, sigmas, coords, colors, grad_slab, g_packed):
        g = OracleBackend.backward(state, sigmas, coords, colors, grad_slab)
        g_packed.copy_(shard.pack(*g))


BACKENDS = {"three-tensors": OracleBackend, "packed-out": OracleBackendPackedOut}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out, backend="three-tensors"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sig, xy, col, H, W = synthetic.kernel_inputs(12, 10, 3.0, seed=5)
        n = sig.shape[0]
        if rank != 0:   # only the "decoder rank" has the real Gaussians; the others receive them
            sig, xy, col = torch.zeros_like(sig), torch.zeros_like(xy), torch.zeros_like(col)
        sig, xy, col = shard.broadcast_gaussians(sig, xy, col, src=0)
        a, b, c = (t.clone().requires_grad_(True) for t in (sig, xy, col))
        slab = shard.splat_band(a, b, c, H, W, dmax=0.4, grad_reduce=mode, backend=BACKENDS[backend])
        r0, r1 = shard.row_band(H, rank, world)
        assert slab.shape == (r1 - r0, W, 3)
        wgt = synthetic.grad_image(H, W, 6)
        (slab * wgt[r0:r1]).sum().backward()
        full = shard.gather_image(slab.detach(), H)
        out[rank] = dict(img=full.numpy(), gs=a.grad.numpy(), gc=b.grad.numpy(), gk=c.grad.numpy(),
                         sl=shard.gaussian_slice(n, rank, world))
    finally:
        dist.destroy_process_group()


def _reduce_scatter_supported():
    return True


@pytest.mark.parametrize("world,mode,backend", [(2, "all_reduce", "three-tensors"), (3, "all_reduce", "three-tensors"),
                                                (2, "reduce_scatter", "three-tensors"), (2, "all_reduce", "packed-out"),
                                                (3, "reduce_scatter", "packed-out")])
def test_row_band_shard_matches_single_process(world, mode, backend):
    from oracle import gs_oracle
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    try:
        mp.spawn(_worker, args=(world, port, mode, out, backend), nprocs=world, join=True)
    except Exception as e:  # gloo builds without reduce_scatter: the NCCL/RCCL path has it; skip here
        if mode == "reduce_scatter" and "reduce_scatter" in str(e).lower():
            pytest.skip(f"gloo lacks reduce_scatter_tensor in this build: {e}")
        raise
    sig, xy, col, H, W = synthetic.kernel_inputs(12, 10, 3.0, seed=5)
    wgt = synthetic.grad_image(H, W, 6)
    ref = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, 0.4)
    gref = gs_oracle.backward_f64(sig.numpy(), xy.numpy(), col.numpy(), wgt.numpy(), 0.4)
    for r in range(world):
        o = out[r]
        np.testing.assert_allclose(o["img"], ref, rtol=0, atol=1e-5)      # every rank gathers the same full image
        for got, want in zip((o["gs"], o["gc"], o["gk"]), gref):
            scale = np.abs(want).max()
            if mode == "all_reduce":
                assert np.abs(got - want).max() <= 1e-5 * scale
            else:   # rank r holds the reduced gradients of its Gaussian slice, zeros elsewhere
                a, b = o["sl"]
                assert np.abs(got[a:b] - want[a:b]).max() <= 1e-5 * scale
                assert np.abs(got[:a]).max(initial=0) == 0 and np.abs(got[b:]).max(initial=0) == 0


def _packed_worker(rank, world, port, mode, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sig, xy, col, H, W = synthetic.kernel_inputs(11, 10, 3.0, seed=5)   # 110 Gaussians: padded chunks at world 3
        n = sig.shape[0]
        # ONE [N,8] buffer per rank: the broadcast lands in it and the plan reads it where it is
        packed = shard.pack(sig, xy, col) if rank == 0 else torch.full((n, 8), float("nan"))
        shard.broadcast_packed(packed, src=0)
        p = packed.clone().requires_grad_(True)
        slab = shard.splat_band_packed(p, H, W, dmax=0.4, grad_reduce=mode, backend=OraclePackedBackend)
        r0, r1 = shard.row_band(H, rank, world)
        assert slab.shape == (r1 - r0, W, 3)
        wgt = synthetic.grad_image(H, W, 6)
        (slab * wgt[r0:r1]).sum().backward()
        out[rank] = dict(slab=slab.detach().numpy(), rows=(r0, r1), g=p.grad.numpy(), sl=shard.gaussian_slice(n, rank, world))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "all_reduce"), (3, "all_reduce"), (2, "reduce_scatter"), (3, "reduce_scatter"),
                                        (8, "reduce_scatter")])
def test_packed_row_band_shard_matches_single_process(world, mode):
    """the north star's pattern on one packed buffer: broadcast_packed + splat_band_packed, gradients
    reduced in place on the [N,8] buffer the backward wrote"""
    from oracle import gs_oracle
    mgr = mp.Manager()
    out = mgr.dict()
    try:
        mp.spawn(_packed_worker, args=(world, _free_port(), mode, out), nprocs=world, join=True)
    except Exception as e:  # gloo builds without reduce_scatter: the NCCL/RCCL path has it; skip here
        if mode == "reduce_scatter" and "reduce_scatter" in str(e).lower():
            pytest.skip(f"gloo lacks reduce_scatter_tensor in this build: {e}")
        raise
    sig, xy, col, H, W = synthetic.kernel_inputs(11, 10, 3.0, seed=5)   # 110 Gaussians: padded chunks at world 3
    wgt = synthetic.grad_image(H, W, 6)
    ref = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, 0.4)
    gref = shard.pack(*(torch.from_numpy(a) for a in
                        gs_oracle.backward_f64(sig.numpy(), xy.numpy(), col.numpy(), wgt.numpy(), 0.4))).numpy()
    for r in range(world):
        o = out[r]
        r0, r1 = o["rows"]
        np.testing.assert_allclose(o["slab"], ref[r0:r1], rtol=0, atol=1e-5)
        for cols in (slice(0, 3), slice(3, 5), slice(5, 8)):
            want, got, scale = gref[:, cols], o["g"][:, cols], np.abs(gref[:, cols]).max()
            if mode == "all_reduce":
                assert np.abs(got - want).max() <= 1e-5 * scale
            else:   # rank r holds the reduced gradients of its Gaussian slice, zeros elsewhere
                a, b = o["sl"]
                assert np.abs(got[a:b] - want[a:b]).max() <= 1e-5 * scale
                assert np.abs(got[:a]).max(initial=0) == 0 and np.abs(got[b:]).max(initial=0) == 0


def test_band_and_slice_partitions():
    for h, world in ((1024, 8), (1000, 3), (7, 8), (8192, 8)):
        bands = [shard.row_band(h, r, world) for r in range(world)]
        assert bands[0][0] == 0 and bands[-1][1] == h
        assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))
        assert max(b - a for a, b in bands) - min(b - a for a, b in bands) <= 1
    for n, world in ((65536, 8), (10, 3), (5, 8)):
        sl = [shard.gaussian_slice(n, r, world) for r in range(world)]
        assert sl[0][0] == 0 and sl[-1][1] == n and all(sl[i][1] == sl[i + 1][0] for i in range(world - 1))


def test_pack_unpack_roundtrip():
    s, x, c = torch.rand(5, 3), torch.rand(5, 2), torch.rand(5, 3)
    p = shard.pack(s, x, c)
    assert p.shape == (5, 8)
    for a, b in zip(shard.unpack(p), (s, x, c)):
        assert torch.equal(a, b)


def test_single_process_without_dist_is_full_image():
    sig, xy, col, H, W = synthetic.kernel_inputs(6, 6, 3.0, seed=9)
    slab = shard.splat_band(sig, xy, col, H, W, dmax=None, backend=OracleBackend)
    assert slab.shape == (H, W, 3)


# ---------------------------------------------------------------------------------------------------
# band-local exchange (BandExchange): every rank produces the Gaussians of its own band
# ---------------------------------------------------------------------------------------------------
class OraclePackedBackend(OracleBackend):
    """CPU stand-in for the packed half of HipBackend; NaN records are dead padding, as in the library."""

    @staticmethod
    def _live(records):
        return torch.isfinite(records).all(dim=1)

    @classmethod
    def forward_packed(cls, records, h, w, dmax, rows, cutoff=0.0):
        live = cls._live(records)
        s, c, k = shard.unpack(records[live])
        slab, _ = OracleBackend.forward(s, c, k, h, w, dmax, rows)
        return slab, (h, w, dmax, rows, live)

    @classmethod
    def forward_packed_into(cls, records, slab, h, w, dmax, rows, cutoff=0.0, flags=0, accumulate=False):
        part, state = cls.forward_packed(records, h, w, dmax, rows, cutoff)
        slab.copy_(slab + part if accumulate else part)
        return state

    @classmethod
    def backward_packed(cls, state, records, grad_slab, g_records):
        h, w, dmax, rows, live = state
        s, c, k = shard.unpack(records[live])
        g = OracleBackend.backward((h, w, dmax, rows), s, c, k, grad_slab)
        g_records.zero_()
        g_records[live] = shard.pack(*g)

    @staticmethod
    def select(own, h, w, dmax, rows, rows_above, rows_below, up, down, up_index, down_index, counts, cutoff=0.0):
        # conservative footprint: the dmax box in rows (+ half a pixel), as the exact-semantics oracle needs
        y = own[:, 4].double()
        hy = 0.5 * (h - 1)
        r0 = torch.ceil((y - dmax + 1.0) * hy - 0.5).clamp(min=0)
        r1 = torch.floor((y + dmax + 1.0) * hy + 0.5).clamp(max=h - 1)
        live = torch.isfinite(own).all(dim=1) & (r0 <= r1)
        go_up = live & (r0 < rows[0]) & (rows_above > 0)
        go_down = live & (r1 >= rows[1]) & (rows_below > 0)
        far = (go_up & (r0 < rows[0] - rows_above)) | (go_down & (r1 >= rows[1] + rows_below))
        cap = up.shape[0]
        for buf, idx, m in ((up, up_index, go_up), (down, down_index, go_down)):
            buf.fill_(float("nan"))
            sel = torch.nonzero(m).flatten()[:cap]
            buf[: sel.numel()] = own[sel]
            idx[: sel.numel()] = sel.int()
        counts[0], counts[1], counts[2], counts[3] = int(go_up.sum()), int(go_down.sum()), int(far.sum()), 0

    @staticmethod
    def merge(g_own, g_up, g_down, up_index, down_index, counts):
        cap = g_up.shape[0]
        for g, idx, n in ((g_up, up_index, int(counts[0])), (g_down, down_index, int(counts[1]))):
            n = min(n, cap)
            g_own.index_add_(0, idx[:n].long(), g[:n])


H_LR, W_LR, SCALE, DMAX_X = 12, 10, 3.0, 0.2


def _h_lr(world):
    """LR rows of the exchange tests: 12, or 32 for the eight-rank case (the node the driver scales to) -- bands of 12 HR rows,
    taller than the 9.6-px dmax box, so that footprints reach the ADJACENT band only (what BandExchange requires)"""
    return 32 if world >= 8 else H_LR


def _exchange_worker(rank, world, port, cap, out, transport="alltoall", overlap=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        h_lr = _h_lr(world)
        sig, xy, col, H, W = synthetic.kernel_inputs(h_lr, W_LR, SCALE, seed=5)
        lr0, lr1 = shard.row_band(h_lr, rank, world)           # this rank "decodes" its own LR rows
        mine = shard.pack(sig, xy, col)[lr0 * W_LR: lr1 * W_LR]
        ex = shard.BandExchange(mine.shape[0], cap, H, W, DMAX_X, backend=OraclePackedBackend, transport=transport, overlap=overlap)
        p = mine.clone().requires_grad_(True)
        slab = shard.splat_band_local(p, ex)
        err = None
        try:
            counts = ex.check()
        except RuntimeError as e:
            counts, err = None, str(e)
        r0, r1 = ex.rows
        wgt = synthetic.grad_image(H, W, 6)
        (slab * wgt[r0:r1]).sum().backward()
        out[rank] = dict(slab=slab.detach().numpy(), rows=(r0, r1), g=p.grad.numpy(), lr=(lr0, lr1), counts=counts, err=err)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,transport,overlap", [(2, "alltoall", False), (3, "alltoall", False), (4, "alltoall", False),
                                                     (8, "alltoall", False), (8, "p2p", False),
                                                     (2, "p2p", False), (3, "p2p", False),
                                                     (2, "alltoall", True), (3, "alltoall", True), (3, "p2p", True)])
def test_band_exchange_matches_single_process(world, transport, overlap):
    """both ways of issuing the two neighbour swaps of a step: ONE all_to_all_single whose split sizes are zero for every
    rank but g-1 / g+1 (default), and batched isend / irecv; overlap: the two-render form (own Gaussians rendered while the
    halos travel, the halo render added on top; halo gradients sent home under the own part's backward)"""
    from oracle import gs_oracle
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_exchange_worker, args=(world, _free_port(), 64, out, transport, overlap), nprocs=world, join=True)
    sig, xy, col, H, W = synthetic.kernel_inputs(_h_lr(world), W_LR, SCALE, seed=5)
    wgt = synthetic.grad_image(H, W, 6)
    ref = gs_oracle.forward_f64(sig.numpy(), xy.numpy(), col.numpy(), H, W, DMAX_X)
    gref = shard.pack(*(torch.from_numpy(a) for a in
                        gs_oracle.backward_f64(sig.numpy(), xy.numpy(), col.numpy(), wgt.numpy(), DMAX_X))).numpy()
    crossing = 0
    for r in range(world):
        o = out[r]
        assert o["err"] is None, o["err"]
        crossing += sum(o["counts"])
        r0, r1 = o["rows"]
        np.testing.assert_allclose(o["slab"], ref[r0:r1], rtol=0, atol=1e-5)
        a, b = o["lr"][0] * W_LR, o["lr"][1] * W_LR
        for cols in (slice(0, 3), slice(3, 5), slice(5, 8)):       # complete gradients of the rank's own Gaussians
            want = gref[a:b, cols]
            assert np.abs(o["g"][:, cols] - want).max() <= 1e-5 * np.abs(gref[:, cols]).max()
    assert crossing > 0     # the case does exercise the exchange


def test_band_exchange_reports_overflow():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_exchange_worker, args=(2, _free_port(), 2, out), nprocs=2, join=True)   # cap far too small
    assert any(out[r]["err"] and "enlarge cap" in out[r]["err"] for r in range(2))
    # ... and cannot go unnoticed even if nobody calls check(): the rank's image and gradient carry a NaN
    for r in range(2):
        if out[r]["err"]:
            assert np.isnan(out[r]["slab"]).any() and np.isnan(out[r]["g"]).any()


def test_band_exchange_is_not_silently_reused():
    """a second splat_band_local forward on the same BandExchange before the first backward would make
    that backward use the later step's records and indices -- it raises instead"""
    sig, xy, col, H, W = synthetic.kernel_inputs(H_LR, W_LR, SCALE, seed=5)
    mine = shard.pack(sig, xy, col)
    ex = shard.BandExchange(mine.shape[0], 8, H, W, DMAX_X, backend=OraclePackedBackend, rank=0, world=1)
    p1, p2 = mine.clone().requires_grad_(True), mine.clone().requires_grad_(True)
    a = shard.splat_band_local(p1, ex)
    b = shard.splat_band_local(p2, ex)
    b.sum().backward()                       # the latest forward owns the buffers: fine
    with pytest.raises(RuntimeError, match="one BandExchange per forward"):
        a.sum().backward()
    assert not torch.isnan(b).any() and not torch.isnan(p2.grad).any()


def test_splat_band_default_reduction_gives_the_source_rank_the_whole_gradient():
    import inspect
    assert inspect.signature(shard.splat_band).parameters["grad_reduce"].default == "all_reduce"


# ---------------------------------------------------------------------------------------------------
# tiled-inference driver with its tiles distributed over the ranks (gsasr_amd/split_and_joint_image.py)
# ---------------------------------------------------------------------------------------------------
def _tiled_worker(rank, world, port, path, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import tiled_models
    from gsasr_amd.split_and_joint_image import split_and_joint_image
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z = np.load(path)
        sc = float(z["scale"])
        res = split_and_joint_image(torch.from_numpy(z["lq"]), sc, int(z["split_size"]), int(z["overlap_size"]),
                                    tiled_models.model_g, tiled_models.model_fea2gs, torch.tensor([sc, sc]),
                                    crop_size=int(z["crop_size"]), cuda_rendering=False, distribute=True)
        out[rank] = res.numpy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 5])
def test_tiled_driver_distributes_tiles_over_ranks(world):
    """12 tiles over 2 and 5 ranks (uneven): every rank ends with the reference's stitched canvas"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiled_frac_s2p5_18x22.npz")
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_tiled_worker, args=(world, _free_port(), path, out), nprocs=world, join=True)
    want = np.load(path)["out"]
    for r in range(world):
        np.testing.assert_allclose(out[r], want, rtol=1e-5, atol=1e-6)
