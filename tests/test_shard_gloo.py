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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out):
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
        slab = shard.splat_band(a, b, c, H, W, dmax=0.4, grad_reduce=mode, backend=OracleBackend)
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


@pytest.mark.parametrize("world,mode", [(2, "all_reduce"), (3, "all_reduce"), (2, "reduce_scatter")])
def test_row_band_shard_matches_single_process(world, mode):
    from oracle import gs_oracle
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    try:
        mp.spawn(_worker, args=(world, port, mode, out), nprocs=world, join=True)
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
