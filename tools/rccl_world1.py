#!/usr/bin/env python
"""First-contact insurance for the multi-GPU path on a ONE-GPU box: a world-size-1 process group over RCCL (backend "nccl")
on which every collective gsasr_amd/shard.py and gsasr_amd/split_and_joint_image.py issue is called once, in the form they
issue it, on device tensors -- broadcast, reduce_scatter_tensor (in place, on views of one buffer), all_reduce,
all_to_all_single with zero splits (what BandExchange._swap sends an edge rank's absent neighbour), the batched isend/irecv of
its "p2p" transport (to self: the only peer a one-rank group has), all_gather_into_tensor and all_gather.  Prints ONE JSON
line {name: "ok" | error}.  Results are checked where a one-rank group defines them (every collective is then the identity).

    python tools/rccl_world1.py            (run by tests/test_bench_dist.py under -m gpu)
"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from gsasr_amd import shard
    res = {"backend": dist.get_backend(), "ranks": dist.get_world_size()}

    def run(name, fn):
        try:
            fn()
            torch.cuda.synchronize(dev)
            res[name] = "ok"
        except Exception as e:      # noqa: BLE001 -- the report is the point
            res[name] = repr(e)[:300]

    n = 4096
    packed = torch.rand(n, 8, device=dev)
    ref = packed.clone()

    def bcast():
        shard.broadcast_packed(packed, src=0)
        assert torch.equal(packed, ref)
    run("broadcast (shard.broadcast_packed)", bcast)

    def rs():
        g = torch.rand(n, 8, device=dev)
        want = g.clone()
        out = shard.reduce_packed_grads_(g, n, "reduce_scatter")
        assert torch.equal(out, want[: out.shape[0]])
    run("reduce_scatter_tensor in place (shard.reduce_packed_grads_)", rs)

    def ar():
        g = torch.rand(n, 8, device=dev)
        want = g.clone()
        shard.reduce_packed_grads_(g, n, "all_reduce")
        assert torch.equal(g, want)
    run("all_reduce (shard.reduce_packed_grads_)", ar)

    def a2a():
        buf_in, buf_out = torch.rand(2048, 8, device=dev), torch.zeros(2048, 8, device=dev)
        # an edge rank of BandExchange._swap: its slice of the send buffer is empty for the neighbour it does not have
        dist.all_to_all_single(buf_out[:0], buf_in[:0], [0], [0])
        w = dist.all_to_all_single(buf_out[:0], buf_in[:0], [0], [0], async_op=True)
        w.wait()
        dist.all_to_all_single(buf_out[:1024], buf_in[:1024], [1024], [1024])      # (to self: a one-rank group's only peer)
        assert torch.equal(buf_out[:1024], buf_in[:1024])
    run("all_to_all_single, zero and self splits (BandExchange transport 'alltoall')", a2a)

    def p2p():
        a, b = torch.rand(1024, 8, device=dev), torch.zeros(1024, 8, device=dev)
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)])
        for r in reqs:
            r.wait()
        torch.cuda.synchronize(dev)
        assert torch.equal(a, b)
    run("batch_isend_irecv to self (BandExchange transport 'p2p')", p2p)

    def agt():
        stack = torch.rand(3, 3, 64, 64, device=dev)
        everyone = torch.empty(1 * 3, 3, 64, 64, device=dev)
        dist.all_gather_into_tensor(everyone, stack)
        assert torch.equal(everyone, stack)
    run("all_gather_into_tensor (split_and_joint_image distribute=True)", agt)

    def ag():
        slab = torch.rand(128, 96, 3, device=dev)
        img = shard.gather_image(slab, 128)
        assert torch.equal(img, slab)
    run("all_gather (shard.gather_image)", ag)

    def band():
        # a BandExchange on this group: select -> swap -> plan -> forward -> backward -> swap -> merge, both transports
        from gsasr_amd import synthetic
        sig, xy, col, H, W = synthetic.kernel_inputs(32, 32, 4.0, seed=1)
        rec = shard.pack(sig, xy, col).to(dev)
        for transport in ("alltoall", "p2p"):
            ex = shard.BandExchange(rec.shape[0], 1024, H, W, 0.1, device=dev, transport=transport)
            ex.own.copy_(rec)
            x = ex.own.clone().requires_grad_(True)
            img = shard.splat_band_local(x, ex)
            img.sum().backward()
            assert torch.isfinite(img).all() and torch.isfinite(x.grad).all()
    run("BandExchange + splat_band_local on the group (both transports)", band)

    dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)      # (RCCL's version banner sits in C stdio: out before the line, not behind it at exit)
    sys.stdout.write(json.dumps(res) + "\n")
    sys.stdout.flush()


if __name__ == "__main__":
    main()
