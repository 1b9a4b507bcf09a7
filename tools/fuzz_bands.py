#!/usr/bin/env python
"""Randomised check of the two row-band data paths of gsasr_amd/shard.py, G bands driven from ONE process (GPU,
development aid):   python tools/fuzz_bands.py [cases] [seed]

  halo path        BandExchange.select -> (copies standing in for the P2P swap) -> local plan over [own | halos] ->
                   backward -> halo gradients returned -> merge          == the single full-image render and gradient
  replicated path  every band plans ALL Gaussians (packed records, tile-stationary backward) and the per-band partial
                   gradients are summed (what reduce-scatter does)       == the same
Random image shapes, 2..8 bands of unequal height, Gaussians assigned to ranks in raster bands or at random, dmax / none."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import _cabi, shard  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
skipped = 0
worst = {"halo img": 0.0, "halo grad": 0.0, "repl img": 0.0, "repl grad": 0.0}
for case in range(cases):
    world = int(rng.integers(2, 9))
    H = int(rng.integers(16 * world, 1600))
    W = int(rng.integers(8, 1600))
    n = int(rng.integers(world, 4000))
    g = torch.Generator().manual_seed(case)
    spx = 10 ** rng.uniform(-0.5, 1.2)                                     # typical sigma in pixels
    sig = torch.stack([spx * (0.3 + torch.rand(n, generator=g)) * 2 / (W - 1), spx * (0.3 + torch.rand(n, generator=g)) * 2 / (H - 1),
                       1.8 * torch.rand(n, generator=g) - 0.9], 1)
    xy = torch.rand(n, 2, generator=g) * 2.2 - 1.1
    col = torch.rand(n, 3, generator=g)
    dmax = [None, float(10 ** rng.uniform(-2.0, -0.3))][int(rng.integers(0, 2))]
    rec = shard.pack(sig, xy, col).to(dev)
    wgt = torch.randn(H, W, 3, generator=g).to(dev)
    what = (case, H, W, n, world, dmax, round(float(spx), 2))
    full, st = shard.HipBackend.forward_packed(rec, H, W, dmax, (0, H))
    gfull = torch.empty_like(rec)
    shard.HipBackend.backward_packed(st, rec, wgt, gfull)
    gmax = [float(gfull[:, c].abs().max()) for c in (slice(0, 3), slice(3, 5), slice(5, 8))]
    # ---- replicated path -----------------------------------------------------------------------------
    gsum = torch.zeros_like(rec)
    for r in range(world):
        rows = shard.row_band(H, r, world)
        slab, stb = shard.HipBackend.forward_packed(rec, H, W, dmax, rows, flags=_cabi.FLAG_BWD_TILE)
        e = float((slab - full[rows[0]:rows[1]]).abs().max()) / max(1.0, float(full.abs().max()))
        worst["repl img"] = max(worst["repl img"], e)
        assert e <= 1e-5, (what, "replicated image", r, e)
        gpart = torch.full_like(rec, float("nan"))
        shard.HipBackend.backward_packed(stb, rec, wgt[rows[0]:rows[1]].contiguous(), gpart)
        gsum += gpart
    for c, m in zip((slice(0, 3), slice(3, 5), slice(5, 8)), gmax):
        e = float((gsum[:, c] - gfull[:, c]).abs().max()) / max(1e-30, m)
        worst["repl grad"] = max(worst["repl grad"], e)
        assert e <= 2e-4, (what, "replicated gradient", e)
    # ---- halo path: Gaussians dealt to the ranks by the band their centre lies in, or at random --------
    by_band = rng.random() < 0.7
    if by_band:
        yrow = ((xy[:, 1] + 1) / 2 * (H - 1)).clamp(0, H - 1)
        owner = torch.zeros(n, dtype=torch.long)
        for r in range(world):
            r0, r1 = shard.row_band(H, r, world)
            owner[(yrow >= r0) & (yrow < r1)] = r
    else:
        owner = torch.from_numpy(rng.integers(0, world, n))
    idx = [torch.nonzero(owner == r).flatten().to(dev) for r in range(world)]
    cap = int(max(8, rng.integers(n // 4 + 8, n + 9)))
    exs = []
    for r in range(world):
        ex = shard.BandExchange(int(idx[r].numel()), cap, H, W, dmax, device=dev, rank=r, world=world)
        if idx[r].numel():
            ex.own.copy_(rec[idx[r]])
        ex.select()
        exs.append(ex)
    try:
        for ex in exs:
            ex.check()
    except RuntimeError:
        skipped += 1            # footprints beyond the adjacent band / capacity: the exchange says so (poisoned outputs are tested elsewhere)
        continue
    for r, ex in enumerate(exs):
        if r > 0:
            ex.from_above.copy_(exs[r - 1].send_down)
        if r < world - 1:
            ex.from_below.copy_(exs[r + 1].send_up)
    for ex in exs:
        slab, stb = shard.HipBackend.forward_packed(ex.records, H, W, dmax, ex.rows)
        e = float((slab - full[ex.rows[0]:ex.rows[1]]).abs().max()) / max(1.0, float(full.abs().max()))
        worst["halo img"] = max(worst["halo img"], e)
        assert e <= 1e-5, (what, "halo image", ex.rank, e, "by band" if by_band else "random owners")
        shard.HipBackend.backward_packed(stb, ex.records, wgt[ex.rows[0]:ex.rows[1]].contiguous(), ex.g_records)
    for r, ex in enumerate(exs):
        c = ex.cap
        if r > 0:
            ex.ret_up.copy_(exs[r - 1].g_records[exs[r - 1].n + c:])
        if r < world - 1:
            ex.ret_down.copy_(exs[r + 1].g_records[exs[r + 1].n: exs[r + 1].n + c])
    for r, ex in enumerate(exs):
        gm = ex.merge()
        if idx[r].numel() == 0:
            continue
        want = gfull[idx[r]]
        for cs, m in zip((slice(0, 3), slice(3, 5), slice(5, 8)), gmax):
            e = float((gm[:, cs] - want[:, cs]).abs().max()) / max(1e-30, m)
            worst["halo grad"] = max(worst["halo grad"], e)
            assert e <= 2e-4, (what, "halo gradient", r, e, "by band" if by_band else "random owners")
print(f"{cases} cases ok ({skipped} halo cases skipped: footprint beyond the adjacent band or capacity): worst {worst}")
