"""Helpers of bench.py (the repo-root bench keeps the driver contract; this file holds the workloads): the hot-path
`Step` with preallocated buffers for every BASELINE.json config, per-kernel timing on the launch stream, pair counts,
the single-GPU legs of the other scale factors, the exact-semantics / drop-in legs and the CPU baselines.  Bench
harness, not product."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling

CONFIGS = {
    # name: (h_lr, w_lr, scale, description)
    "c2": (256, 256, 4.0, "config2: 256x256 LR -> x4 (1024^2 HR), 65536 Gaussians, fwd+bwd"),
    "c3": (512, 512, 12.0, "config3: 512x512 LR -> x12 (6144^2 HR), 262144 Gaussians, fwd only"),
    "c4": (1024, 1024, 8.0, "config4: 1024x1024 LR -> x8 (8192^2 HR), 1048576 Gaussians, fwd+bwd, row-band shard"),
    # GSASR's real density at inference size (Fea2GS emits 16 Gaussians per LR pixel, utils/fea2gs.py:546-551): not a
    # BASELINE config, the workload VERDICT r2 item 6 asks for
    "c2x16": (256, 256, 4.0, "256x256 LR -> x4 (1024^2 HR) at 16 Gaussians per LR pixel: 1048576 Gaussians, fwd+bwd"),
    # rasterizer work of BASELINE config 5 (the training step): 16 samples of 48x48 LR crops x4, 16 Gaussians per LR
    # pixel (fea2gs), dmax 0.5, raw decoder parameters in, gradient w.r.t. them out -- ONE batched canvas
    "c5": (48, 48, 4.0, "config5 rasterizer: batch 16 x (48x48 LR -> x4, 36864 Gaussians), prologue+fwd+bwd, batched canvas"),
    # the same batch trained on `sample_coords` (sample_size = 2304 = 48^2 of the 192^2 pixels per sample): only those
    # pixels are evaluated; value = SAMPLED pixels per second
    "c5s": (48, 48, 4.0, "config5 rasterizer with sample_coords: batch 16 x (36864 Gaussians, 2304 of 192^2 pixels), prologue+fwd+bwd, sampled-pixel kernels"),
    # BASELINE config 5 as stated: the END-TO-END training step -- EDSR-baseline-shaped encoder + Fea2GS-shaped producer
    # (tools/c5_models.py) -> batched HIP splat -> per-sample crop -> L1 -> backward through the rasterizer -> Adam step
    "c5e2e": (48, 48, 4.0, "config5 end-to-end training step: EDSR-baseline-shaped encoder + Fea2GS-shaped producer -> HIP splat "
                           "(batch 16 x 48x48 LR crops x4, 16 Gaussians/LR px, dmax 0.5) -> L1 -> backward -> Adam"),
}


class Step:
    """One hot-path pass with preallocated buffers (nothing is allocated inside the timed region)."""

    def __init__(self, args, dev, rank, world):
        from gsasr_amd import _cabi, synthetic
        from gsasr_amd.shard import row_band
        self.cabi, self.dev, self.rank, self.world = _cabi, dev, rank, world
        h_lr, w_lr, scale, _ = CONFIGS[args.config]
        self.batched = args.config in ("c5", "c5s")
        self.sampled = args.config == "c5s"
        if self.batched:
            self.init_batched(args, dev, h_lr, w_lr, scale)
            return
        self.strong = args.config == "c4"
        if world > 1 and not self.strong:
            h_lr = h_lr * world      # weak scaling: stack `world` config-sized images vertically
        self.fwd_only = args.fwd_only or args.config == "c3"
        sig, xy, col, H, W = synthetic.kernel_inputs(h_lr, w_lr, scale, seed=0, gpp=16 if args.config == "c2x16" else 1, device="cpu")
        self.H, self.W, self.n = H, W, sig.shape[0]
        self.n_rank = self.n
        self.rows = row_band(H, rank, world)
        self.dmax = None if args.dmax < 0 else args.dmax
        self.cutoff = args.cutoff
        nrows = self.rows[1] - self.rows[0]
        self.grad_img = synthetic.grad_image(H, W, 1)[self.rows[0]:self.rows[1]].contiguous().to(dev)
        self.img = torch.zeros(nrows, W, 3, device=dev)
        self.dist = world > 1 or args.force_dist
        self.halo = self.dist and args.exchange == "halo"
        self.overlap = bool(getattr(args, "overlap", False))
        self.ex = None
        if self.halo:
            # sharded producer: this rank holds the Gaussians of its own LR rows (raster order => one slice)
            import torch.distributed as dist
            from gsasr_amd import shard
            lr0, lr1 = row_band(h_lr, rank, world)
            mine = shard.pack(sig, xy, col)[lr0 * w_lr: lr1 * w_lr].to(dev)
            n_local = mine.shape[0]
            probe = shard.BandExchange(n_local, max(n_local, 1), H, W, self.dmax, self.cutoff, device=dev)
            probe.own.copy_(mine)
            probe.select()
            need = torch.tensor([max(probe.check())], device=dev)       # raises if a footprint outreaches a band
            if world > 1:
                dist.all_reduce(need, op=dist.ReduceOp.MAX)
            self.halo_records = int(need.item())
            cap = max(1024, -(-int(self.halo_records * 1.25) // 1024) * 1024)
            del probe
            self.ex = shard.BandExchange(n_local, cap, H, W, self.dmax, self.cutoff, device=dev)
            self.ex.own.copy_(mine)
            self.n_rank = n_local
            # (x8 row bands: the tile-stationary backward, the library's own choice for whole images at this scale)
            self.plan = _cabi.plan_packed(self.ex.records, H, W, self.dmax, rows=self.rows, cutoff=self.cutoff,
                                          flags=_cabi.FLAG_BWD_TILE if (self.strong and not args.fwd_only) else 0)
            # one trial swap each way before anything is timed, per transport: ONE all_to_all_single per direction first,
            # batched P2P ops next; a transport that cannot do either shows up here and the collective pattern takes over.
            # Every rank takes the same data path: the weakest rank's result decides.
            ok = 0.0
            for level, transport in ((2.0, "alltoall"), (1.0, "p2p")):
                self.ex.transport = transport
                works = level
                try:
                    self.ex.exchange_forward()
                    self.ex.exchange_backward()
                    torch.cuda.synchronize(dev)
                except Exception as e:
                    works = 0.0
                    print(f"[bench] rank {rank}: halo exchange over {transport} unavailable ({e!r})", file=sys.stderr)
                if world > 1:
                    flag = torch.tensor([works], device=dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    works = float(flag.item())
                if works == level:
                    ok = level
                    break
            if ok == 0.0:
                print(f"[bench] rank {rank}: using broadcast + reduce_scatter", file=sys.stderr)
            if ok > 0.0:
                return
            self.halo, self.ex = False, None
            self.n_rank = self.n
        self.sig, self.xy, self.col = (t.to(dev) for t in (sig, xy, col))
        self.g = [torch.zeros_like(t) for t in (self.sig, self.xy, self.col)]
        # row bands that are handed every Gaussian of the image: the tile-stationary backward, which launches over
        # the band's tiles only (the Gaussian-stationary one spends a wave per off-band Gaussian)
        band_flags = _cabi.FLAG_BWD_TILE if (world > 1 and not self.fwd_only) else 0
        if self.fwd_only:
            band_flags |= _cabi.FLAG_FORWARD_ONLY        # (inference: no backward records in the plan)
        self.plan = _cabi.plan(self.sig, self.xy, self.col, H, W, self.dmax, rows=self.rows, cutoff=self.cutoff, flags=band_flags)
        if self.dist:
            # the north star's pattern (BASELINE config 4): ONE [N,8] buffer the broadcast lands in and the plan reads where
            # it is (GSASR_FLAG_STRIDE8), ONE [N,8] gradient buffer the backward writes and the reduce-scatter runs on in place
            from gsasr_amd import shard
            self.shard = shard
            self.packed = shard.pack(self.sig, self.xy, self.col)
            per = (self.n + world - 1) // world
            self.gpad = torch.zeros(per * world, 8, device=dev)
            self.plan = _cabi.plan_packed(self.packed, H, W, self.dmax, rows=self.rows, cutoff=self.cutoff, flags=band_flags)

    def init_batched(self, args, dev, h_lr, w_lr, scale):
        """config 5: every rank runs its own batch of 16 samples (data parallel over samples: no exchange)"""
        import ctypes
        from gsasr_amd import _cabi, synthetic
        B, gpp = 16, 16
        self.strong, self.fwd_only, self.dist, self.halo, self.ex = False, args.fwd_only, False, False, None
        self.dmax = 0.5 if args.dmax == 0.1 else (None if args.dmax < 0 else args.dmax)   # training box unless overridden
        self.cutoff = args.cutoff
        H = W = int(h_lr * scale)
        self.B, self.H, self.W = B, B * H * self.world, W     # H x W reported = all samples' pixels on all ranks
        self.pix_rank = B * H * W
        p = torch.stack([synthetic.gs_parameters(h_lr, w_lr, seed=b + 100 * self.rank, gpp=gpp) for b in range(B)])
        self.p = p.to(dev)
        self.n = self.n_rank = B * p.shape[1]
        self.steps = torch.full((B,), 1.2 / scale, device=dev)
        self.bdims = _cabi.make_batch_dims(p.shape[1], [(H, W)] * B, W, H, self.dmax, cutoff=self.cutoff,
                                           flags=_cabi.FLAG_OVERWRITE_IMAGE | _cabi.FLAG_CHW_IMAGE)
        L = _cabi.lib()
        nbytes = L.gsasr_step_workspace_bytes(ctypes.byref(self.bdims))
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.img = torch.empty(B, 3, self.bdims.slot, W, device=dev)
        self.grad_img = torch.rand(B, self.bdims.slot, W, 3, device=dev)
        self.gp = torch.empty_like(self.p)
        self.rows = (0, self.bdims.h)
        self.plan = _cabi.Plan(self.bdims, self.ws, dev)
        if self.sampled:
            S = h_lr * w_lr                    # sample_size: as many points as one LR crop has pixels
            g = torch.Generator().manual_seed(1234 + self.rank)
            self.pts = torch.stack([torch.randint(0, H, (B, S), generator=g), torch.randint(0, W, (B, S), generator=g)],
                                   dim=2).to(device=dev, dtype=torch.int32).contiguous()
            self.S = S
            self.out = torch.empty(B, 3, S, device=dev)
            self.grad_out = torch.rand(B, 3, S, device=dev)
            self.sws = torch.empty(L.gsasr_sample_workspace_bytes(ctypes.byref(self.bdims), S), dtype=torch.uint8, device=dev)
            self.H, self.W = B * S * self.world, 1     # "pixels" reported = the sampled ones
            self.pix_rank = B * S

    def batched_forward(self):
        import ctypes
        c = self.cabi
        if self.sampled:
            c.check(c.lib().gsasr_step_sample_forward(self.p.data_ptr(), self.steps.data_ptr(), ctypes.byref(self.bdims),
                                                      self.ws.data_ptr(), self.ws.numel(), self.pts.data_ptr(), self.S,
                                                      self.out.data_ptr(), self.sws.data_ptr(), self.sws.numel(),
                                                      c._stream(self.dev)), "gsasr_step_sample_forward")
            return
        keep = self.bdims.flags & ~(c.FLAG_COUNTERS_CLEAN | c.FLAG_PARITY)      # (persistent workspace: see do_plan)
        self.nplans = getattr(self, "nplans", 0) + 1
        if torch.cuda.is_current_stream_capturing():
            self.nplans = 0
        self.bdims.flags = keep if self.nplans <= 1 else keep | c.FLAG_COUNTERS_CLEAN | (c.FLAG_PARITY if self.nplans % 2 == 0 else 0)
        c.check(c.lib().gsasr_step_forward(self.p.data_ptr(), self.steps.data_ptr(), ctypes.byref(self.bdims),
                                           self.ws.data_ptr(), self.ws.numel(), self.img.data_ptr(), c._stream(self.dev)),
                "gsasr_step_forward")

    def batched_backward(self):
        import ctypes
        c = self.cabi
        if self.sampled:    # (points = NULL: the sorted points of the forward call are still in the scratch)
            c.check(c.lib().gsasr_step_sample_backward(self.p.data_ptr(), self.steps.data_ptr(), self.grad_out.data_ptr(),
                                                       self.gp.data_ptr(), ctypes.byref(self.bdims), self.ws.data_ptr(),
                                                       self.ws.numel(), None, self.S, self.sws.data_ptr(), self.sws.numel(),
                                                       c._stream(self.dev)), "gsasr_step_sample_backward")
            return
        c.check(c.lib().gsasr_step_backward(self.p.data_ptr(), self.steps.data_ptr(), self.grad_img.data_ptr(),
                                            self.gp.data_ptr(), ctypes.byref(self.bdims), self.ws.data_ptr(),
                                            self.ws.numel(), c._stream(self.dev)), "gsasr_step_backward")

    # --- the three stages, callable separately for per-kernel timing -------------------------------------
    def do_plan(self):
        import ctypes
        p = self.plan
        p.pool_key = None       # this workspace is re-planned by hand from here on: it must never go back to the package's pool
        if not getattr(self, "own_dims", False):     # (the package shares one dims struct between plans of a shape: take a copy to edit)
            p.dims = type(p.dims).from_buffer_copy(p.dims)
            self.own_dims = True
        # the workspace persists across steps: every plan zeroes the other parity's cell counters on the side, so after
        # the first one no memset launch is needed (GSASR_FLAG_COUNTERS_CLEAN / GSASR_FLAG_PARITY alternate)
        c = self.cabi
        keep = p.dims.flags & ~(c.FLAG_COUNTERS_CLEAN | c.FLAG_PARITY)
        self.nplans = getattr(self, "nplans", 0) + 1
        if torch.cuda.is_current_stream_capturing():
            self.nplans = 0         # a captured plan is replayed with the same parity: it must zero its own counters
        p.dims.flags = keep if self.nplans <= 1 else keep | c.FLAG_COUNTERS_CLEAN | (c.FLAG_PARITY if self.nplans % 2 == 0 else 0)
        if self.halo:
            self.cabi.check(c.lib().gsasr_splat_plan(self.ex.records.data_ptr(), self.ex.records.data_ptr() + 12,
                                                     self.ex.records.data_ptr() + 20, ctypes.byref(p.dims), p.workspace.data_ptr(),
                                                     p.workspace.numel(), c._stream(self.dev)), "plan")
            return
        if self.dist:       # columns of the packed records
            base = self.packed.data_ptr()
            self.cabi.check(c.lib().gsasr_splat_plan(base, base + 12, base + 20, ctypes.byref(p.dims), p.workspace.data_ptr(),
                                                     p.workspace.numel(), c._stream(self.dev)), "plan")
            return
        self.cabi.check(self.cabi.lib().gsasr_splat_plan(self.sig.data_ptr(), self.xy.data_ptr(), self.col.data_ptr(),
                                                         ctypes.byref(p.dims), p.workspace.data_ptr(),
                                                         p.workspace.numel(), self.cabi._stream(self.dev)), "plan")

    def do_forward(self):
        self.cabi.forward(self.plan, self.img, overwrite=True)      # what rendering_cuda_dmax enqueues

    def do_backward(self):
        if self.halo:
            self.cabi.backward_packed(self.plan, self.ex.records, self.grad_img, self.ex.g_records, overwrite=True)
            return
        if self.dist:
            self.cabi.backward_packed(self.plan, self.packed, self.grad_img, self.gpad[: self.n], overwrite=True)
            return
        self.cabi.backward(self.plan, self.sig, self.xy, self.col, self.grad_img, *self.g, overwrite=True)

    def __call__(self):
        if self.batched:
            self.batched_forward()               # prologue + plan + forward of all 16 samples
            if not self.fwd_only:
                self.batched_backward()          # splat backward + prologue backward
            return
        if self.halo and getattr(self, "overlap", False):
            # the two-render form (shard.BandExchange(overlap=True)): own Gaussians planned and splatted while the halos
            # travel, the halo records added on top; backward halo part first, its gradients fly home under the own part
            ex, c = self.ex, self.cabi
            n, H, W = ex.n, self.H, self.W
            tile = c.FLAG_BWD_TILE if (self.strong and not self.fwd_only) else 0
            ex.select()
            fly = ex._swap("fwd", ex.send_up, ex.from_above, ex.send_down, ex.from_below, async_op=True)
            po = c.plan_packed(ex.records[:n], H, W, self.dmax, rows=self.rows, cutoff=ex.cutoff, flags=tile | ex.plan_flags)
            c.forward(po, self.img, overwrite=True)
            ex._wait(fly)
            ph = c.plan_packed(ex.records[n:], H, W, self.dmax, rows=self.rows, cutoff=ex.cutoff, flags=tile)
            c.forward(ph, self.img, overwrite=False)
            if not self.fwd_only:
                c.backward_packed(ph, ex.records[n:], self.grad_img, ex.g_records[n:], overwrite=True)
                fly = ex._swap("bwd", ex.g_records[n:n + ex.cap], ex.ret_up, ex.g_records[n + ex.cap:], ex.ret_down, async_op=True)
                c.backward_packed(po, ex.records[:n], self.grad_img, ex.g_records[:n], overwrite=True)
                ex._wait(fly)
                ex.merge()
            return
        if self.halo:
            self.ex.exchange_forward()          # Gaussians crossing a band edge -> neighbours (P2P)
            self.do_plan()
            self.do_forward()
            if not self.fwd_only:
                self.do_backward()
                self.ex.exchange_backward()     # their partial gradients come back and are merged
            return
        if self.dist:
            self.shard.broadcast_packed(self.packed, src=0)          # Gaussians from the decoder rank, binned where they land
        self.do_plan()
        self.do_forward()
        if not self.fwd_only:
            self.do_backward()
            if self.dist:                                            # per-Gaussian grads, summed over bands, in place
                self.shard.reduce_packed_grads_(self.gpad, self.n, "reduce_scatter")

    # algorithmic bytes: fwd = 32 N + 24 H W with the reference's accumulate-into contract; the
    # step stores into a fresh image (GSASR_FLAG_OVERWRITE_IMAGE), so the image is written once and never read:
    # fwd = 32 N + 12 H W.  bwd = 64 N + 12 H W.  (per rank: own rows)
    def bytes_fwd(self):
        return 32 * self.n_rank + 12 * (self.rows[1] - self.rows[0]) * self.W

    def bytes_bwd(self):
        return 64 * self.n_rank + 12 * (self.rows[1] - self.rows[0]) * self.W


def time_stage(fn, iters, dev):
    """average device time of one call of `fn`, with events recorded on the stream the kernels run on"""
    st = torch.cuda.current_stream(dev)
    fn()
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(st)
        fn()
        b.record(st)
    torch.cuda.synchronize(dev)
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return sum(ts) / len(ts), ts[len(ts) // 2]


def effective_tau(step, cutoff):
    """(tau, K) the step's CURRENT plan built its windows with: read back from the plan header (data-derived under the
    adaptive default of the bounded op: ln(K / 1e-5), K = the plan's bound on the dmax boxes covering one pixel)"""
    tau, k = step.cabi.plan_cutoff(step.plan)
    if not tau > 0:
        return step.cabi.resolve_cutoff(cutoff, step.plan.dims.s), 0
    return tau, k


def window_pairs(sig, xy, H, W, dmax, tau, rows):
    """(Gaussian, pixel) pairs: inside the reference's dmax box (what gs_cuda_dmax sums) and
    inside the window the kernels sweep (box ∩ marginal support |d| <= sigma*sqrt(2 tau)); exact integer
    counts from the same double-precision window arithmetic as the plan (gaussian_box in gsasr_splat.hip)."""
    s64, x64 = sig.double().cpu(), xy.double().cpu()
    hx, hy = 0.5 * (W - 1), 0.5 * (H - 1)
    cx, cy = (x64[:, 0] + 1.0) * hx, (x64[:, 1] + 1.0) * hy
    out = []
    for cut in (False, True):
        ex = torch.full_like(cx, float("inf") if dmax is None else dmax)
        ey = ex.clone()
        if cut and tau > 0:
            k = (2.0 * tau) ** 0.5
            ex, ey = torch.minimum(ex, k * s64[:, 0]), torch.minimum(ey, k * s64[:, 1])
        c0 = torch.ceil(cx - ex * hx).clamp(min=0)
        c1 = torch.floor(cx + ex * hx).clamp(max=W - 1)
        r0 = torch.ceil(cy - ey * hy).clamp(min=rows[0])
        r1 = torch.floor(cy + ey * hy).clamp(max=rows[1] - 1)
        ok = torch.isfinite(cx) & torch.isfinite(cy)
        out.append(int(((c1 - c0 + 1).clamp(min=0) * (r1 - r0 + 1).clamp(min=0))[ok].sum().item()))
    return out


# kernels of a stage, by substring of the rocprofv3 kernel name (the tile-stationary backward counts its gather)
STAGE_KERNELS = {"forward": ["k_render_fwd"], "backward": ["k_render_bwd", "k_bwd_gather", "k_prologue_bwd_gather"]}
SQ_COUNTERS = ("SQ_INSTS_VALU", "SQ_BUSY_CYCLES")


def live_counters(argv_extra, groups=None, sq=True, timeout=240):
    """Counters of the kernels of each stage, COUNTED IN THIS RUN: short child runs of bench.py (the same workload, five steps,
    nothing else) under `rocprofv3 --pmc FETCH_SIZE`, `--pmc WRITE_SIZE` and (sq) `--pmc SQ_INSTS_VALU SQ_BUSY_CYCLES` --
    separate passes, counters + kernel trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Per stage:
      hbm_bytes  = (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch (the guide's gfx950 correction: FETCH_SIZE counts half of a wide
                   read stream; WRITE_SIZE calibrated on the forward's image store), summed over the stage's kernels;
      valu_busy  = SQ_INSTS_VALU x 32 SE x 4 cycles / (1024 SIMDs x SQ_BUSY_CYCLES): the VALU issue occupancy (a lower bound:
                   v_exp_f32 holds the port for 8 cycles) -- the counters are per-SE averages.
    Returns ({stage: {...}}, note) or (None, why-not)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    groups = groups or STAGE_KERNELS
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None, "rocprofv3 not found"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    passes = [("FETCH_SIZE",), ("WRITE_SIZE",)] + ([SQ_COUNTERS] if sq else [])
    for ctrs in passes:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            env = dict(os.environ, TMPDIR="/tmp", GSASR_BENCH_CHILD="1")
            cmd = [prof, "--pmc", *ctrs, "--kernel-trace", "-d", tmp, "-o", "p", "--", sys.executable, os.path.join(root, "bench.py"),
                   "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-graph", "--no-extras", "--no-live-pmc", *argv_extra]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            except (subprocess.TimeoutExpired, OSError) as e:
                return None, f"{ctrs[0]} pass: {e!r}"
            db = None
            for dirpath, _, files in os.walk(tmp):
                for f in files:
                    if f.endswith("_results.db"):
                        db = os.path.join(dirpath, f)
            if r.returncode != 0 or db is None:
                return None, f"{ctrs[0]} pass failed (rc {r.returncode}): {r.stderr[-300:]}"
            con = sqlite3.connect(db)
            pcols = [c[1] for c in con.execute("pragma table_info(pmc_events)")]
            kcol = "name" if "name" in pcols else "kernel_name"
            ccol = "counter_name" if "counter_name" in pcols else "pmc_name"
            vcol = "value" if "value" in pcols else "counter_value"
            for ctr in ctrs:
                for name, val in con.execute(f"select {kcol}, avg({vcol}) from pmc_events where {ccol} = ? group by {kcol}", (ctr,)):
                    for stage, subs in groups.items():
                        if name and any(sub in name for sub in subs):
                            got.setdefault(stage, {}).setdefault(ctr, 0.0)
                            got[stage][ctr] += float(val)
                            got[stage].setdefault("kernels", set()).add(name.replace("(anonymous namespace)::", "").split("(")[0][:60])
            con.close()
    out = {}
    for stage, v in got.items():
        e = {"hbm_bytes": int((2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024),
             "fetch_KiB": round(v.get("FETCH_SIZE", 0.0), 1), "write_KiB": round(v.get("WRITE_SIZE", 0.0), 1),
             "kernels": sorted(v.get("kernels", ()))}
        if v.get("SQ_BUSY_CYCLES"):
            e["valu_busy"] = round(v.get("SQ_INSTS_VALU", 0.0) * 32.0 * 4.0 / (1024.0 * v["SQ_BUSY_CYCLES"]), 4)
            e["valu_insts"] = int(v.get("SQ_INSTS_VALU", 0.0) * 32)
        out[stage] = e
    if not out:
        return None, "no matching kernels in the counter passes"
    return out, ("LIVE: counted in this run by child runs of this command under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc "
                 "SQ_INSTS_VALU SQ_BUSY_CYCLES (separate passes, counters + kernel trace only); bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB "
                 "per launch -- the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md; valu_busy = SQ_INSTS_VALU x 32 x 4 / (1024 x SQ_BUSY_CYCLES)")


def bound_label(valu_busy):
    """which resource the counters say binds a render kernel: the VALU issue port from 0.7 of its cycles up (v_exp_f32 holds it
    twice as long as the formula counts), else the wave's chain of dependent round trips; never HBM on this path (frac says how far)"""
    if valu_busy is None:
        return "valu-issue/latency (no SQ counters in this run)"
    return "valu-issue" if valu_busy >= 0.7 else "latency (dependent round trips at low occupancy)"


def live_hbm_traffic(argv_extra, kernel_substrings, timeout=150):
    """(round 4 interface) HBM bytes per launch of the kernels whose names contain one of `kernel_substrings`"""
    res, note = live_counters(argv_extra, {k: [k] for k in kernel_substrings}, sq=False, timeout=timeout)
    if res is None:
        return None, note
    return {k: v["hbm_bytes"] for k, v in res.items()}, note


def copy_bandwidth(dev):
    """measured HBM stream rate on this box: device-to-device copy of 1 GiB (read + write = 2 GiB of traffic)"""
    n = 1 << 28
    a, b = torch.empty(n, device=dev), torch.empty(n, device=dev)
    a.fill_(1.0)
    avg, _ = time_stage(lambda: b.copy_(a), 5, dev)
    return 2.0 * 4.0 * n / (avg * 1e-3) / 1e9


# VALU ceilings of the pair evaluation (instruction costs measured by tools/valu_rate.hip on this
# chip: a wave64 VALU instruction -- packed fp32 included -- issues in 4 cycles per SIMD, v_exp_f32 in 8).  One packed
# trip evaluates 128 (Gaussian, pixel) pairs:
#   forward  : 12 VALU + 2 v_exp_f32 = 64 cycles   (fwd_eval_one in gsasr_splat.hip)
#   forward, wide kernel (16 x 16 sub-tiles, x5 and up): a record's column part (4 VALU) serves TWO packed trips of
#              6 VALU + 2 v_exp_f32 each: 16 VALU + 4 v_exp_f32 = 96 cycles per 256 pairs = 48 per 128 (fwd_eval_lds16)
#   backward : 14 VALU + 2 v_exp_f32 = 72 cycles   (bwd_trip: residual, exponent, <grad, colour>, 3 moments, 3 colour sums)
SIMDS, CLOCK_HZ = 1024, 2.4e9
PAIR_CEILING = {"forward": {"valu": 12, "exp": 2, "cycles_per_128_pairs": 64},
                "forward_wide": {"valu": 8, "exp": 2, "cycles_per_128_pairs": 48},
                "backward": {"valu": 14, "exp": 2, "cycles_per_128_pairs": 72}}
for _k in PAIR_CEILING.values():
    _k["pairs_per_s"] = SIMDS * CLOCK_HZ / _k["cycles_per_128_pairs"] * 128.0


def pair_ceiling(step, stage):
    """the ceiling entry of `stage` for the kernel this step's plan actually runs (the forward has two)"""
    if stage == "forward" and not getattr(step, "batched", False) and step.cabi.forward_subtile_width(step.plan) == 16:
        return PAIR_CEILING["forward_wide"]
    return PAIR_CEILING[stage]


def flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def emit(out):
    """the ONE JSON line, as the LAST thing on stdout: libraries in the process (RCCL prints a version banner through C
    stdio when NCCL_DEBUG=VERSION, as on this image) have their buffered output flushed first, so that it cannot land
    behind the line at exit (the other ranks flush theirs before the final barrier, see main)"""
    flush_c_stdio()
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()


def wall_ms(fn, n, dev, warm=3, warm_s=0.0):
    """mean wall time of `fn` over n calls; `warm` untimed calls first, and -- warm_s > 0 -- as many more as it takes to keep
    the GPU busy for that long (a leg that follows host-bound work finds the clocks down: 25 steps of 0.3 ms did not bring
    them back, and the x12 leg of one run read 0.69 ms per step for kernels that take 0.32)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    t_end = time.perf_counter() + warm_s
    while time.perf_counter() < t_end:
        for _ in range(5):
            fn()
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / n * 1e3


def exact_runs(args, dev, pixels):
    """the SAME config-2 step with the support cull weakened to the reference's exact set of non-zero fp32 terms
    (tau = 104) and switched off (tau < 0: every in-box pair, the reference's operation count; SURVEY.md 7 hard part 1)"""
    import copy
    out = {}
    for name, tau in (("tau104_reference_nonzero_terms", 104.0), ("nocull_every_in_box_pair", -1.0)):
        a = copy.copy(args)
        a.cutoff = tau
        st = Step(a, dev, 0, 1)
        ms = min(wall_ms(st, 10, dev) for _ in range(2))      # (two takes: one collection of round 5 caught an 80 ms stall in the only one)
        tau_eff, _ = effective_tau(st, tau)
        in_box, swept = window_pairs(st.sig, st.xy, st.H, st.W, st.dmax, tau_eff, st.rows)
        out[name] = {"cutoff_tau": tau, "ms_per_step": ms, "value": pixels / (ms * 1e-3) / 1e6, "unit": "HR Mpixels/s",
                     "pairs_swept_per_direction": swept, "pairs_in_dmax_box": in_box}
        del st
    return out


def dropin_run(args, dev):
    """what a user who only aliases the reference's imports gets: `GSCUDA.apply(sigmas, coords,
    colors, torch.zeros(H,W,3), dmax)` + autograd, allocations, memset, read-modify-write image and all, wall clock"""
    from gsasr_amd import synthetic
    from gsasr_amd.gs_cuda.gswrapper import GSCUDA as G0
    from gsasr_amd.gs_cuda_dmax.gswrapper import GSCUDA as G1
    h_lr, w_lr, scale, _ = CONFIGS["c2"]
    sig, xy, col, H, W = synthetic.kernel_inputs(h_lr, w_lr, scale, seed=0, device="cpu")
    a, b, c = (t.to(dev).requires_grad_(True) for t in (sig, xy, col))
    wgt = synthetic.grad_image(H, W, 1).to(dev)
    dmax = None if args.dmax < 0 else args.dmax

    def step():
        a.grad = b.grad = c.grad = None
        z = torch.zeros(H, W, 3, device=dev)
        img = G0.apply(a, b, c, z) if dmax is None else G1.apply(a, b, c, z, dmax)
        img.backward(wgt)

    ms = wall_ms(step, 100, dev, warm=20)
    # What PyTorch's own engine costs under this calling convention on this host, whatever the node does: a Function whose
    # forward and backward launch nothing.  `.backward()` hands the node to the device's autograd worker thread; with
    # torch.autograd.set_multithreading_enabled(False) -- a caller's choice, one line around the training loop -- the engine runs
    # it on the calling thread.  Both forms are timed for the null node and for the drop-in.
    class _Null(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a_, b_, c_, img_):
            ctx.save_for_backward(a_, b_, c_)
            return img_

        @staticmethod
        @torch.autograd.function.once_differentiable
        def backward(ctx, g_):
            a_, b_, c_ = ctx.saved_tensors
            return torch.empty_like(a_), torch.empty_like(b_), torch.empty_like(c_), None

    z0 = torch.zeros(H, W, 3, device=dev)

    def null_step():
        a.grad = b.grad = c.grad = None
        _Null.apply(a, b, c, z0).backward(wgt)

    floor = wall_ms(null_step, 200, dev, warm=20)
    ms_st = floor_st = None
    try:
        with torch.autograd.set_multithreading_enabled(False):
            floor_st = wall_ms(null_step, 200, dev, warm=20)
            ms_st = wall_ms(step, 100, dev, warm=20)
    except Exception as e:      # (an older torch without the switch)
        print(f"[bench] single-threaded autograd engine unavailable: {e!r}", file=sys.stderr)
    # ... and the pybind-shaped module underneath it (`gscuda.gs_render` / `gs_render_backward`, reference
    # gswrapper.cpp:9-73 -> the C launchers gsasr_gs_render_dmax / _backward_dmax: scratch allocated stream-ordered and the
    # Gaussians binned in BOTH calls, as a maintainer who rebinds only the two launchers would get it)
    from gsasr_amd import gscuda
    ad, bd, cd = a.detach(), b.detach(), c.detach()
    g = [torch.zeros_like(t) for t in (ad, bd, cd)]

    def launchers():
        z = torch.zeros(H, W, 3, device=dev)
        for t in g:
            t.zero_()
        if dmax is None:
            gscuda.gs_render(ad, bd, cd, z, ad.shape[0], H, W, 3)
            gscuda.gs_render_backward(ad, bd, cd, wgt, *g, ad.shape[0], H, W, 3)
        else:
            gscuda.gs_render(ad, bd, cd, z, ad.shape[0], H, W, 3, dmax)
            gscuda.gs_render_backward(ad, bd, cd, wgt, *g, ad.shape[0], H, W, 3, dmax)

    ms_l = wall_ms(launchers, 30, dev, warm=5)
    return {"ms_per_step": ms, "value": H * W / (ms * 1e-3) / 1e6, "unit": "HR Mpixels/s",
            "engine_floor_ms": floor, "engine_floor_note": "a null autograd Function (no launches) through the same apply + .backward(grad): PyTorch's own host cost per step on this box",
            "ms_per_step_engine_on_calling_thread": ms_st, "engine_floor_on_calling_thread_ms": floor_st,
            "engine_on_calling_thread_note": "the same loops inside torch.autograd.set_multithreading_enabled(False): no hand-off to the device's autograd worker thread",
            "launchers_ms_per_step": ms_l, "launchers_value": H * W / (ms_l * 1e-3) / 1e6,
            "what": "GSCUDA.apply(sigmas, coords, colors, torch.zeros(H,W,3)[, dmax]) + .backward(grad) through torch autograd, "
                    "wall clock incl. host launch overhead, allocations and the accumulate-into (+=) image"}


def published_inputs():
    """the inputs of the reference's ONE committed measurement (utils/gs_cuda/profile.py:104-113): 512 x 512 x 3 image,
    s = 262 144 Gaussians, sigmas = 0.2 * rand(s, 3) with the two sigma columns x 5 (so sigma in [0, 1) of the [-1, 1] grid:
    image-spanning Gaussians, rho in [0, 0.2)), coords uniform on the grid, colours rand -- seeded here"""
    g = torch.Generator().manual_seed(0)
    s, H, W = 512 * 512, 512, 512
    sigmas = 0.2 * torch.rand(s, 3, generator=g)
    sigmas[:, :2] = 5 * sigmas[:, :2]
    coords = 2 * torch.rand(s, 2, generator=g) - 1.0
    colors = torch.rand(s, 3, generator=g)
    return sigmas, coords, colors, H, W


def published_leg(args, dev, calls=10):
    """The reference's only published number of this path, run here: `gaussiansplatting_render` of the UNBOUNDED op
    (utils/gs_cuda/gswrapper.py:41-48) under no_grad, mean wall time of 10 synchronised calls -- 1 184.5 ms per call in
    utils/gs_cuda/profile.log:44 (hardware unstated there; the binaries are sm_80 and the README names A100).  Timed at the
    library's default support cutoff AND with the cull off (every one of the reference's s * H * W = 6.9e10 pairs), through
    the same drop-in function, plus the plan / forward kernels alone through the C ABI."""
    from gsasr_amd import _cabi
    from gsasr_amd.gs_cuda.gswrapper import gaussiansplatting_render
    sigmas, coords, colors, H, W = published_inputs()
    a, b, c = sigmas.to(dev), coords.to(dev), colors.to(dev)
    ref_ms = 1184.4923
    out = {"workload": "utils/gs_cuda/profile.py:104-113: gs_cuda (unbounded) forward, 512x512x3 image, 262144 Gaussians with sigma in [0,1) "
                       "of the grid, no_grad, mean of 10 calls of gaussiansplatting_render incl. its torch.zeros",
           "published_ms_per_call": ref_ms, "published_source": "utils/gs_cuda/profile.log:44 (hardware unstated; sm_80 binaries, README names A100)",
           "pairs_reference": sigmas.shape[0] * H * W}
    keep = _cabi.get_default_cutoff()
    try:
        for name, tau in (("default_cutoff", 0.0), ("nocull_every_pair", -1.0)):
            _cabi.set_default_cutoff(tau)

            def call():
                with torch.no_grad():
                    return gaussiansplatting_render(a, b, c, (H, W, 3))

            ms = wall_ms(call, calls, dev, warm=2)
            plan = _cabi.plan(a, b, c, H, W, None, cutoff=tau, flags=_cabi.FLAG_FORWARD_ONLY)
            img = torch.empty(H, W, 3, device=dev)
            fwd_avg, _ = time_stage(lambda: _cabi.forward(plan, img, overwrite=True), 5, dev)
            tau_eff, k_box = _cabi.plan_cutoff(plan)
            if not tau_eff > 0:
                tau_eff = _cabi.resolve_cutoff(tau, sigmas.shape[0])
            _, swept = window_pairs(sigmas, coords, H, W, None, tau_eff if tau >= 0 else 0.0, (0, H))
            out[name] = {"cutoff_tau": round(float(tau_eff), 3) if tau >= 0 else -1, "ms_per_call": ms, "vs_published": ref_ms / ms,
                         "value": H * W / (ms * 1e-3) / 1e6, "unit": "HR Mpixels/s (fwd only)",
                         "forward_kernel_ms": fwd_avg, "pairs_swept": swept, "Gpairs_per_s": swept / (fwd_avg * 1e-3) / 1e9,
                         "valu_frac": swept / (fwd_avg * 1e-3) / PAIR_CEILING["forward"]["pairs_per_s"]}
            del plan, img
    finally:
        _cabi.set_default_cutoff(keep)
    return out


def band8_leg(args, dev, t_full_ms=None, world=8, rank=3):
    """BASELINE config 4's strong scaling, priced on ONE GPU: the work of one of eight ranks (rows [3072, 4096) of the 8192^2
    image) timed alone, three ways -- (a) sharded producer, ONE plan over [own | halos] (BandExchange); (b) the same as TWO
    renders (BandExchange(overlap=True): own Gaussians planned and splatted while the halos would travel, the halo records
    added on top; backward halo part first) -- the kernels the overlap form enqueues, back to back; (c) replicated set: the
    band is handed all 1 048 576 Gaussians (broadcast + reduce-scatter pattern).  With T_full the single-GPU step of the
    whole image, T_full / (8 T_band) is the ceiling of the 8-GPU strong-scaling efficiency before any byte moves."""
    from gsasr_amd import _cabi, shard, synthetic
    h_lr, w_lr, scale, _ = CONFIGS["c4"]
    sig, xy, col, H, W = synthetic.kernel_inputs(h_lr, w_lr, scale, seed=0, device="cpu")
    rec = shard.pack(sig, xy, col).to(dev)
    dmax = None if args.dmax < 0 else args.dmax
    rows = shard.row_band(H, rank, world)
    nrows = rows[1] - rows[0]
    grad = synthetic.grad_image(H, W, 1)[rows[0]:rows[1]].contiguous().to(dev)
    slab = torch.empty(nrows, W, 3, device=dev)
    exs = {}
    for r in (rank - 1, rank, rank + 1):
        lr0, lr1 = shard.row_band(h_lr, r, world)
        ex = shard.BandExchange((lr1 - lr0) * w_lr, 16384, H, W, dmax, args.cutoff, device=dev, rank=r, world=world)
        ex.own.copy_(rec[lr0 * w_lr: lr1 * w_lr])
        ex.select()
        exs[r] = ex
    ex = exs[rank]
    n_up, n_down = ex.check()
    ex.from_above.copy_(exs[rank - 1].send_down)
    ex.from_below.copy_(exs[rank + 1].send_up)
    ex.ret.zero_()
    n, tile = ex.n, _cabi.FLAG_BWD_TILE

    def one_plan():
        ex.select()
        p = _cabi.plan_packed(ex.records, H, W, dmax, rows=rows, cutoff=ex.cutoff, flags=tile | ex.plan_flags)
        _cabi.forward(p, slab, overwrite=True)
        _cabi.backward_packed(p, ex.records, grad, ex.g_records, overwrite=True)
        ex.merge()

    def two_renders():
        ex.select()
        po = _cabi.plan_packed(ex.records[:n], H, W, dmax, rows=rows, cutoff=ex.cutoff, flags=tile | ex.plan_flags)
        _cabi.forward(po, slab, overwrite=True)
        ph = _cabi.plan_packed(ex.records[n:], H, W, dmax, rows=rows, cutoff=ex.cutoff, flags=tile)
        _cabi.forward(ph, slab, overwrite=False)
        _cabi.backward_packed(ph, ex.records[n:], grad, ex.g_records[n:], overwrite=True)
        _cabi.backward_packed(po, ex.records[:n], grad, ex.g_records[:n], overwrite=True)
        ex.merge()

    gall = torch.empty_like(rec)

    def replicated():
        p = _cabi.plan_packed(rec, H, W, dmax, rows=rows, cutoff=args.cutoff, flags=tile)
        _cabi.forward(p, slab, overwrite=True)
        _cabi.backward_packed(p, rec, grad, gall, overwrite=True)

    out = {"workload": f"one of {world} row bands of config 4 (rank {rank}: HR rows [{rows[0]},{rows[1]}) of 8192^2, fwd+bwd) on this one GPU",
           "gaussians_own": n, "halo_records": [n_up, n_down], "gaussians_image": int(rec.shape[0])}
    for name, fn in (("sharded_one_plan", one_plan), ("sharded_two_renders_overlap_form", two_renders), ("replicated_all_gaussians", replicated)):
        ms = wall_ms(fn, 20, dev, warm=5, warm_s=0.1)
        out[name] = {"ms_per_step": ms, "value_if_8_ranks": H * W / (ms * 1e-3) / 1e6}
        if t_full_ms:
            out[name]["strong_scaling_ceiling_8"] = t_full_ms / (world * ms)
    out["t_full_ms"] = t_full_ms
    out["note"] = ("no byte moves here: selection, plan(s), forward, backward, merge of ONE rank; the exchange itself is "
                   f"2 x {n_up + n_down} records of 32 B per step (sharded) or 2 x 32 MiB (replicated)")
    return out


def cpu_baseline(args):
    """The oracle's fp32 restatement of the reference kernels (OpenMP over the host cores) on a bounded
    sample of the SAME workload: a row band of config 2 (sized for ~10 s on this host) with all 65 536 Gaussians, forward + backward."""
    from gsasr_amd import synthetic
    from oracle import gs_oracle
    h_lr, w_lr, scale, _ = CONFIGS["c2"]
    sig, xy, col, H, W = synthetic.kernel_inputs(h_lr, w_lr, scale, seed=0)
    dmax = None if args.dmax < 0 else args.dmax
    s, c, k = sig.numpy(), xy.numpy(), col.numpy()
    cores = gs_oracle.num_threads()     # OpenMP team = physical cores of the host (logical CPUs: os.cpu_count())
    full_wgt = synthetic.grad_image(H, W, 1)

    def run(rows):
        wgt = full_wgt[rows[0]:rows[1]].contiguous().numpy()
        t0 = time.perf_counter()
        gs_oracle.forward_f32(s, c, k, H, W, dmax, rows=rows)
        t1 = time.perf_counter()
        gs_oracle.backward_f32(s, c, k, wgt, dmax, h=H, rows=rows)
        return t0, t1, time.perf_counter()

    # size the sample for ~10 s of wall time on whatever host this is: calibrate on 2 rows per thread
    probe = min(H, max(16, 2 * cores))
    t0, _, t2 = run((H // 2 - probe // 2, H // 2 - probe // 2 + probe))
    nrows = int(min(H, max(probe, probe * 10.0 / max(t2 - t0, 1e-3))))
    rows = (H // 2 - nrows // 2, H // 2 - nrows // 2 + nrows)
    t0, t1, t2 = run(rows)
    px = (rows[1] - rows[0]) * W
    out = {"value": px / (t2 - t0) / 1e6, "unit": "HR Mpixels/s", "cores": cores, "logical_cpus": os.cpu_count(), "kind": "port",
           "sample": f"oracle/gs_ref.c fp32 restatement of gs_cuda{'_dmax' if dmax is not None else ''} (OpenMP, {cores} threads), "
                     f"config-2 inputs (N=65536, 1024^2 grid), HR rows [{rows[0]},{rows[1]}) = {px} px, fwd {t1 - t0:.2f}s + bwd {t2 - t1:.2f}s"}
    # the reference's pure-PyTorch path (utils/gaussian_splatting.py rendering_python), BASELINE.json config 1
    try:
        from oracle import host_ref
        torch.manual_seed(0)
        g = torch.randn(4096, 9)
        g[:, 7:9] = torch.rand(4096, 2)
        torch.set_num_threads(cores)            # the same thread count as the oracle's OpenMP team (physical cores)
        t0 = time.perf_counter()
        img1 = host_ref.rendering_python(g, (256, 256), torch.tensor([4.0, 4.0]))
        dt = time.perf_counter() - t0
        # config 1's known answer (tests/golden/rendering_python_config1_stats.npz): what was timed is the path
        assert abs(float(img1.mean()) - 0.21793251) <= 2e-6 and abs(float(img1.max()) - 1.79732013) <= 2e-5, \
            (float(img1.mean()), float(img1.max()))
        out["pytorch_path"] = {"value": 256 * 256 / dt / 1e6, "unit": "HR Mpixels/s (fwd only)",
                               "cores": cores, "kind": "port", "known_answer_checked": True,
                               "sample": f"rendering_python restatement (reference utils/gaussian_splatting.py:11-84), BASELINE.json "
                                         f"config 1 in full: 4096 Gaussians -> 256^2 HR, x4, forward only, {dt:.2f}s"}
    except Exception as e:  # never let the baseline break the bench line
        out["pytorch_path"] = {"error": repr(e)}
    return out


def run_c5e2e(args, dev, rank, world, as_leg=False):
    """BASELINE.json config 5 end to end.  One step = reference
    `optimize_parameters` (TrainTestGSASR/basicsr/models/gsasr_model.py:175-245) with this package's batched rasterizer.
    N > 1: independent replicas (one batch per rank, no collective: the rasterizer path itself does not shard here)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import c5_models
    from gsasr_amd import gaussian_splatting as gsp
    h_lr, w_lr, scale, desc = CONFIGS["c5e2e"]
    B, H, W = 16, int(h_lr * scale), int(w_lr * scale)
    torch.manual_seed(1234 + rank)
    enc, dec = c5_models.EncoderEDSRShaped().to(dev), c5_models.Fea2GSDecoder().to(dev)
    opt = torch.optim.Adam(list(enc.parameters()) + list(dec.parameters()), lr=2e-4)
    lq, gt = torch.rand(B, 3, h_lr, w_lr, device=dev), torch.rand(B, 3, H, W, device=dev)
    sizes, scales = [(H, W)] * B, [scale] * B
    dmax = 0.5 if args.dmax == 0.1 else args.dmax

    def step():
        return c5_models.training_step(enc, dec, opt, lq, gt, sizes, scales, batched=True, dmax=dmax)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 3)):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, params = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    # the rasterizer's share: the same batched splat forward + backward alone, on the decoder's last output
    pd = params.detach()
    g = torch.rand(B, 3, H, W, device=dev)
    sms = [(scale, scale)] * B

    def raster():
        pa = pd.requires_grad_(True)
        gsp.generate_2D_gaussian_splatting_batch(sizes, pa, scales, sms, dmax=dmax).backward(g)
        pa.grad = None

    ms_raster = wall_ms(raster, 20, dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in evs:
        a.record()
        raster()
        b.record()
    torch.cuda.synchronize(dev)
    dev_raster = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    n_params = sum(p.numel() for p in list(enc.parameters()) + list(dec.parameters()))
    if rank != 0:
        return
    out = {"metric": "HR Mpixels/sec, end-to-end training step (x4, 16 Gaussians/LR px, batch 16, L1, Adam)",
           "value": world * B * H * W / (ms * 1e-3) / 1e6, "unit": "HR Mpixels/s", "n_gpus": world, "steps": args.steps,
           "warmup": max(args.warmup, 3), "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": desc, "batch": B, "lr": [h_lr, w_lr], "H": H, "W": W, "gaussians_per_sample": 16 * h_lr * w_lr,
                      "dmax": dmax, "producer_parameters": n_params, "loss": float(loss),
                      "parallelism": f"independent replicas x{world}" if world > 1 else "single",
                      "producers": "tools/c5_models.py: EDSR-baseline-shaped encoder (16 res blocks, 64 ch) + the Fea2GS decoder's architecture at "
                                   "the shipped EDSR-baseline configuration (channel 180, 6 heads, 12x12 windows, 144 seeds, 1 x 2 cross-attention + "
                                   "6 x 6 self-attention layers, pixel-shuffle 2 x 2, five MLP heads), own code, random init"},
           "rasterizer": {"ms_fwd_bwd_wall": ms_raster, "ms_fwd_bwd_device": dev_raster, "share_of_step": dev_raster / ms,
                          "note": "generate_2D_gaussian_splatting_batch forward + backward alone on the decoder's output "
                                  "(prologue + plan + splat, splat backward + chain rule), events on the launch stream"}}
    if world == 1 and not args.no_cpu_baseline:
        # one SAMPLE of the same step on the host cores with the oracle as the rasterizer back end
        from oracle import gs_oracle, host_ref
        cores = gs_oracle.num_threads()
        torch.set_num_threads(cores)
        enc_c, dec_c = c5_models.EncoderEDSRShaped(), c5_models.Fea2GSDecoder()
        enc_c.load_state_dict({k: v.cpu() for k, v in enc.state_dict().items()})
        dec_c.load_state_dict({k: v.cpu() for k, v in dec.state_dict().items()})
        x, y = lq[:1].cpu(), gt[:1].cpu()
        t0 = time.perf_counter()
        p1 = dec_c(enc_c(x), torch.tensor([scale]))[0]
        sig, xy, col, _ = host_ref.prologue(p1, (H, W), torch.tensor([scale, scale]), dmax=dmax)
        img = torch.from_numpy(gs_oracle.forward_f32(sig.detach().numpy(), xy.detach().numpy(), col.detach().numpy(), H, W, dmax))
        grad = torch.sign(img.permute(2, 0, 1)[None] - y) / img.numel()          # d L1 / d image
        gk = gs_oracle.backward_f32(sig.detach().numpy(), xy.detach().numpy(), col.detach().numpy(),
                                    grad[0].permute(1, 2, 0).contiguous().numpy(), dmax)
        torch.autograd.backward([sig, xy, col], [torch.from_numpy(a) for a in gk])
        t1 = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": H * W / t1 / 1e6, "unit": "HR Mpixels/s", "cores": cores, "logical_cpus": os.cpu_count(),
                               "kind": "port", "sample": f"ONE sample of the step (encoder + producer in torch on the CPU, oracle/gs_ref.c "
                               f"fp32 restatement of gs_cuda_dmax forward + backward, autograd to the producers; no optimizer step): {t1:.2f} s"}
    if as_leg:
        return out
    emit(out)




# ---------------------------------------------------------------------------------------------------------------------
# per-stage device times + roofline of one Step (shared by the headline line and by the legs below)
# ---------------------------------------------------------------------------------------------------------------------
def stage_times(step, dev, iters=30):
    """{plan, forward, backward}: average / median device time of each stage, measured with events on the launch stream
    (not part of any timed region), + algorithmic bytes and GB/s where SURVEY.md 8(d) defines them"""
    step.nplans = 0
    kern = {}
    if step.batched:   # step-level entry points: the forward stage includes the prologue and the plan
        nb, px = 36 * step.n_rank, 12 * step.pix_rank
        stages = (("forward", step.batched_forward, nb + px), ("backward", step.batched_backward, 2 * nb + px))
    else:
        stages = (("plan", step.do_plan, None), ("forward", step.do_forward, step.bytes_fwd()),
                  ("backward", step.do_backward, step.bytes_bwd()))
    for name, fn, nbytes in stages:
        if name == "backward" and step.fwd_only:
            continue
        avg, med = time_stage(fn, iters, dev)
        kern[name] = {"avg_ms": avg, "median_ms": med}
        if nbytes:
            kern[name].update({"algorithmic_bytes": nbytes, "GBps": nbytes / (avg * 1e-3) / 1e9})
    return kern


def pair_rates(step, kern, cutoff):
    """swept (Gaussian, pixel) pairs of the step's windows and the fraction of the pair-evaluation ceiling each kernel reaches"""
    tau, _ = effective_tau(step, cutoff)
    in_box, swept = window_pairs(step.sig, step.xy, step.H, step.W, step.dmax, tau, step.rows)
    frac = {k: swept / (kern[k]["avg_ms"] * 1e-3) / pair_ceiling(step, k)["pairs_per_s"] for k in kern if k in PAIR_CEILING}
    return in_box, swept, frac


def single_gpu_leg(args, dev, config, steps=None, **override):
    """One more BASELINE scale factor on THIS GPU, next to the headline line (the north star asks for
    x4 / x8 / x12): the same Step machinery on config `config`, a short timed loop (plain stream launches), per-kernel
    device times and its own roofline.  Sized to take well under a second of GPU time."""
    import copy
    a = copy.copy(args)
    a.config, a.fwd_only, a.force_dist = config, False, False
    for k, v in override.items():
        setattr(a, k, v)
    st = Step(a, dev, 0, 1)
    n = steps or (20 if st.H * st.W > 4e7 else 50)
    # (as many untimed steps as the standalone run of this config gets before its timed loop: three leave the GPU's clocks
    # still ramping after the host-side set-up, which cost the x12 and x8 legs 5-9%)
    ms = wall_ms(st, n, dev, warm=max(10, n // 2), warm_s=0.15)
    kern = stage_times(st, dev, iters=10 if st.H * st.W > 4e7 else 20)
    dom = max((k for k in kern if k != "plan"), key=lambda k: kern[k]["avg_ms"])
    if st.batched:      # (pair counts need the kernel-frame tensors: the batched step keeps raw decoder parameters only)
        in_box, swept, vfrac = None, None, {}
    else:
        in_box, swept, vfrac = pair_rates(st, kern, a.cutoff)
    tau_eff, k_box = effective_tau(st, a.cutoff)
    h_lr, w_lr, scale, desc = CONFIGS[config]
    if st.batched and not st.sampled:      # pair counts of the batched canvas: the kernel-frame tensors of every sample, through the package's own prologue
        try:
            from gsasr_amd import gaussian_splatting as gsp
            hs = ws = int(h_lr * scale)
            in_box = swept = 0
            for b_ in range(st.B):
                sx, sy, rho, xy_, col_ = gsp._activate(st.p[b_].cpu())
                sg, xk, _, _, _ = gsp._to_kernel_frame(sx, sy, rho, xy_, col_, (hs, ws), 1.2 / scale)
                ib, sw = window_pairs(sg, xk, hs, ws, st.dmax, tau_eff, (0, hs))
                in_box, swept = in_box + ib, swept + sw
            vfrac = {k: swept / (kern[k]["avg_ms"] * 1e-3) / PAIR_CEILING[k]["pairs_per_s"] for k in kern if k in PAIR_CEILING}
        except Exception as e:
            print(f"[bench] pair counts of the batched canvas failed: {e!r}", file=sys.stderr)
    # counters of the stages' kernels, COUNTED NOW for this config on this box and build (three short child runs)
    live, live_note = None, "skipped (--no-live-pmc)"
    if not getattr(args, "no_live_pmc", False) and "GSASR_BENCH_CHILD" not in os.environ:
        try:
            live, live_note = live_counters(["--config", config, "--cutoff", str(a.cutoff), "--dmax", str(a.dmax)] + (["--fwd-only"] if a.fwd_only else []))
        except Exception as e:
            live, live_note = None, repr(e)
    traffic = (live or {}).get(dom, {}).get("hbm_bytes")
    valu_busy = {k: v.get("valu_busy") for k, v in (live or {}).items()}
    out = {"workload": desc, "H": st.H, "W": st.W, "gaussians": st.n, "dmax": st.dmax if st.dmax is not None else -1,
           "cutoff_tau": round(tau_eff, 3), "cutoff_k_box": k_box,
           "cutoff_tau_conservative": round(st.cabi.resolve_cutoff(a.cutoff, st.plan.dims.s), 3),
           "what": "fwd only" if st.fwd_only else "fwd+bwd",
           "forward_subtile_px": None if st.batched else st.cabi.forward_subtile_width(st.plan),    # 16 = the wide forward (x5 and up), 8 = the 8 x 16 kernels
           "steps": n, "ms_per_step": ms, "value": st.H * st.W / (ms * 1e-3) / 1e6, "unit": "HR Mpixels/s",
           "kernels": kern,
           "roofline": {"bound": bound_label(valu_busy.get(dom)), "kernel": ", ".join((live or {}).get(dom, {}).get("kernels", [])) or {"forward": "k_render_fwd", "backward": "k_render_bwd"}[dom],
                        "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kern[dom]["GBps"] / HBM_PEAK_GBS,
                        "frac_of_measured_copy": (kern[dom]["GBps"] / args.copy_GBps if getattr(args, "copy_GBps", None) else None),
                        "frac_note": "frac = achieved / peak is the HBM fraction SURVEY 8(d) defines (algorithmic bytes / stage time / 8 TB/s); "
                                     "`bound` names the resource the live counters say binds the kernel",
                        "traffic": traffic, "traffic_over_algorithmic": (round(traffic / kern[dom]["algorithmic_bytes"], 2) if traffic else None),
                        "traffic_source": live_note, "counters": live,
                        "valu_busy": valu_busy, "valu_frac": vfrac, "pairs_in_swept_window": swept, "pairs_in_dmax_box": in_box}}
    del st
    torch.cuda.empty_cache()
    return out


def strong_c4_leg(args, dev, rank, world, steps=5, exchange="broadcast", overlap=False, config="c4"):
    """BASELINE config 4 on whatever group this run has: the 8192^2 image strong-scaled over the ranks' row bands.
    exchange = "broadcast": as the north star states it -- Gaussians broadcast once per step as ONE packed [N,8] buffer,
    per-Gaussian gradients reduce-scattered in place (64 MB per rank and step over xGMI whatever the band height);
    exchange = "halo": the same image with a SHARDED producer -- every rank holds the Gaussians of its own LR rows and swaps
    only those whose footprint crosses a band edge with ranks g-1 / g+1 (shard.BandExchange: a few hundred KB per step).
    Every multi-rank line carries both, so that a real `--gpus 8` run measures them side by side whatever its headline is."""
    import copy
    import torch.distributed as dist
    a = copy.copy(args)
    # (config = the headline's own weak-scaled config: the same leg serves as "the other exchange beside the headline")
    a.config, a.exchange, a.fwd_only, a.force_dist, a.overlap = config, exchange, False, True, overlap
    st = Step(a, dev, rank, world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(8):
        st()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        st()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / steps * 1e3
    kern = stage_times(st, dev, iters=5) if not (st.halo and overlap) else {}
    out = {"workload": CONFIGS[config][3], "scaling": "strong" if st.strong else "weak", "exchange": "halo" if st.halo else "broadcast", "overlap": bool(st.halo and overlap), "transport": (st.ex.transport if st.halo else "collectives"), "n_gpus": world, "steps": steps, "ms_per_step": ms,
           "value": st.H * st.W / (ms * 1e-3) / 1e6, "unit": "HR Mpixels/s", "rows_per_rank": st.rows[1] - st.rows[0],
           "kernels_rank0": kern,
           "rccl": {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                    "bytes_sent_per_rank_per_step": (32 * st.n + 32 * st.n) if world > 1 else 0,
                    "pattern": "broadcast of ONE packed [N,8] buffer (binned where it lands) + in-place reduce_scatter_tensor of the [N,8] gradients"}}
    if st.halo:
        st.ex.check()       # the timed steps dropped nothing (capacity) -- raises otherwise
        out["gaussians_per_rank"] = st.n_rank
        out["rccl"].update({"bytes_sent_per_rank_per_step": 2 * 2 * st.ex.cap * 32 if world > 1 else 0,
                            "halo_records_max_per_edge": st.halo_records, "capacity": st.ex.cap,
                            "pattern": ("2 x all_to_all_single with non-zero splits for ranks g-1/g+1 only" if st.ex.transport == "alltoall"
                                        else "2 x batch_isend_irecv with ranks g-1/g+1") + " (forward records, backward gradients)"})
    elif exchange == "halo":
        out["note"] = "halo exchange unavailable on this group (see stderr): this leg ran broadcast + reduce_scatter"
    del st
    torch.cuda.empty_cache()
    return out
