#!/usr/bin/env python
"""bench.py -- HR Mpixels/s of the rasterizer hot path (forward + backward) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dmax 0.1] [--config c2|c3|c4|c5|c5s]

A "step" is one pass of the hot path over one batch of synthetic Gaussians already resident in HBM:
plan (bin) -> forward splat into a fresh [H,W,3] image -> backward to {sigmas, coords, colors}, exactly what
`rendering_cuda_dmax(...)` + `.backward()` of gsasr_amd.gaussian_splatting enqueue, through the C ABI.

  N = 1   BASELINE.json config 2: 256x256 LR -> x4 (1024^2 HR), 65 536 Gaussians (1 per LR pixel), fp32.
  N > 1   weak scaling of the row-band shard: the image grows to (1024*N) x 1024 with
          65 536*N Gaussians; rank g renders rows [g*1024,(g+1)*1024).
          --exchange halo (default): rank g holds the Gaussians of its own band (sharded producer); per step
            ONE batched point-to-point swap of the Gaussians that cross a band edge with ranks g-1/g+1 before
            the plan, and ONE swap of their partial gradients after the backward (RCCL send/recv over xGMI).
          --exchange broadcast: ONE broadcast of all [N,8] Gaussians from rank 0 and ONE reduce_scatter of
            the per-Gaussian gradients per step (the literal 8e pattern; volume grows with N).

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel, measured
live with events on the launch stream) and `cpu_baseline` (the oracle on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

sys.path.insert(0, os.path.join(ROOT, "tools"))
from benchlib import (CONFIGS, HBM_PEAK_GBS, PAIR_CEILING, Step, band8_leg, bound_label, pair_ceiling, copy_bandwidth, cpu_baseline, dropin_run, effective_tau, emit,  # noqa: E402
                      exact_runs, live_counters, flush_c_stdio, published_leg, run_c5e2e, single_gpu_leg, stage_times, strong_c4_leg, window_pairs)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dmax", type=float, default=0.1, help="box half-size; <0 = unbounded gs_cuda op")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--cutoff", type=float, default=0.0, help="support cutoff tau (0 = library default: adaptive ln(N/1e-5))")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step in a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not count the dominant kernel's HBM bytes live (two short child runs under rocprofv3 --pmc); "
                         "roofline.traffic is then replayed from profiles/pmc_latest.json")
    ap.add_argument("--no-extras", action="store_true", help="skip the exact-semantics (tau 104 / no cull) and drop-in legs of config 2")
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--exchange", default="halo", choices=["halo", "broadcast"],
                    help="multi-rank data path: neighbour halo swap (default) or broadcast + reduce_scatter")
    ap.add_argument("--overlap", action="store_true",
                    help="halo exchange as TWO renders per band: own Gaussians under the swap, halos added on top (shard.BandExchange(overlap=True))")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank code path even at world size 1 (self-test)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend: nccl (= RCCL over xGMI, the measured configuration) or gloo -- a functional "
                         "self-test of the multi-rank code path on a box with fewer GPUs than ranks (ranks then share devices)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: re-execute this very command line under
    torch.distributed.run, one rank per GPU on this node (rendezvous on 127.0.0.1, a free port).  The ranks inherit
    stdout, so rank 0's JSON line is the last line of this process's output as well; the exit status is the launcher's."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    if args.backend == "nccl" and torch.cuda.device_count() < args.gpus:
        print(f"[bench] --gpus {args.gpus} over RCCL needs {args.gpus} GPUs, this node shows {torch.cuda.device_count()} "
              "(--backend gloo runs the same code path functionally with ranks sharing devices)", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] --gpus {args.gpus} without a launcher: spawning {args.gpus} ranks ({' '.join(cmd[1:9])} ...)", file=sys.stderr)
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dev = torch.device("cuda", local % torch.cuda.device_count() if args.backend == "gloo" else local)
    torch.cuda.set_device(dev)
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"

    if args.config == "c5e2e":
        run_c5e2e(args, dev, rank, world)
        if world > 1 or args.force_dist:
            import torch.distributed as dist
            dist.destroy_process_group()
        return

    step = Step(args, dev, rank, world)

    def barrier():
        if world > 1 or args.force_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    # optional hipGraph of the whole step (single GPU): removes host launch gaps from the measurement
    graph, launch = None, "eager"
    for _ in range(min(3, args.warmup)):
        step()
    torch.cuda.synchronize(dev)
    if world == 1 and not args.no_graph and not args.force_dist:
        try:
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                step()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    step()
            torch.cuda.current_stream(dev).wait_stream(side)
            graph, launch = g, "hipgraph"
        except Exception as e:
            print(f"[bench] graph capture unavailable ({e!r}); timing eager launches", file=sys.stderr)
            graph = None
    run = (lambda: graph.replay()) if graph is not None else step
    if graph is not None:
        # replaying the graph is not always the faster way to enqueue a step: on this stack the five-node graph costs
        # ~7 us per replay more than plain stream launches, which the GPU pipelines back to back.  Time both, keep the
        # faster (the choice is reported as config.launch).
        def quick(fn, n=20):
            fn()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / n
        t_graph = quick(run)
        step.nplans = 0             # (replays left the counters of the captured parity used: the eager path starts over)
        t_eager = quick(step)
        if t_eager < t_graph:
            run, launch = step, "eager (faster than hipgraph replay: %.1f vs %.1f us/step)" % (t_eager * 1e6, t_graph * 1e6)
        else:
            launch = "hipgraph (faster than eager: %.1f vs %.1f us/step)" % (t_graph * 1e6, t_eager * 1e6)

    for _ in range(args.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if step.ex is not None:
        step.ex.check()      # the timed steps dropped nothing (capacity) -- raises otherwise
    if "GSASR_BENCH_CHILD" in os.environ:      # a counter pass of live_counters: the steps above are all it is for
        return

    ms = dt / args.steps * 1e3
    mpix = step.H * step.W / (dt / args.steps) / 1e6        # whole-job HR pixels per second (all ranks)

    # per-kernel device time, measured live (not part of the timed region)
    kern = stage_times(step, dev, iters=30)
    dom = max((k for k in kern if k != "plan"), key=lambda k: kern[k]["avg_ms"])
    traffic, traffic_source, live = None, None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")   # (committed passes: the fallback when rocprofv3 is missing)
    dom_kernel = {"forward": "k_render_fwd", "backward": "k_render_bwd"}[dom]
    if world == 1 and rank == 0 and not args.no_live_pmc and not args.force_dist:
        # counted NOW, on this box, for this build: three child runs of the same workload under rocprofv3 --pmc
        extra = ["--config", args.config, "--dmax", str(args.dmax), "--cutoff", str(args.cutoff)] + (["--fwd-only"] if args.fwd_only else [])
        try:
            live, note = live_counters(extra)
        except Exception as e:      # the counters must never take the line with them
            live, note = None, repr(e)
        if live and live.get(dom, {}).get("hbm_bytes"):
            traffic, traffic_source = live[dom]["hbm_bytes"], note
        else:
            print(f"[bench] live counters unavailable ({note}); replaying profiles/pmc_latest.json", file=sys.stderr)
    if traffic is None and os.path.exists(pmc) and args.config == "c2" and world == 1:
        try:
            traffic = json.load(open(pmc)).get(dom_kernel, {}).get("hbm_bytes")
            traffic_source = ("profiles/pmc_latest.json (REPLAYED: HBM bytes per launch of this kernel from the committed rocprofv3 "
                              "--pmc FETCH_SIZE / WRITE_SIZE passes of the same command, not counted in this run)")
        except Exception:
            traffic = None
    kname = {"forward": "k_sample_fwd", "backward": "k_sample_bwd"} if getattr(step, "sampled", False) else \
        {"forward": "k_render_fwd", "backward": "k_render_bwd"}
    valu_busy_live = {k: v.get("valu_busy") for k, v in (live or {}).items()}
    roofline = {"bound": "hbm", "kernel": ", ".join((live or {}).get(dom, {}).get("kernels", [])) or kname[dom],
                "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": kern[dom]["GBps"] / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_over_algorithmic": round(traffic / kern[dom]["algorithmic_bytes"], 2) if traffic and kern[dom].get("algorithmic_bytes") else None,
                "traffic_source": traffic_source if traffic is not None else None,
                "binding_resource": bound_label(valu_busy_live.get(dom)),
                "bound_note": "`bound` = the roof `frac` is measured against (HBM: SURVEY 8(d)'s algorithmic bytes / stage time / 8 TB/s); the kernel is NOT "
                              "HBM-bound: `binding_resource` is what the live SQ counters say binds it (valu_busy = VALU issue occupancy)",
                "valu_busy": valu_busy_live, "counters": live,
                "note": "stage times from events on the launch stream; outputs are stored, not accumulated (no memsets)"}
    # The binding unit is the fp32 VALU / v_exp_f32 pipe, not HBM: report pair rates and the VALU
    # occupancy next to the HBM fraction.
    if rank == 0:
        try:
            roofline["hbm_copy_GBps_measured"] = round(copy_bandwidth(dev), 1)
            roofline["frac_of_measured_copy"] = kern[dom]["GBps"] / roofline["hbm_copy_GBps_measured"]
        except Exception as e:
            roofline["hbm_copy_GBps_measured"] = None
            print(f"[bench] copy bandwidth probe failed: {e!r}", file=sys.stderr)
        if not step.halo and not step.batched:
            tau, _ = effective_tau(step, args.cutoff)
            in_box, swept = window_pairs(step.sig, step.xy, step.H, step.W, step.dmax, tau, step.rows)
            valu = {"pairs_in_dmax_box": in_box, "pairs_in_swept_window": swept,
                    "Gpairs_per_s": {k: round(swept / (kern[k]["avg_ms"] * 1e-3) / 1e9, 1) for k in kern if k != "plan"},
                    "note": "pairs = (Gaussian, pixel) terms; box = what gs_cuda_dmax sums, swept window = box ∩ support cutoff"}
            # fraction of the pair-evaluation ceiling (the roof that binds): swept pairs/s over SIMDs * clock / cycles per
            # packed trip * 128 pairs, with the instruction counts the ceiling assumes
            valu["ceiling"] = {k: pair_ceiling(step, k) for k in kern if k in PAIR_CEILING}
            valu["ceiling_note"] = ("1024 SIMDs x 2.4 GHz; wave64 VALU (packed fp32 included) = 4 cycles, v_exp_f32 = 8 cycles "
                                    "(tools/valu_rate.hip); one packed trip = 128 pairs")
            roofline["valu_frac"] = {k: swept / (kern[k]["avg_ms"] * 1e-3) / pair_ceiling(step, k)["pairs_per_s"]
                                     for k in kern if k in PAIR_CEILING}
            if valu_busy_live:
                valu["valu_busy_rocprof"] = dict(valu_busy_live, source="counted live in this run (roofline.counters)")
            elif os.path.exists(pmc) and args.config == "c2" and world == 1:
                try:
                    j = json.load(open(pmc))
                    valu["valu_busy_rocprof"] = {"forward": j["k_render_fwd"].get("valu_busy"),
                                                 "backward": j["k_render_bwd"].get("valu_busy"),
                                                 "source": "profiles/pmc_latest.json (REPLAYED: SQ_INSTS_VALU, SQ_BUSY_CYCLES of the committed passes)"}
                except Exception:
                    pass
            roofline["valu"] = valu

    c4_strong = c4_strong_halo = c4_strong_halo_overlap = other_exchange = None
    if (world > 1 or args.force_dist) and args.config == "c2" and not step.batched:
        # the headline's own weak-scaled workload over the OTHER exchange, beside it in the same line (halo first, the north
        # star's broadcast + reduce-scatter next to it), with its own rccl block
        try:
            other_exchange = strong_c4_leg(args, dev, rank, world, exchange="broadcast" if args.exchange == "halo" else "halo", config="c2")
        except Exception as e:
            other_exchange = {"error": repr(e)}
    if (world > 1 or args.force_dist) and args.config in ("c2", "c4") and not step.batched:
        # BASELINE config 4's own pattern -- strong-scaled 8192^2, packed broadcast + in-place reduce-scatter -- as a leg of
        # EVERY multi-rank line, whatever exchange the headline used
        try:
            c4_strong = strong_c4_leg(args, dev, rank, world)
        except Exception as e:
            c4_strong = {"error": repr(e)}
        # ... and the same image with a sharded producer and the nearest-neighbour halo swap (what DESIGN.md section 5 proposes
        # for this path: 64 MB per rank and step of collectives become a few hundred KB)
        try:
            c4_strong_halo = strong_c4_leg(args, dev, rank, world, exchange="halo")
        except Exception as e:
            c4_strong_halo = {"error": repr(e)}
        # ... and with the swap hidden behind the own Gaussians' render (two renders per band)
        try:
            c4_strong_halo_overlap = strong_c4_leg(args, dev, rank, world, exchange="halo", overlap=True)
        except Exception as e:
            c4_strong_halo_overlap = {"error": repr(e)}
    if world > 1:      # every rank's C-stdio output (RCCL banner) is out before rank 0 prints the line
        import torch.distributed as dist
        flush_c_stdio()
        sys.stdout.flush()
        dist.barrier()
    if rank == 0:
        h_lr, w_lr, scale, desc = CONFIGS[args.config]
        tau_eff, k_box = effective_tau(step, args.cutoff)      # (read back from the plan header: data-derived under the default)
        out = {
            "metric": ("HR Mpixels/sec fwd+bwd (x4, 1 Gaussian/LR px); achieved HBM GB/s vs roofline" if args.config == "c2"
                       else "sampled HR Mpixels/sec prologue+fwd+bwd (x4, 16 Gaussians/LR px, batch 16, sample_coords)" if getattr(step, "sampled", False)
                       else "HR Mpixels/sec prologue+fwd+bwd (x4, 16 Gaussians/LR px, batch 16)" if step.batched else f"HR Mpixels/sec {'fwd only' if step.fwd_only else 'fwd+bwd'} (x{scale:g}, {16 if args.config == 'c2x16' else 1} Gaussian{'s' if args.config == 'c2x16' else ''}/LR px)"),
            "value": mpix, "unit": "HR Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong" if step.strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc + (f"; weak-scaled to {step.H}x{step.W} HR / {step.n} Gaussians over {world} row bands"
                                           if world > 1 and not step.strong and not step.batched else ""),
                       "H": step.H, "W": step.W, "gaussians": step.n, "dmax": step.dmax if step.dmax is not None else -1,
                       "cutoff_tau": round(tau_eff, 3), "cutoff_k_box": k_box,
                       "cutoff_tau_conservative": round(step.cabi.resolve_cutoff(args.cutoff, step.plan.dims.s), 3),
                       "cutoff_note": "terms with exponent < -tau are skipped; default tau = ln(K/1e-5) with K = the plan's own bound on how many "
                                      "Gaussians can lose a non-negligible term on one pixel -- counted from the plan's cell histogram over the class' largest support, "
                                      "and for the bounded op also over the dmax box (gs_cuda_dmax/gs.cu:41-50), the smaller count wins; K <= N -- "
                                      "bounds the image error by 1e-5 x max|colour| per pixel for any input (colours are <= 1 after the host prologue; "
                                      "parity tolerance 1e-4, all parity tests run at this default); "
                                      "--cutoff 104 sums the reference's exact set of non-zero fp32 terms, --cutoff -1 every in-box term",
                       "launch": launch,
                       "forward_subtile_px": None if step.batched else step.cabi.forward_subtile_width(step.plan),   # 16 = wide forward (x5 and up)
                       "parallelism": (f"data-parallel x{world} (one batch per rank, no exchange)" if step.batched and world > 1
                                       else f"row-band x{world}" if world > 1 else "single")},
            "roofline": roofline, "kernels": kern,
        }
        if step.dist:
            out["config"]["exchange"] = (f"halo swap with ranks g-1/g+1 ({step.ex.transport if step.halo else ''}), "
                                         f"{step.halo_records} records max per edge, capacity {step.ex.cap}"
                                         if step.halo else "broadcast of all Gaussians + reduce_scatter of their gradients")
        if step.dist:
            # self-check of the first real multi-GPU run: how many ranks RCCL saw and what each moved per step
            import torch.distributed as dist
            per_rank = (2 * 2 * step.ex.cap * 32 if step.halo else 32 * step.n + 32 * step.n)
            out["config"]["rccl"] = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                                     "bytes_sent_per_rank_per_step": per_rank if world > 1 else 0,
                                     "pattern": (("2 x all_to_all_single with non-zero splits for ranks g-1/g+1 only" if step.ex.transport == "alltoall"
                                                  else "2 x batch_isend_irecv with ranks g-1/g+1") + " (forward records, backward gradients)")
                                                if step.halo else "broadcast of ONE packed [N,8] buffer + in-place reduce_scatter_tensor of the [N,8] gradients"}
        if world == 1 and args.config == "c2" and not args.no_extras and not args.force_dist:
            out["exact"] = exact_runs(args, dev, step.H * step.W)
            out["dropin"] = dropin_run(args, dev)
            # the other scale factors the north star names (x12 forward only = BASELINE config 3, x8 fwd+bwd = config 4 on this
            # ONE GPU) and GSASR's real Gaussian density at x4: each leg with its own kernel times and roofline
            step = None             # (its buffers make room for the 8192^2 leg)
            torch.cuda.empty_cache()
            out["configs"] = {}
            args.copy_GBps = roofline.get("hbm_copy_GBps_measured")
            for name in ("c3", "c4", "c2x16", "c5"):
                try:
                    out["configs"][name] = single_gpu_leg(args, dev, name)
                except Exception as e:      # a leg must never take the headline line with it
                    out["configs"][name] = {"error": repr(e)}
            # BASELINE config 5 END TO END (encoder + producer -> batched splat -> L1 -> backward -> Adam), a handful of steps
            try:
                import copy
                a5 = copy.copy(args)
                a5.config, a5.steps, a5.warmup, a5.no_cpu_baseline = "c5e2e", 5, 3, True
                out["configs"]["c5e2e"] = run_c5e2e(a5, dev, 0, 1, as_leg=True)
            except Exception as e:
                out["configs"]["c5e2e"] = {"error": repr(e)}
            # one of eight ranks' share of config 4, alone on this GPU: the strong-scaling ceiling before hardware exists
            try:
                t_full = out["configs"].get("c4", {}).get("ms_per_step")
                out["c4_band8"] = band8_leg(args, dev, t_full)
            except Exception as e:
                out["c4_band8"] = {"error": repr(e)}
            # the reference's one published measurement of this path (utils/gs_cuda/profile.py:104-113, profile.log:44)
            try:
                out["published"] = published_leg(args, dev)
            except Exception as e:
                out["published"] = {"error": repr(e)}
        if other_exchange is not None:
            out["weak_other_exchange"] = other_exchange
        if c4_strong is not None:
            out["c4_strong"] = c4_strong
        if c4_strong_halo is not None:
            out["c4_strong_halo"] = c4_strong_halo
        if c4_strong_halo_overlap is not None:
            out["c4_strong_halo_overlap"] = c4_strong_halo_overlap
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        emit(out)
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
