"""The multi-rank code path of bench.py (row-band shard: halo exchange, and broadcast + reduce-scatter) executed for real
with two processes -- on ONE GPU, over gloo, as a functional self-test: the development boxes have a single GPU and
RCCL refuses two ranks on one device.  Timing from such a run means nothing; what is checked is that every rank gets
through plan / exchange / forward / backward / exchange and that rank 0 prints the contract's JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LAUNCHER = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", "29541"]


@pytest.mark.gpu
@pytest.mark.parametrize("launcher,extra", [([sys.executable], ["--exchange", "halo"]),
                                            ([sys.executable], ["--exchange", "broadcast"]),
                                            ([sys.executable], ["--config", "c4"]),
                                            (LAUNCHER, ["--exchange", "halo"])],
                         ids=["self-weak-halo", "self-weak-broadcast", "self-strong-c4", "torchrun-weak-halo"])
def test_bench_two_ranks_functional(launcher, extra):
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE in the environment spawns its own ranks (the form
    the driver uses at N = 1); under `python -m torch.distributed.run ... bench.py --gpus 2` (the driver's N > 1 form)
    it must not spawn again."""
    cmd = [*launcher, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--no-cpu-baseline", *extra]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["higher_is_better"] is True
    assert d["config"]["rccl"]["ranks"] == 2 and d["config"]["rccl"]["bytes_sent_per_rank_per_step"] > 0
    assert d["scaling"] == ("strong" if "c4" in extra else "weak")
    assert set(("roofline", "kernels", "metric", "unit", "ms_per_step", "dtype", "data")) <= set(d)
    # every multi-rank line carries BASELINE config 4's own pattern: strong-scaled 8192^2, packed broadcast + reduce-scatter
    leg = d["c4_strong"]
    assert "error" not in leg, leg
    assert leg["scaling"] == "strong" and leg["n_gpus"] == 2 and leg["value"] > 0 and leg["rows_per_rank"] == 4096
    assert leg["rccl"]["ranks"] == 2 and leg["rccl"]["bytes_sent_per_rank_per_step"] == 64 * 1048576
    assert "reduce_scatter" in leg["rccl"]["pattern"] and "broadcast" in leg["rccl"]["pattern"]
    # ... and the same image with the sharded producer + nearest-neighbour halo swap
    halo = d["c4_strong_halo"]
    assert "error" not in halo and "note" not in halo, halo
    assert halo["scaling"] == "strong" and halo["n_gpus"] == 2 and halo["value"] > 0 and halo["rows_per_rank"] == 4096
    assert halo["gaussians_per_rank"] == 1048576 // 2
    assert 0 < halo["rccl"]["bytes_sent_per_rank_per_step"] < 64 * 1048576 // 8
    assert halo["rccl"]["halo_records_max_per_edge"] <= halo["rccl"]["capacity"]
    # ... and with the swap hidden behind the own Gaussians' render (BandExchange(overlap=True)'s sequence), same transport
    ov = d["c4_strong_halo_overlap"]
    assert "error" not in ov and "note" not in ov, ov
    assert ov["overlap"] is True and ov["transport"] == halo["transport"] and ov["value"] > 0 and ov["rows_per_rank"] == 4096


@pytest.mark.gpu
def test_rccl_world_size_one_runs_every_collective_of_the_shard():
    """no multi-GPU box has run this path yet, so the first contact with RCCL happens HERE -- a one-rank
    "nccl" group on the test GPU, every collective the shard issues called once in the shard's own form (tools/rccl_world1.py)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1.py")], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d.pop("backend") == "nccl" and d.pop("ranks") == 1
    assert len(d) == 8 and all(v == "ok" for v in d.values()), d


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["halo", "broadcast"])
def test_bench_force_dist_over_rccl(exchange):
    """`bench.py --force-dist --backend nccl`: the multi-rank code path of the bench -- process group, exchange, barriers, the
    rccl blocks of the line -- on a one-rank RCCL group; the halo form and the north star's broadcast form side by side"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--backend", "nccl", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--exchange", exchange]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["rccl"]["backend"] == "nccl" and d["config"]["rccl"]["ranks"] == 1
    other = d["weak_other_exchange"]
    assert "error" not in other, other
    assert other["exchange"] == ("broadcast" if exchange == "halo" else "halo") and other["rccl"]["ranks"] == 1 and other["value"] > 0
    for leg in ("c4_strong", "c4_strong_halo", "c4_strong_halo_overlap"):
        assert "error" not in d[leg], (leg, d[leg])
        assert d[leg]["rccl"]["backend"] == "nccl" and d[leg]["rccl"]["ranks"] == 1 and d[leg]["value"] > 0
