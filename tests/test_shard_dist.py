"""N>1 path on CPU: the LPT utterance partition (Python rule and the library's jb_lpt_partition,
which the multi-device entries split by), the rendezvous / max-over-ranks plumbing, and bench.py's
own launcher + rank code with world_size 2 over gloo (no GPU)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jbonsai_amd import shard, synth


def test_lpt_partition_properties():
    lens = synth.mixed_lengths(4096)
    assert min(lens) >= 400 and max(lens) <= synth.T_128S
    assert lens == synth.mixed_lengths(4096)  # seed-fixed
    for world in (1, 2, 4, 8):
        parts = shard.lpt_partition(lens, world)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(lens)))
        assert shard.imbalance(lens, parts) < 1.001
    assert shard.lpt_partition([], 4) == [[], [], [], []]
    assert shard.lpt_partition([5], 2) == [[0], []]
    with pytest.raises(ValueError):
        shard.lpt_partition([1], 0)


def test_native_lpt_equals_python_rule():
    """jb_lpt_partition (jb_multi.cpp: what jb_synthesize_batch_multi / jb_paramgen_vocode_batch_multi
    split by) and shard.lpt_partition (what bench.py's ranks split by) are the same rule."""
    lens = synth.mixed_lengths(4096)
    for world in (1, 2, 3, 8):
        assert shard.lpt_partition_native(lens, world) == shard.lpt_partition(lens, world)
    ties = [7, 7, 7, 3, 3, 9, 0, 0, 7]
    for world in (1, 2, 4, 16):
        assert shard.lpt_partition_native(ties, world) == shard.lpt_partition(ties, world)
    assert shard.lpt_partition_native([], 3) == [[], [], []]
    import jbonsai_amd as J
    with pytest.raises(J.JbError):
        shard.lpt_partition_native([1], 0)


def test_bench_launcher_spawns_ranks_and_plans_config3():
    """`bench.py --gpus 2` with no torchrun environment spawns its two ranks itself; in dry-run mode
    (no GPU) the ranks rendezvous over gloo, build their LPT shares of the config-3 job (pdf row
    indices, sub-batches), exchange per-rank records and run the variable-length gather."""
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["JB_BENCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--utts", "96"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] and d["n_gpus"] == 2 and d["gather_ms"] is not None
    # one identity record per rank (PCI bus id and RCCL-reported communicator size are None without a GPU)
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and all("pci_bus_id" in r and "comm_size_from_rccl" in r for r in d["ranks"])
    plan = d["config3_plan"]
    lens = synth.mixed_lengths(96)
    assert sum(plan["per_rank_frames"]) == sum(lens)
    assert plan["per_rank_frames"] == [sum(lens[i] for i in p) for p in shard.lpt_partition(lens, 2)]
    assert plan["imbalance_max_over_mean_frames"] < 1.05 and plan["scaling"] == "strong"
    assert "error" not in plan


def _children_of(pid):
    out = subprocess.run(["ps", "-o", "pid=", "--ppid", str(pid)], capture_output=True, text=True).stdout.split()
    return [int(x) for x in out]


def test_bench_launcher_deadline_and_signals():
    """The launcher never waits for ever and never leaves ranks behind: with hanging ranks it ends them at
    JB_BENCH_TIMEOUT_S and exits 124; on SIGTERM it ends them and exits 128+15."""
    import signal
    import time

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(JB_BENCH_DRYRUN="1", JB_BENCH_DRYRUN_SLEEP_S="300", JB_BENCH_TIMEOUT_S="12")
    t0 = time.monotonic()
    p = subprocess.Popen([sys.executable, str(root / "bench.py"), "--gpus", "2", "--utts", "8"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    kids = []
    while time.monotonic() - t0 < 20 and len(kids) < 2:
        kids = _children_of(p.pid)
        time.sleep(0.2)
    assert len(kids) == 2
    assert p.wait(timeout=120) == 124
    assert time.monotonic() - t0 < 90
    time.sleep(0.5)
    for k in kids:
        assert not Path(f"/proc/{k}").exists() or "Z" in Path(f"/proc/{k}/stat").read_text().split()[2]
    env["JB_BENCH_TIMEOUT_S"] = "600"
    p = subprocess.Popen([sys.executable, str(root / "bench.py"), "--gpus", "2", "--utts", "8"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t0 = time.monotonic()
    kids = []
    while time.monotonic() - t0 < 20 and len(kids) < 2:
        kids = _children_of(p.pid)
        time.sleep(0.2)
    assert len(kids) == 2
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=60) == 128 + signal.SIGTERM
    time.sleep(0.5)
    for k in kids:
        assert not Path(f"/proc/{k}").exists()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = synth.mixed_lengths(64)
    mine = shard.shard_for_rank(lens, rank, world)
    # each rank "processes" its shard; whole-job units = sum over ranks, time = max over ranks
    units = torch.tensor([float(sum(lens[i] for i in mine))], dtype=torch.float64)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(units, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    got = [None] * world
    dist.all_gather_object(got, mine)
    if rank == 0:
        q.put((units.item(), t.item(), got))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    units, tmax, shards = q.get(timeout=120)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    lens = synth.mixed_lengths(64)
    assert units == float(sum(lens))
    assert tmax == 2.0
    assert sorted(i for s in shards for i in s) == list(range(64))
