"""N>1 path on CPU: the LPT utterance partition and the rendezvous / max-over-ranks
plumbing bench.py uses, with world_size-2 gloo (no GPU)."""
import os
import socket

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
