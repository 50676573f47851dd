"""Multi-process (gloo, world_size 2, CPU) tests of the schema-encode sharding and the module-KV all-gather
(promptcache_amd/parallel.py): the N > 1 path of the cache engine, minus the GPU forward passes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from promptcache_amd import parallel


def test_shard_jobs_is_a_balanced_deterministic_partition():
    costs = [456, 815, 600, 612, 590, 777, 480, 501, 733, 640, 455, 700, 690]
    for world in (1, 2, 4, 8):
        shards = parallel.shard_jobs(costs, world)
        assert sorted(i for s in shards for i in s) == list(range(len(costs)))      # partition
        assert all(s == sorted(s) for s in shards)                                   # ascending per rank
        loads = [sum(costs[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(costs)                                 # LPT bound
        assert shards == parallel.shard_jobs(costs, world)                           # deterministic
    assert parallel.shard_jobs([], 4) == [[], [], [], []]
    assert parallel.shard_jobs([5, 5], 4) == [[0], [1], [], []]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert parallel.rank_world() == (rank, world)
        # 7 "scaffold passes" with ragged numbers of owned segments; a pass's KV is a pure function of
        # (job, segment) so every rank can check what it receives
        costs = [30, 10, 25, 5, 40, 12, 18]
        seg_lens = [[3, 1], [1], [7, 2, 1], [], [4], [1, 1, 1], [9]]
        unit = 16                                     # elements per token (stands for L*2*Hkv*D)
        shards = parallel.shard_jobs(costs, world)
        owner = {i: r for r, idxs in enumerate(shards) for i in idxs}

        def kv(job, j, n):
            base = 1000 * job + 100 * j
            return (torch.arange(n * unit, dtype=torch.float32) * 0.25 + base).to(torch.float16)

        order = [(i, j, n) for i in range(len(costs)) for j, n in enumerate(seg_lens[i])]
        table = [(owner[i], n * unit) for i, _, n in order]
        local = [kv(i, j, n) for i, j, n in order if owner[i] == rank]
        got = parallel.allgather_segments(local, table, rank, world, "cpu")
        assert len(got) == len(order)
        for (i, j, n), t in zip(order, got):
            assert torch.equal(t, kv(i, j, n)), (i, j)
        # every segment is a 16-byte-aligned view of ONE gathered buffer
        base = min(t.data_ptr() for t in got)
        assert all((t.data_ptr() - base) % 16 == 0 for t in got)
        assert len({t.untyped_storage().data_ptr() for t in got}) == 1
        # mismatch between the plan and what a rank holds is an error, not silent corruption
        try:
            parallel.allgather_segments(local[:-1], table, rank, world, "cpu")
            ok = len(local) == 0
        except ValueError:
            ok = True
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_allgather_segments_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(0, True), (1, True)]
