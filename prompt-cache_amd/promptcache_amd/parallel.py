"""Multi-GPU schema encode: shard the independent scaffold passes over the GPUs of one node and
all-gather the resulting module KV (one process per GPU, ``torch.distributed`` -- backend ``nccl`` is
RCCL over xGMI on ROCm; ``gloo`` on CPU for the tests).

The reference has no distributed code; what shards is its
``SchemaCache._process`` loop (``promptcache/cache_engine.py:217-304``): one forward pass per scaffold
path, no cross-path dependency.  Single-prompt TTFT stays on one GPU (replicas only).

Exchange step: each rank packs the segment stores it owns into ONE flat fp16 buffer; shards are
padded to the largest and exchanged with a single ``all_gather_into_tensor`` (one large collective
instead of one per segment: on the fully connected 8-GPU xGMI node each GPU receives (G-1)/G of the
library over its 7 links concurrently).  Every rank knows every segment's owner and size from the
(deterministic) plan, so no metadata is exchanged, and the received segments are used in place as views
of the gathered buffer (``pc_kv_gather`` takes arbitrary source pointers).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


def rank_world() -> Tuple[int, int]:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_jobs(costs: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time-first partition of jobs (cost = scaffold token count) over ``world``
    ranks.  Deterministic; each rank's list is returned in ascending job order."""
    loads = [0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in sorted(range(len(costs)), key=lambda i: (-costs[i], i)):
        r = min(range(world), key=lambda r: (loads[r], r))
        shards[r].append(i)
        loads[r] += costs[i]
    return [sorted(s) for s in shards]


def allgather_segments(local: Sequence[torch.Tensor], seg_table: Sequence[Tuple[int, int]], rank: int, world: int,
                       device, group=None) -> List[torch.Tensor]:
    """``seg_table[j] = (owner_rank, numel)`` for every segment in global order; ``local`` holds this
    rank's segments in the same relative order.  Returns one flat fp16 tensor per segment, in global
    order, all views of one gathered buffer."""
    import torch.distributed as dist

    per_rank = [0] * world
    offsets = []
    for owner, numel in seg_table:
        offsets.append(per_rank[owner])
        per_rank[owner] += numel
    shard = max(per_rank) if per_rank else 0
    shard = (shard + 7) // 8 * 8  # keep every shard 16-byte aligned
    mine = [j for j, (owner, _) in enumerate(seg_table) if owner == rank]
    if len(mine) != len(local):
        raise ValueError(f"rank {rank} owns {len(mine)} segments but holds {len(local)}")
    dtype = local[0].dtype if local else torch.float16
    send = torch.empty(shard, dtype=dtype, device=device)
    for j, t in zip(mine, local):
        if t.numel() != seg_table[j][1]:
            raise ValueError(f"segment {j}: expected {seg_table[j][1]} elements, got {t.numel()}")
        send[offsets[j]:offsets[j] + t.numel()].copy_(t.reshape(-1))
    recv = torch.empty(world * shard, dtype=dtype, device=device)
    if shard > 0:
        dist.all_gather_into_tensor(recv, send, group=group)
    out = []
    for j, (owner, numel) in enumerate(seg_table):
        base = owner * shard + offsets[j]
        out.append(recv[base:base + numel])
    return out
