"""Multi-GPU schema encode: shard the independent scaffold passes over the GPUs of one node and
all-gather the resulting module KV (one process per GPU, ``torch.distributed`` -- backend ``nccl`` is
RCCL over xGMI on ROCm; ``gloo`` on CPU for the tests).

The reference has no distributed code; what shards is its
``SchemaCache._process`` loop (``promptcache/cache_engine.py:217-304``): one forward pass per scaffold
path, no cross-path dependency.  Single-prompt TTFT stays on one GPU (replicas only).

Exchange step (``exchange_slabs``): a rank's encode writes the segment stores it owns straight into ONE
contiguous slab (``carve``: the stores ARE views of the slab, there is no pack copy); every rank then
receives every other rank's slab at its exact size -- one broadcast per owner, issued back to back and
asynchronously (what an uneven all-gather is on RCCL: grouped point-to-point transfers over the fully
connected xGMI links; no padding to the largest shard travels).  Every rank knows every segment's owner
and size from the (deterministic) plan, so no metadata is exchanged, and the received segments are used
in place as views of the received slabs (``pc_kv_gather`` takes arbitrary source pointers).  The handles
are returned to the caller: a library encode overlaps the exchange of schema k with the encode of
schema k + 1 (``CacheEngine.add_schemas``).

``allgather_segments`` is the round-1 form (pack + one max-padded ``all_gather_into_tensor``), kept for
callers that hold loose segment tensors.
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


def carve(sizes: Sequence[int], dtype, device, align: int = 8) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """One contiguous slab holding segments of ``sizes`` elements back to back (each start ``align``-element = 16-byte
    aligned) -> (slab, [flat view per segment]).  The encode writes the stores it owns through these views."""
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + align - 1) // align * align
    slab = torch.empty(total, dtype=dtype, device=device)
    return slab, [slab[o:o + n] for o, n in zip(offs, sizes)]


def exchange_slabs(mine: torch.Tensor, sizes_by_rank: Sequence[Sequence[int]], rank: int, world: int, device, dtype=None,
                   group=None, async_op: bool = False):
    """Every rank contributes one slab laid out by ``carve(sizes_by_rank[rank])``; returns ``(views_by_rank, handles)``:
    ``views_by_rank[r]`` = the flat per-segment views of rank r's slab (this rank's own slab is ``mine`` itself, the
    others are received at their exact size: nothing is padded, nothing is re-packed).  One broadcast per non-empty
    owner, in rank order on every rank (collective call order must match).  ``async_op``: return the pending work
    handles instead of waiting -- the caller keeps computing and waits before the segments are read."""
    import torch.distributed as dist
    dtype = dtype or mine.dtype
    views_by_rank, handles = [], []
    for r in range(world):
        if r == rank:
            slab, views = mine, carve_views(mine, sizes_by_rank[r])
        else:
            slab, views = carve(sizes_by_rank[r], dtype, device)
        views_by_rank.append(views)
        if slab.numel() > 0:
            h = dist.broadcast(slab, src=r if group is None else dist.get_global_rank(group, r), group=group, async_op=True)
            handles.append(h)
    if not async_op:
        for h in handles:
            h.wait()
        handles = []
    return views_by_rank, handles


def carve_views(slab: torch.Tensor, sizes: Sequence[int], align: int = 8) -> List[torch.Tensor]:
    """The per-segment views of a slab that ``carve(sizes)`` laid out."""
    out, off = [], 0
    for n in sizes:
        out.append(slab[off:off + n])
        off += (n + align - 1) // align * align
    if off != slab.numel():
        raise ValueError(f"slab of {slab.numel()} elements does not match the plan ({off})")
    return out
