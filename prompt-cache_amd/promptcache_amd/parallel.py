"""Multi-GPU schema encode: shard the independent scaffold passes over the GPUs of one node and
all-gather the resulting module KV (one process per GPU, ``torch.distributed`` -- backend ``nccl`` is
RCCL over xGMI on ROCm; ``gloo`` on CPU for the tests).

The reference has no distributed code; what shards is its
``SchemaCache._process`` loop (``promptcache/cache_engine.py:217-304``): one forward pass per scaffold
path, no cross-path dependency.  Single-prompt TTFT stays on one GPU (replicas only).

Exchange step (``exchange_slabs``): a rank's encode writes the segment stores it owns straight into ONE
contiguous slab (``carve``: the stores ARE views of the slab, there is no pack copy); every rank then
receives every other rank's slab at its exact size in ONE grouped point-to-point step
(``dist.batch_isend_irecv`` = ``ncclGroupStart`` ... ``ncclSend`` / ``ncclRecv`` ... ``ncclGroupEnd`` on RCCL):
an owner's slab leaves for its G - 1 peers over G - 1 different xGMI links at once and every rank ingests
from all of its peers concurrently -- the direct, non-ring form SURVEY section 8e asks for on the fully
connected node (a broadcast per owner is a ring: its wire time is total KV / ONE link).  Nothing is padded
to the largest shard.  Every rank knows every segment's owner and size from the (deterministic) plan, so
no metadata is exchanged, and the received segments are used in place as views of the received slabs
(``pc_kv_gather`` takes arbitrary source pointers).  The handles are returned to the caller: a library
encode overlaps the exchange of schema k with the encode of schema k + 1 (``CacheEngine.add_schemas``).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

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


def plan_library(items: Sequence[tuple], world: int) -> Tuple[List[List[List[int]]], List[int]]:
    """Schedule a schema LIBRARY over ``world`` ranks.  ``items[k] = (trunk, costs[, needs_trunk])``: ``costs[i]`` = rows pass i of
    schema k runs through the model on its own, ``trunk`` = rows a rank has to run first when it takes a pass that BUILDS ON the
    trunk (the root scaffold's prefix the suffix passes read; 0 when the passes are independent), ``needs_trunk[i]`` = whether
    pass i is such a pass (default: all of them; a scaffold encoded in full is not, and a rank that holds only those is not
    charged).  Returns ``(shards, loads)``: ``shards[k][r]`` = the passes of schema k rank r encodes (ascending), ``loads[r]`` =
    the rows rank r runs in total.

    Whole schemas first (longest-processing-time-first: every trunk computed once, on the rank that needs it), then the
    residual imbalance is levelled at PASS granularity: the schemas of the most loaded ranks are poured over the ranks below a
    common water line, each rank that receives trunk-dependent passes paying the schema's trunk once (recomputing a few hundred
    trunk rows beats shipping ~1 MB per row of trunk K/V and its split-precision residuals behind the owner's encode).  The
    water line is searched for the smallest makespan this greedy reaches; deterministic, identical on every rank."""
    K = len(items)
    trunks = [it[0] for it in items]
    costs_of = [list(it[1]) for it in items]
    needs = [list(it[2]) if len(it) > 2 else [True] * len(it[1]) for it in items]
    total = [(trunks[k] if any(needs[k]) else 0) + sum(costs_of[k]) for k in range(K)]

    def lpt_whole():
        loads = [0] * world
        owner = [0] * K
        for k in sorted(range(K), key=lambda k: (-total[k], k)):
            r = min(range(world), key=lambda r: (loads[r], r))
            owner[k] = r
            loads[r] += total[k]
        return owner, loads

    def as_shards(assign):
        return [[sorted(assign[k].get(r, [])) for r in range(world)] for k in range(K)]

    def holds_trunk(k, passes):
        return any(needs[k][i] for i in passes)

    owner, loads0 = lpt_whole()
    best_assign = [{owner[k]: list(range(len(costs_of[k])))} for k in range(K)]
    best_loads = list(loads0)
    if world == 1 or K == 0:
        return as_shards(best_assign), best_loads
    ideal = sum(total) / world
    for eps in (0.0, 0.02, 0.04, 0.07, 0.1, 0.15, 0.2, 0.3, 0.45, 0.7, 1.0):
        line = ideal * (1.0 + eps)
        # schemas that fit under the line as a whole keep one owner (largest first onto the least loaded rank that still has
        # room); the others are poured
        loads = [0.0] * world
        assign: List[dict] = [dict() for _ in range(K)]
        poured = []
        for k in sorted(range(K), key=lambda k: (-total[k], k)):
            costs = costs_of[k]
            r = min(range(world), key=lambda r: (loads[r], r))
            if loads[r] + total[k] <= line or len(costs) <= 1:
                assign[k][r] = list(range(len(costs)))
                loads[r] += total[k]
            else:
                poured.append(k)
        for k in poured:
            trunk, costs = trunks[k], costs_of[k]
            rest = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
            while rest:
                r = min(range(world), key=lambda r: (loads[r], r))
                mine = assign[k].setdefault(r, [])
                took = False
                for i in list(rest):                      # largest passes that still fit under the line
                    extra = trunk if (needs[k][i] and not holds_trunk(k, mine)) else 0     # the first trunk-dependent pass brings the trunk
                    if loads[r] + extra + costs[i] <= line or not took:
                        mine.append(i)
                        loads[r] += extra + costs[i]
                        rest.remove(i)
                        took = True
        # local improvement: passes move from the most loaded rank to the least loaded one while that lowers the makespan
        for _ in range(4 * sum(len(c) for c in costs_of)):
            hi = max(range(world), key=lambda r: (loads[r], -r))
            lo = min(range(world), key=lambda r: (loads[r], r))
            move = None
            for k in range(K):
                mine = assign[k].get(hi)
                if not mine or len(costs_of[k]) <= 1:
                    continue
                trunk, costs = trunks[k], costs_of[k]
                theirs = assign[k].get(lo, [])
                for i in mine:
                    join = trunk if (needs[k][i] and not holds_trunk(k, theirs)) else 0          # lo starts running the trunk
                    leave = trunk if (needs[k][i] and not holds_trunk(k, [j for j in mine if j != i])) else 0   # hi stops
                    new_hi, new_lo = loads[hi] - costs[i] - leave, loads[lo] + costs[i] + join
                    gain = loads[hi] - max(new_hi, new_lo)
                    if gain > 1e-9 and (move is None or gain > move[0]):
                        move = (gain, k, i, new_hi, new_lo)
            if move is None:
                break
            _, k, i, new_hi, new_lo = move
            assign[k][hi].remove(i)
            if not assign[k][hi]:
                del assign[k][hi]
            assign[k].setdefault(lo, []).append(i)
            loads[hi], loads[lo] = new_hi, new_lo
        if max(loads) < max(best_loads) - 1e-9:
            best_assign, best_loads = assign, [int(round(v)) for v in loads]
    return as_shards(best_assign), best_loads


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
                   group=None, async_op: bool = False, rank_map: Optional[Sequence[int]] = None):
    """Every rank contributes one slab laid out by ``carve(sizes_by_rank[rank])``; returns ``(views_by_rank, handles)``:
    ``views_by_rank[r]`` = the flat per-segment views of rank r's slab (this rank's own slab is ``mine`` itself, the
    others are received at their exact size: nothing is padded, nothing is re-packed).  ONE grouped point-to-point
    step: this rank's slab goes to every peer and every non-empty peer slab is received, all in one
    ``batch_isend_irecv`` (RCCL: one ncclGroup -- the G - 1 sends leave over G - 1 links concurrently; the peers are
    walked starting behind this rank so that no two ranks open with the same destination).  ``async_op``: return the
    pending work handles instead of waiting -- the caller keeps computing and waits before the segments are read.
    ``rank_map`` (tests): logical rank -> rank of the process group, so that a one-GPU box can run the grouped send / recv
    on RCCL as a loop onto itself (``tests/test_gpu_sharded_encode.py``)."""
    import torch.distributed as dist
    dtype = dtype or mine.dtype
    views_by_rank = []
    slabs = []
    for r in range(world):
        if r == rank:
            slab, views = mine, carve_views(mine, sizes_by_rank[r])
        else:
            slab, views = carve(sizes_by_rank[r], dtype, device)
        views_by_rank.append(views)
        slabs.append(slab)
    glob = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    if rank_map is not None:
        inner = glob
        glob = lambda r: inner(rank_map[r])  # noqa: E731
    ops = []
    for d in range(1, world):
        src = (rank - d) % world                     # receive from the rank d places behind, send to the one d ahead
        dst = (rank + d) % world
        if slabs[src].numel() > 0:
            ops.append(dist.P2POp(dist.irecv, slabs[src], glob(src), group))
        if mine.numel() > 0:
            ops.append(dist.P2POp(dist.isend, mine, glob(dst), group))
    handles = list(dist.batch_isend_irecv(ops)) if ops else []
    if not async_op:
        for h in handles:
            h.wait()
        handles = []
    return views_by_rank, handles


def exchange_bytes(sizes_by_rank: Sequence[Sequence[int]], itemsize: int = 2, align: int = 8) -> List[int]:
    """Bytes each rank RECEIVES in ``exchange_slabs`` (every peer's slab at its carved size) -- the planner's input."""
    slab = [sum((n + align - 1) // align * align for n in sizes) * itemsize for sizes in sizes_by_rank]
    total = sum(slab)
    return [total - b for b in slab]


def carve_views(slab: torch.Tensor, sizes: Sequence[int], align: int = 8) -> List[torch.Tensor]:
    """The per-segment views of a slab that ``carve(sizes)`` laid out."""
    out, off = [], 0
    for n in sizes:
        out.append(slab[off:off + n])
        off += (n + align - 1) // align * align
    if off != slab.numel():
        raise ValueError(f"slab of {slab.numel()} elements does not match the plan ({off})")
    return out
