"""Cache engine: module-KV precompute (schema encode), staged-KV gather, request assembly.

Mirrors the reference's ``promptcache/cache_engine.py`` surface -- ``TokenSequenceCache`` (:50-83),
``PromptCache`` (:87-165), ``SchemaCache`` (:168-322), ``CacheEngine`` (:331-522) -- with the data path
re-designed for one MI355X:

  * module KV never leaves HBM: a segment store is one ``[L][2][Hkv][len][D]`` fp16 tensor filled by
    ``pc_kv_slice_store`` straight from the encode pass's arena (the reference slices per layer and
    ``.cpu()``s every piece, :283-296, then re-uploads it from pageable host memory on every prompt, :148-149);
  * ``PromptCache.update`` is ONE ``pc_kv_gather`` launch over a segment table instead of
    ``2 * n_layers * n_segments`` ``copy_`` calls (:135-151);
  * the staged buffer is the KV arena the model appends to in place, so the prefill never re-copies it.

Schema encode shards over the GPUs of a node when ``torch.distributed`` is initialised
(``parallel.py``): every scaffold is an independent forward pass (:217-304).
"""
from __future__ import annotations

import gc
import os
import itertools
import time
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import _native, parallel
from .model import LanguageModel
from .model.kv_arena import KVArena, StagedKV, StagingPlan
from .pml import Module, ModuleRef, Path, Prompt, Schema, TokenSequence, UnionModule  # noqa: F401

KVCache = List[Tuple[torch.Tensor, torch.Tensor]]


def pad_batch(batch_list: List[List[int]], pad_id: int) -> Tuple[List[List[int]], List[List[int]]]:
    """Right-pad to the longest row; returns (padded, 0/1 mask) (reference :38-47)."""
    width = max(map(len, batch_list))
    padded = [row + [pad_id] * (width - len(row)) for row in batch_list]
    mask = [[1] * len(row) + [0] * (width - len(row)) for row in batch_list]
    return padded, mask


class TokenSequenceCache:
    """Module KV of one text segment: one ``[L][2][Hkv][len][D]`` fp16 tensor.

    Two tiers, as in the reference (``host_cache`` / ``device_cache``, ``upload`` / ``free``, :50-83), with the
    defaults turned around for a 288 GB part: the store is born in HBM and stays there; a library that does not fit
    moves segments to PINNED host memory (``offload``), from where ``pc_kv_gather`` reads them in place over PCIe --
    pinned allocations are mapped into the device address space, so a staged prompt may mix HBM and host segments
    in the one gather launch.  ``upload`` brings a segment back, ``free`` drops the HBM copy."""

    def __init__(self, seq: TokenSequence, store: torch.Tensor):
        self.token_sequence = seq
        self._n = len(seq)
        self.usage_counter = 0
        self.device_store: Optional[torch.Tensor] = store if store.is_cuda else None
        self.host_store: Optional[torch.Tensor] = None if store.is_cuda else self._pinned(store)

    @staticmethod
    def _pinned(t: torch.Tensor) -> torch.Tensor:
        if t.is_cuda or not t.is_pinned():
            out = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            out.copy_(t)                      # device -> pinned host: synchronous on the current stream
            return out
        return t

    def inc_usage_counter(self):
        self.usage_counter += 1

    @property
    def store(self) -> torch.Tensor:
        """The copy a gather reads: HBM when resident, else the pinned host tensor."""
        return self.device_store if self.device_store is not None else self.host_store

    def offload(self) -> None:
        """Move the segment to the host tier (keeps an existing host copy; drops the HBM copy)."""
        if self.host_store is None:
            self.host_store = self._pinned(self.device_store)
        self.device_store = None

    def upload(self, device=None) -> None:
        """Reference :65-68: make the segment HBM-resident (no-op when it already is)."""
        if self.device_store is None:
            self.device_store = self.host_store.to(device if device is not None else "cuda", non_blocking=True)

    def free(self) -> None:
        """Reference :70-73: release the HBM copy; the segment stays usable from the host tier."""
        if self.device_store is not None:
            self.offload()

    @staticmethod
    def _views(store: torch.Tensor) -> KVCache:
        return [(store[i, 0], store[i, 1]) for i in range(store.shape[0])]

    @property
    def host_cache(self) -> Optional[KVCache]:
        return None if self.host_store is None else self._views(self.host_store)

    @property
    def device_cache(self) -> Optional[KVCache]:
        return None if self.device_store is None else self._views(self.device_store)

    @property
    def cache(self) -> KVCache:
        """Per-layer ``(K, V)`` views ``[Hkv, len, D]``, device copy first (the reference's ``cache``, :75-80)."""
        return self._views(self.store)

    def __len__(self):
        return self._n


class PromptCache:
    """The staged, contiguous KV buffer (one arena, batch 1) and what is currently staged in it."""

    def __init__(self, max_ctx_length: int, num_layers: int, num_head: int, head_dim: int, target_device):
        self.max_ctx_length = max_ctx_length
        self.num_head = num_head
        self.head_dim = head_dim
        self.arena = KVArena(1, num_layers, num_head, max_ctx_length, head_dim, target_device)
        self.staged: List[TokenSequenceCache] = []
        self.length = 0
        # when set, update() brackets the gather launch with HIP events on the launch stream
        # (kernel-duration measurement for the roofline report; see bench.py)
        self.record_events = False
        self.last_gather_events = None
        self.last_gather_tokens = 0
        # Leave the copy to the first forward over the staged buffer (set by CacheEngine for models whose cached-prefill
        # attention stages while it reads, ``supports_fused_gather``): update() then only records WHAT is staged
        # (``KVArena.pending``); the rows arrive with the first ``lm()`` call -- or with one pc_kv_gather launch as soon as
        # somebody looks at the returned views.  TTFT (cache_time + the first lm() call) moves the K/V once instead of
        # gather read + gather write + attention read.
        self.defer_gather = False

    def reset(self):
        self.staged, self.length = [], 0
        self.arena.tail_base, self.arena.tail_len = -1, 0
        self.arena.pending = None

    @torch.inference_mode()
    def update(self, modules: Sequence[TokenSequenceCache]):
        """Stage exactly ``modules`` (most-used first, stable) with one gather launch.

        Segments already staged at the same place from the previous prompt are kept (the intent of the
        reference's retention logic, :121-129; its comparison of the sorted new list with the unsorted
        previous list is a latent layout bug and is not reproduced -- the staged layout is always the
        concatenation of ``ordered``)."""
        ordered = sorted(modules, key=lambda e: e.usage_counter, reverse=True)
        keep = 0
        if self.arena.pending is not None:       # the previous staging was never carried out: nothing of it can be kept
            self.arena.pending = None
            self.staged = []
        for m, prev in zip(ordered, self.staged):
            if m is prev or m.token_sequence is prev.token_sequence:
                keep += 1
            else:
                break
        offset = sum(len(m) for m in ordered[:keep])
        ptrs, lens, offs = [], [], []
        for m in ordered[keep:]:
            ptrs.append(m.store.data_ptr())
            lens.append(len(m))
            offs.append(offset)
            offset += len(m)
        if offset > self.max_ctx_length:
            raise ValueError(f"prompt modules need {offset} staged tokens but max_ctx_length is {self.max_ctx_length}")
        a = self.arena
        if self.defer_gather and ptrs and all(m.device_store is not None for m in ordered[keep:]):
            # module KV in HBM: the first forward stages (segments in the host tier keep the explicit gather: the copy kernel
            # streams them over PCIe, the attention would fetch them in latency-bound pieces)
            a.pending = StagingPlan(list(zip(ptrs, lens, offs)), offset, [m.device_store for m in ordered[keep:]])
            self.last_gather_events = None
        else:
            if self.record_events:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            _native.kv_gather(ptrs, lens, offs, a.buf, a.L, a.Hkv, a.D, a.cap)
            if self.record_events:
                ev[1].record()
                self.last_gather_events = ev
                self.last_gather_tokens = sum(lens)
        self.staged = list(ordered)
        self.length = offset
        a.length = offset
        a.tail_base, a.tail_len = -1, 0          # a new staging: the residual tail of the previous generation is void

    def __len__(self):
        return self.length

    @property
    def cache(self) -> StagedKV:
        return self.arena.views(self.length).unbatched()


class SchemaCache:
    def __init__(self, schema: Schema, lm: LanguageModel, batch_size: int = 1, target_device=None, no_cache=False,
                 module_memory: str = "device"):
        self.schema = schema
        self.lm = lm
        self.module_memory = module_memory
        self.cache_l1: Dict[int, TokenSequenceCache] = {}
        self.cache_l2: Dict[Tuple[int, int], Tuple[TokenSequenceCache, TokenSequenceCache]] = {}
        self.target_device = lm.device if target_device is None else target_device
        self.encode_stats: Dict[str, float] = {}
        self._pending: list = []        # exchange work handles not yet waited for (async multi-GPU library encode)
        self._jobs = None
        if not no_cache:
            self.encode(batch_size)

    def encode(self, batch_size: int = 1, owner_rank: Optional[int] = None, async_exchange: bool = False,
               shards: Optional[List[List[int]]] = None) -> None:
        """Run the module-KV precompute.  ``shards[r]`` = the passes rank r encodes (a library schedule,
        ``parallel.plan_library`` through ``CacheEngine.add_schemas``); ``owner_rank``: the WHOLE schema on that rank (the
        others only receive); neither: this schema's passes are levelled over all ranks.
        ``async_exchange``: leave the module-KV exchange in flight (``wait_exchange`` before the segments are read)."""
        self._process(batch_size, owner_rank, async_exchange, shards)
        if self.module_memory == "host":
            self.wait_exchange()
            for c in self.cache_l1.values():
                c.offload()
            gc.collect()

    def wait_exchange(self) -> None:
        """Wait for the module-KV exchange left in flight by an asynchronous encode.  ``encode_stats["exchange_exposed_s"]``
        accumulates what the wait cost: host time blocked (gloo) or the time the compute stream stalled behind the collective
        (RCCL: ``work.wait()`` is a stream wait, bracketed by events) -- the part of the exchange later encodes did NOT hide."""
        if not self._pending:
            return
        on_gpu = torch.cuda.is_available() and torch.device(self.target_device).type == "cuda"
        if on_gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        t0 = time.perf_counter()
        for h in self._pending:
            h.wait()
        host = time.perf_counter() - t0
        self._pending = []
        stall = 0.0
        if on_gpu:
            e1.record()
            e1.synchronize()
            stall = e0.elapsed_time(e1) * 1e-3
        self.encode_stats["exchange_exposed_s"] = self.encode_stats.get("exchange_exposed_s", 0.0) + max(host, stall)

    def reexchange_seconds(self) -> float:
        """Measurement only (bench.py --gpus N): run this schema's module-KV exchange AGAIN, alone and blocking -- every rank
        sends the slab it encoded to every peer into scratch slabs -- and return its wall time on this rank.  Collective: every
        rank must call it, in the same order.  0.0 for a schema that was encoded on one rank."""
        ex = getattr(self, "_exchange", None)
        rank, world = parallel.rank_world()
        if ex is None or world <= 1:
            return 0.0
        self.wait_exchange()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        parallel.exchange_slabs(ex["slab"], ex["sizes"], rank, world, ex["slab"].device)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        return time.perf_counter() - t0

    def plan_cost(self) -> int:
        """Tokens this schema's encode runs through the model (suffixes behind the shared trunk prefix counted once, every
        pass cut behind its last owned token: exactly ``encode_stats["computed_tokens"]`` of a one-rank encode)."""
        return sum(self.pass_costs())

    def plan_items(self) -> Tuple[int, List[int], List[bool]]:
        """``(trunk, costs, needs_trunk)`` for ``parallel.plan_library``: rows a rank runs first when it takes a pass that builds on
        the trunk (the root scaffold up to the longest shared prefix; 0 without trunk reuse), the rows each pass then adds, and
        which passes those are -- the root pass itself and every suffix pass; a scaffold encoded in full (prefix 0) runs without
        the trunk, so a rank that holds only such passes is not charged for it."""
        jobs, prefix = self._plan_with_prefix()
        need = self._need(jobs, prefix)
        if any(prefix):
            return need[0], [0] + [need[i] - prefix[i] for i in range(1, len(jobs))], [True] + [p > 0 for p in prefix[1:]]
        return 0, list(need), [False] * len(jobs)

    def plan_forwards(self, mine: Optional[Sequence[int]] = None, batch_size: int = 1) -> List[int]:
        """Rows (padding included) of every forward a rank runs for its share ``mine`` of this schema's passes (default: all),
        in the order ``_process`` runs them: the trunk, the packed whole scaffolds, the packed suffix groups.  Host arithmetic
        on the token layout (``bench.py --plan-only`` prices each forward on a measured rows -> seconds curve)."""
        jobs, prefix = self._plan_with_prefix()
        need = self._need(jobs, prefix)
        mine = list(range(len(jobs))) if mine is None else list(mine)
        shared = [i for i in mine if prefix[i] > 0]
        whole = [i for i in mine if prefix[i] == 0]
        out: List[int] = []
        if shared:
            out.append(need[0])
            if 0 in whole:
                whole.remove(0)
        for idxs in self._pack(whole, need, batch_size):
            out.append(len(idxs) * max(need[i] for i in idxs))
        suffix_len = [need[i] - prefix[i] for i in range(len(jobs))]
        if self._ragged_possible():
            groups = self._pack(shared, suffix_len, batch_size)
        else:
            by_prefix: Dict[int, List[int]] = {}
            for i in shared:
                by_prefix.setdefault(prefix[i], []).append(i)
            groups = [g for _, members in sorted(by_prefix.items()) for g in self._pack(members, suffix_len, batch_size)]
        for idxs in groups:
            out.append(len(idxs) * max(suffix_len[i] for i in idxs))
        return out

    def pass_costs(self) -> List[int]:
        """Rows each pass of the plan runs through the model (job order of ``_plan_with_prefix``)."""
        jobs, prefix = self._plan_with_prefix()
        need = self._need(jobs, prefix)
        return [need[i] - prefix[i] for i in range(len(jobs))]

    def _need(self, jobs, prefix) -> List[int]:
        """Rows a pass has to run: up to its last owned token.  Under the causal mask the K/V of a token depend on nothing
        behind it, so the tail of a scaffold that this pass does not own (the reference runs it and throws it away,
        cache_engine.py:243-296) is cut off -- and the root pass also covers the longest prefix a suffix pass builds on."""
        truncate = self.truncate_scaffolds and self._batch_invariant()      # (LLM.int8: a call's rows choose its outlier columns)
        need = []
        for i, j in enumerate(jobs):
            end = self._owned_end(j)
            if i == 0:
                end = max([end] + list(prefix))
            need.append(end if truncate else len(j["token_ids"]))
        return need

    @staticmethod
    def _owned_end(job) -> int:
        pos = job["position_ids"]
        return max(pos.index(tc.offset) + len(tc) for tc in job["owned"])

    def _plan_with_prefix(self):
        """(jobs, prefix): the plan plus, per job, the number of leading tokens it shares with the root scaffold (job 0)
        when it is encoded as a suffix over the trunk's K/V (0 = encoded in full).  Cached: tokenising every scaffold of a
        large schema is the expensive host step, and a library scheduler asks for the cost before it encodes."""
        if self._jobs is not None:
            return self._jobs
        jobs = self._plan()
        # ---- trunk reuse.  Scaffolds are emitted in position order and differ from the root scaffold (job 0: every
        # union at its default member) from the first token of their own union member on, so under the causal mask
        # the K/V of their common prefix is the root pass's K/V.  Such a scaffold is encoded as "suffix over the
        # trunk's prefix rows" -- the very cached-prefill the engine exists for -- instead of from scratch (the
        # reference re-encodes every scaffold in full, cache_engine.py:221-248).
        prefix = [0] * len(jobs)
        if self.share_trunk and len(jobs) > 1 and self._batch_invariant():
            t_ids, t_pos = jobs[0]["token_ids"], jobs[0]["position_ids"]
            for i in range(1, len(jobs)):
                ids, pos = jobs[i]["token_ids"], jobs[i]["position_ids"]
                n = 0
                lim = min(len(ids), len(t_ids)) - 1          # at least one token is always run
                while n < lim and ids[n] == t_ids[n] and pos[n] == t_pos[n]:
                    n += 1
                # A pass is cut behind its last owned token (``_need``), and at least one row must run through the model:
                # a union member whose tokens equal, or are a prefix of, the default member's (duplicate documents, members
                # cut to the same ``max_tokens``) shares MORE than it has to run -- the prefix stops one row short of the cut.
                if self.truncate_scaffolds:
                    n = min(n, self._owned_end(jobs[i]) - 1)
                # never cut an owned segment in two: the prefix ends where a segment it runs into begins (its rows then
                # come out of ONE arena when the suffix pass reads the trunk in place, see _process).  Only where that path
                # exists: a model whose suffix batches carry a copy of the trunk keeps the longer prefix (less recompute).
                if self._in_place_possible():
                    for tc in jobs[i]["owned"]:
                        s0 = pos.index(tc.offset)
                        if s0 < n < s0 + len(tc):
                            n = s0
                if n >= self.share_trunk_min and n >= len(ids) // 5:
                    prefix[i] = n
        self._jobs = (jobs, prefix)
        return self._jobs

    def _ragged_possible(self) -> bool:
        hf = getattr(self.lm, "hf_model", None)
        return bool(getattr(hf, "supports_ragged_past", False)) and not bool(getattr(self.lm, "use_full_position_ids", False)) and \
            self.ragged_suffix_batches

    def _in_place_possible(self) -> bool:
        """Can the suffix batches of this model read the trunk's rows in place (pc_attn's shared prefix)?"""
        return self._ragged_possible() and self.shared_prefix_in_place and \
            bool(getattr(getattr(self.lm, "hf_model", None), "supports_shared_prefix", False))

    # ------------------------------------------------------------------------------------------
    def _plan(self):
        """Scaffolds to encode and, per scaffold, the segments whose KV it finally owns.

        The reference encodes every path in order and lets later passes overwrite ``cache_l1`` entries
        (:296), so a segment keeps the KV of the LAST scaffold containing it.  Resolving that up front
        gives each segment exactly one owner pass (passes that own nothing are skipped) -- which is also
        what makes the passes shardable across GPUs."""
        paths = self.schema.encode_paths()
        jobs = []
        owner: Dict[int, int] = {}
        for k, path in enumerate(paths):
            scaffold = self.schema.get_scaffold(path)
            token_ids, position_ids = scaffold.token_ids(), scaffold.position_ids()
            targets = scaffold.select(path).all_token_sequences()
            for tc in targets:
                owner[id(tc)] = k
            jobs.append(dict(path=path, token_ids=token_ids, position_ids=position_ids, targets=targets))
        for k, job in enumerate(jobs):
            job["owned"] = [tc for tc in job["targets"] if owner[id(tc)] == k and len(tc) > 0]
        return [j for j in jobs if j["owned"]]

    @torch.inference_mode()
    def _process(self, batch_size: int = 1, owner_rank: Optional[int] = None, async_exchange: bool = False,
                 shards: Optional[List[List[int]]] = None):
        lm = self.lm
        L, Hkv, D = lm.get_cache_shape()
        dev = lm.device
        jobs, prefix = self._plan_with_prefix()
        rank, world = parallel.rank_world()
        if shards is not None:
            if len(shards) != world or sorted(i for sh in shards for i in sh) != list(range(len(jobs))):
                raise ValueError("shards must partition the schema's passes over the ranks")
            shards = [sorted(sh) for sh in shards]
        elif owner_rank is not None:
            # schema-level sharding: one rank encodes every pass (and the trunk exactly once), the others receive
            shards = [list(range(len(jobs))) if r == owner_rank else [] for r in range(world)]
        else:
            # level the passes over the ranks by what each really costs, the trunk a taker re-runs included (world == 1 ->
            # everything on this rank)
            shards = parallel.plan_library([self.plan_items()], world)[0][0]
        mine = shards[rank]
        # every rank's segments live back to back in ONE slab per rank (ascending job order, then plan order inside a
        # job): the encode writes its stores through views of the slab, and the exchange moves whole slabs in place
        seg_numel = lambda tc: L * 2 * Hkv * len(tc) * D                                        # noqa: E731
        sizes_by_rank = [[seg_numel(tc) for i in idxs for tc in jobs[i]["owned"]] for idxs in shards]
        my_slab, my_views = parallel.carve(sizes_by_rank[rank], torch.float16, dev)
        view_of: Dict[Tuple[int, int], torch.Tensor] = {}
        it = iter(my_views)
        for i in mine:
            for k in range(len(jobs[i]["owned"])):
                view_of[(i, k)] = next(it)

        need = self._need(jobs, prefix)          # rows each pass runs: up to its last owned token
        encoded_tokens = computed_tokens = 0
        per_job: Dict[int, List[Tuple[TokenSequence, torch.Tensor]]] = {}
        full_pos = bool(getattr(lm, "use_full_position_ids", False))
        custom_store_hooks = (type(lm).store_k_hook is not LanguageModel.store_k_hook or
                              type(lm).store_v_hook is not LanguageModel.store_v_hook)

        def store_owned(job_idx: int, arena: KVArena, row: int, shared: Optional[Tuple[KVArena, int]] = None):
            """``shared = (trunk arena, n_pre)``: the pass read its first n_pre keys out of the trunk arena in place, and
            ``arena`` holds its rows from n_pre on (a segment lies on one side: _plan_with_prefix ends the prefix there)."""
            job = jobs[job_idx]
            pos, owned = job["position_ids"], job["owned"]
            # position ids of a scaffold may be interleaved, but each segment is contiguous (:275-279)
            src_off = [pos.index(tc.offset) for tc in owned]
            lens = [len(tc) for tc in owned]
            stores = [view_of[(job_idx, k)].view(L, 2, Hkv, n, D) for k, n in enumerate(lens)]
            if shared is None:
                _native.kv_slice_store(arena.buf[row], arena.cap, src_off, lens, [s.data_ptr() for s in stores], L, Hkv, D)
            else:
                trunk, n_pre = shared
                own = [k for k in range(len(owned)) if src_off[k] >= n_pre]
                pre_ = [k for k in range(len(owned)) if src_off[k] < n_pre]
                assert all(src_off[k] + lens[k] <= n_pre for k in pre_), "an owned segment straddles the shared prefix"
                if own:
                    _native.kv_slice_store(arena.buf[row], arena.cap, [src_off[k] - n_pre for k in own], [lens[k] for k in own],
                                           [stores[k].data_ptr() for k in own], L, Hkv, D)
                if pre_:
                    _native.kv_slice_store(trunk.buf[0], trunk.cap, [src_off[k] for k in pre_], [lens[k] for k in pre_],
                                           [stores[k].data_ptr() for k in pre_], L, Hkv, D)
            if custom_store_hooks:
                # an adapter that overrides store_k_hook / store_v_hook (reference :284-285) sees the per-layer
                # [Hkv, len, D] slices exactly as the reference hands them over; identity hooks cost nothing
                for st in stores:
                    for li in range(L):
                        st[li, 0].copy_(lm.store_k_hook(st[li, 0]))
                        st[li, 1].copy_(lm.store_v_hook(st[li, 1]))
            per_job[job_idx] = list(zip(owned, stores))

        shared = [i for i in mine if prefix[i] > 0]
        whole = [i for i in mine if prefix[i] == 0]
        trunk_arena: Optional[KVArena] = None
        if shared:
            job = jobs[0]
            out = lm(input_ids=torch.tensor([job["token_ids"][:need[0]]], device=dev, dtype=torch.long),
                     position_ids=torch.tensor([job["position_ids"][:need[0]]], device=dev, dtype=torch.long), use_cache=True,
                     many_rows=True, kv_only=True)
            trunk_arena = out.past_key_values.arena
            computed_tokens += need[0]
            if 0 in whole:                                   # this rank also owns the root pass: store from the same run
                whole.remove(0)
                encoded_tokens += len(job["token_ids"])
                store_owned(0, trunk_arena, 0)
            del out

        for idxs in self._pack(whole, need, batch_size):
            ids_pad, mask = pad_batch([jobs[i]["token_ids"][:need[i]] for i in idxs], lm.eos_token_id)
            pos_pad, _ = pad_batch([jobs[i]["position_ids"][:need[i]] for i in idxs], 0)
            out = lm(input_ids=torch.tensor(ids_pad, device=dev, dtype=torch.long),
                     position_ids=torch.tensor(pos_pad, device=dev, dtype=torch.long),
                     attention_mask=torch.tensor(mask, dtype=torch.float16),     # (host: the HIP path only checks it is right-padded)
                     use_cache=True, many_rows=True, kv_only=True)
            arena: KVArena = out.past_key_values.arena
            for row, i in enumerate(idxs):
                encoded_tokens += len(jobs[i]["token_ids"])
                computed_tokens += need[i]
                store_owned(i, arena, row)
            del out, arena

        # Suffix passes.  Where the model takes one past length per batch row (``supports_ragged_past``) ALL of them are
        # packed together, longest suffix first, whatever union they belong to: a batch row holds its own trunk prefix
        # [0, n_pre_i) and appends its suffix behind it, so one forward carries thousands of rows and the projections
        # run in the MFMA-bound regime (members of a single union alone are a few hundred rows: ~65 % of that rate).
        # Otherwise (ALiBi: one position row per batch row) scaffolds sharing the same prefix length -- the members of
        # one union -- go through one batched forward.
        ragged = self._ragged_possible()
        suffix_len = [need[i] - prefix[i] for i in range(len(jobs))]
        row_bytes = L * 2 * Hkv * D * 2 * 2                  # K and V, fp16, + the residual planes of the encode arenas
        # Where the attention kernel takes a shared key prefix (``supports_shared_prefix``: the split-precision many-row
        # path) the suffix batch reads the trunk's rows IN PLACE, out of the trunk arena: its own arena holds the suffix
        # rows only.  Otherwise every batch row carries a copy of its prefix (11 GB of device copies per persona encode).
        in_place = self._in_place_possible() and trunk_arena is not None and trunk_arena.lo is not None
        if self._in_place_possible() and trunk_arena is not None and not in_place:
            # (planned for in-place reading, served by copies: fine -- a prefix that ends on a segment boundary is valid for both)
            pass
        if in_place:
            groups = [(None, idxs) for idxs in self._pack(shared, suffix_len, batch_size)]
        elif ragged:
            groups = [(None, idxs) for idxs in self._pack(shared, suffix_len, batch_size, prefix, row_bytes)]
        else:
            by_prefix: Dict[int, List[int]] = {}
            for i in shared:
                by_prefix.setdefault(prefix[i], []).append(i)
            groups = [(n_pre, idxs) for n_pre, members in sorted(by_prefix.items())
                      for idxs in self._pack(members, suffix_len, batch_size, prefix, row_bytes)]
        for n_same, idxs in groups:
            group = [jobs[i] for i in idxs]
            pre = [prefix[i] for i in idxs]
            n_max = max(pre)
            width = max(suffix_len[i] for i in idxs)
            if in_place:
                ids_pad, mask = pad_batch([jobs[i]["token_ids"][prefix[i]:need[i]] for i in idxs], lm.eos_token_id)
                pos_pad, _ = pad_batch([jobs[i]["position_ids"][prefix[i]:need[i]] for i in idxs], 0)
                out = lm(input_ids=torch.tensor(ids_pad, device=dev, dtype=torch.long),
                         position_ids=torch.tensor(pos_pad, device=dev, dtype=torch.long),
                         attention_mask=torch.tensor(mask, dtype=torch.float16),     # (host: the HIP path only checks it is right-padded)
                         use_cache=True, many_rows=True, kv_only=True, shared_prefix=(trunk_arena, pre))
                arena = out.past_key_values.arena
                for row, i in enumerate(idxs):
                    encoded_tokens += len(jobs[i]["token_ids"])
                    computed_tokens += suffix_len[i]
                    store_owned(i, arena, row, (trunk_arena, prefix[i]))
                del out, arena
                continue
            arena = KVArena(len(group), L, Hkv, n_max + width, D, dev)
            has_lo = trunk_arena.lo is not None and trunk_arena.lo_len >= n_max     # the trunk's keys stay split-precision
            if has_lo:
                arena.with_lo()
            if n_same is not None:
                arena.buf[:, :, :, :, :n_max].copy_(trunk_arena.buf[:, :, :, :, :n_max].expand(len(group), -1, -1, -1, -1, -1))
                if has_lo:
                    arena.lo[:, :, :, :, :n_max].copy_(trunk_arena.lo[:, :, :, :, :n_max].expand(len(group), -1, -1, -1, -1, -1))
            else:
                for row, n_pre in enumerate(pre):
                    arena.buf[row, :, :, :, :n_pre].copy_(trunk_arena.buf[0, :, :, :, :n_pre])
                    if has_lo:
                        arena.lo[row, :, :, :, :n_pre].copy_(trunk_arena.lo[0, :, :, :, :n_pre])
            if has_lo:
                arena.lo_len = n_max
            arena.length = n_max
            ids_pad, mask = pad_batch([jobs[i]["token_ids"][prefix[i]:need[i]] for i in idxs], lm.eos_token_id)
            pos_pad, _ = pad_batch([jobs[i]["position_ids"][prefix[i]:need[i]] for i in idxs], 0)
            if full_pos:                                 # ALiBi models take the position id of every key
                pos_pad = [list(jobs[0]["position_ids"][:n_max]) + row for row in pos_pad]
            extra = {} if n_same is not None else {"past_lens": torch.tensor(pre, device=dev, dtype=torch.int32)}
            out = lm(input_ids=torch.tensor(ids_pad, device=dev, dtype=torch.long),
                     position_ids=torch.tensor(pos_pad, device=dev, dtype=torch.long),
                     attention_mask=torch.tensor(mask, dtype=torch.float16),     # (host: the HIP path only checks it is right-padded)
                     past_key_values=arena.views(), use_cache=True, many_rows=True, kv_only=True, **extra)
            arena = out.past_key_values.arena
            for row, i in enumerate(idxs):
                encoded_tokens += len(jobs[i]["token_ids"])
                computed_tokens += suffix_len[i]
                store_owned(i, arena, row)
            del out, arena
        del trunk_arena
        if world > 1:
            # one exchange step: every GPU ends with the whole module library (slabs travel as they are: no padding to
            # the largest shard, no pack copy; see parallel.exchange_slabs)
            t_x = time.perf_counter()
            views_by_rank, handles = parallel.exchange_slabs(my_slab, sizes_by_rank, rank, world, dev, async_op=async_exchange)
            if not async_exchange and torch.device(dev).type == "cuda":
                torch.cuda.synchronize()
            blocking_s = 0.0 if async_exchange else time.perf_counter() - t_x      # a blocking exchange is exposed in full
            self._pending.extend(handles)
            # bytes_rx: the planner's figure (parallel.exchange_bytes, from the token layout); bytes_rx_buffers: what the receive
            # slabs this rank really allocated and posted irecvs for hold (tests assert the two agree)
            self._exchange = dict(slab=my_slab, sizes=sizes_by_rank, bytes_rx=parallel.exchange_bytes(sizes_by_rank)[rank],
                                  bytes_rx_buffers=sum(v.numel() * v.element_size() for r, vs in enumerate(views_by_rank)
                                                       if r != rank for v in vs))
            for r, idxs in enumerate(shards):
                vit = iter(views_by_rank[r])
                for i in idxs:
                    for tc in jobs[i]["owned"]:
                        self.cache_l1[id(tc)] = TokenSequenceCache(tc, next(vit).view(L, 2, Hkv, len(tc), D))
        else:
            for i in sorted(per_job):
                for tc, store in per_job[i]:
                    self.cache_l1[id(tc)] = TokenSequenceCache(tc, store)
        self.encode_stats = dict(passes=len(mine), total_passes=len(jobs), encoded_tokens=encoded_tokens,
                                 computed_tokens=computed_tokens, trunk_shared_passes=len(shared), owner_rank=owner_rank,
                                 cached_tokens=sum(len(c) for c in self.cache_l1.values()),
                                 owned_cached_tokens=sum(len(tc) for i in mine for tc in jobs[i]["owned"]),   # ... this rank encoded
                                 exchange_bytes_rx=(self._exchange["bytes_rx"] if world > 1 else 0),
                                 exchange_bytes_rx_buffers=(self._exchange["bytes_rx_buffers"] if world > 1 else 0),
                                 exchange_exposed_s=(blocking_s if world > 1 else 0.0))
        if os.environ.get("PC_ENCODE_GC", "1") != "0":          # (dev knob: what the collection costs the encode's wall time)
            gc.collect()

    # tokens (padding included) one encode forward may carry when scaffolds are packed into a batch
    encode_token_budget = 8192
    encode_rows_max = 32                    # passes per forward (each holds its own arena row: prefix + suffix K/V)
    encode_forward_cost = int(os.environ.get("PC_ENC_FWD_COST", "150"))   # fixed cost of one more forward, in token rows (_pack)
    # run a scaffold only up to its last owned token (exact under the causal mask; see _process)
    truncate_scaffolds = os.environ.get("PC_TRUNCATE_SCAFFOLDS", "1") != "0"
    # encode scaffolds as suffixes over the root scaffold's K/V where they share a prefix with it (see _process)
    share_trunk = os.environ.get("PC_SHARE_TRUNK", "1") != "0"
    share_trunk_min = 32
    # pack suffix passes of different unions into one batch (per-row past lengths; models with supports_ragged_past)
    ragged_suffix_batches = os.environ.get("PC_RAGGED_SUFFIX", "1") != "0"
    # suffix batches read the trunk's K/V in place instead of holding a copy per batch row (models with supports_shared_prefix)
    shared_prefix_in_place = os.environ.get("PC_PREFIX_IN_PLACE", "1") != "0"

    def _batch_invariant(self) -> bool:
        """False when a row's result depends on which other rows travel in the same forward -- LLM.int8 picks its fp16
        outlier COLUMNS over all rows of a call (model/llama_hip.py ``llm_int8``).  The reference encodes one whole scaffold
        per call (``batch_size`` 1, cache_engine.py:232-248), so such a model gets exactly that: no trunk reuse, no packing."""
        return bool(getattr(getattr(self.lm, "hf_model", None), "batch_invariant", True))

    # bytes one encode forward's group arena (rows x (trunk prefix + suffix width) x K/V + residual planes) may take
    encode_arena_bytes = int(float(os.environ.get("PC_ENC_ARENA_GB", "48")) * 2 ** 30)

    def _pack(self, mine: List[int], lengths: List[int], batch_size: int, prefix: Optional[List[int]] = None,
              bytes_per_row_token: int = 0) -> List[List[int]]:
        """Group this rank's scaffold passes into right-padded batches.  ``batch_size`` is the reference's knob
        (``cache_engine.py:232``: consecutive passes, in order); with the default of 1 the engine packs on its own:
        longest first, as many passes per forward as fit ``encode_token_budget`` rows -- dense GEMMs at M ~ 600
        rows run at ~75 % of their M ~ 5000 rate on MI355X, and padded rows never influence real ones."""
        if batch_size > 1:
            return [mine[i:i + batch_size] for i in range(0, len(mine), batch_size)]
        if not self._batch_invariant():
            return [[i] for i in mine]
        # Longest first, cut into consecutive groups so that the padded rows (rows x width of the group's longest pass)
        # plus a fixed cost per forward are minimal: passes of similar length travel together.  (Greedy filling to the
        # budget padded the persona schema's 149..282-token suffixes to 282 and 258: 17 % of the rows were padding; the
        # optimal cut pads 6 %.)  The many-row GEMM runs at its full rate from ~500 rows up (profiles/r02_dense_splitk.txt),
        # so smaller groups cost only the per-forward launches: ~2 ms, ~150 rows' worth.
        order = sorted(mine, key=lambda i: (-lengths[i], i))
        n = len(order)
        if n == 0:
            return []
        INF = float("inf")
        best = [INF] * (n + 1)
        cut = [0] * (n + 1)
        best[0] = 0
        for j in range(1, n + 1):
            for i in range(max(0, j - self.encode_rows_max), j):
                rows = (j - i) * lengths[order[i]]
                if rows > self.encode_token_budget and j - i > 1:
                    continue
                if prefix is not None and bytes_per_row_token and j - i > 1:
                    # every row of the group's arena also holds its trunk prefix (sized to the longest one of the group)
                    n_max = max(prefix[order[k]] for k in range(i, j))
                    if (j - i) * (n_max + lengths[order[i]]) * bytes_per_row_token > self.encode_arena_bytes:
                        continue
                c = best[i] + rows + self.encode_forward_cost
                if c < best[j]:
                    best[j], cut[j] = c, i
        groups, j = [], n
        while j > 0:
            groups.append(order[cut[j]:j])
            j = cut[j]
        return groups[::-1]

    def get_cache_l1(self, seq: TokenSequence) -> Optional[TokenSequenceCache]:
        return self.cache_l1.get(id(seq))

    def get_cache_l2(self, seq1: TokenSequence, seq2: TokenSequence):
        key = (max(id(seq1), id(seq2)), min(id(seq1), id(seq2)))
        return self.cache_l2.get(key)


class CacheEngine:
    def __init__(self, max_ctx_length: int, lm: LanguageModel, target_device=None, module_memory: Optional[str] = None):
        """``module_memory``: where add_schema leaves the module KV -- ``"device"`` (HBM, default) or ``"host"`` (pinned
        host memory, gathered over PCIe: the reference's default placement, :283-296; for libraries beyond HBM).
        Individual segments move with ``TokenSequenceCache.upload`` / ``free``.  Env default: PC_MODULE_MEMORY."""
        _native.load()
        self.module_memory = module_memory or os.environ.get("PC_MODULE_MEMORY", "device")
        if self.module_memory not in ("device", "host"):
            raise ValueError(f"module_memory must be 'device' or 'host', not {self.module_memory!r}")
        self.lm = lm
        self.schemas: Dict[str, SchemaCache] = {}
        self.target_device = lm.device if target_device is None else target_device
        num_layers, num_head, head_dim = lm.get_cache_shape()
        self.prompt_cache = PromptCache(max_ctx_length=max_ctx_length, num_layers=num_layers, num_head=num_head,
                                        head_dim=head_dim, target_device=self.target_device)
        # PC_DEFER_GATHER=0: PromptCache.update copies at once, as rounds 1-3 did
        self.prompt_cache.defer_gather = bool(getattr(getattr(lm, "hf_model", None), "supports_fused_gather", False)) and \
            os.environ.get("PC_DEFER_GATHER", "1") != "0"

    def add_schema(self, schema: Union[str, Schema], batch_size: int = 1, max_tokens: Optional[int] = None,
                   no_cache: bool = False):
        if isinstance(schema, str):
            schema = Schema(schema, self.lm, max_tokens=max_tokens)
        if schema.name in self.schemas:
            raise ValueError(f"There is already a schema named {schema.name} in the cache")
        self.schemas[schema.name] = SchemaCache(schema, self.lm, batch_size, target_device=self.target_device,
                                                no_cache=no_cache, module_memory=self.module_memory)

    def add_schemas(self, schemas: Sequence[Union[str, Schema]], batch_size: int = 1, max_tokens: Optional[int] = None) -> None:
        """Encode a whole module LIBRARY (the reference loops ``add_schema`` over its schema files, eval.py:172-181).
        One GPU: the same loop.  Several GPUs (``torch.distributed`` initialised):
        whole schemas are dealt to the ranks by longest-processing-time-first on the rows each encode really runs (every trunk
        computed once, on the rank that needs it), then the residual imbalance -- 8 schemas on 8 ranks would leave the rank
        with the largest schema 2.6x the average -- is levelled at PASS granularity: suffix passes of the most loaded ranks'
        schemas move to the ranks below the water line, which re-run that schema's trunk (``library_schedule``).  The
        exchange of a schema is left in flight while the next ones are encoded (RCCL runs on its own stream), and all of
        them are waited for at the end."""
        parsed = [Schema(sc, self.lm, max_tokens=max_tokens) if isinstance(sc, str) else sc for sc in schemas]
        names = [sc.name for sc in parsed]
        for nm in names:
            if nm in self.schemas or names.count(nm) > 1:
                raise ValueError(f"There is already a schema named {nm} in the cache")
        rank, world = parallel.rank_world()
        caches = [SchemaCache(sc, self.lm, batch_size, target_device=self.target_device, no_cache=True,
                              module_memory=self.module_memory) for sc in parsed]
        order, shards = self.library_schedule(caches, world)
        for k in order:
            caches[k].encode(batch_size, shards=shards[k] if world > 1 else None, async_exchange=world > 1)
        for c in caches:
            self.schemas[c.schema.name] = c
        for c in caches:
            c.wait_exchange()

    @staticmethod
    def library_schedule(caches: Sequence["SchemaCache"], world: int):
        """``(order, shards)`` of a library encode over ``world`` ranks: ``shards[k][r]`` = the passes of schema k rank r encodes
        (``parallel.plan_library``: whole schemas by LPT, the residual imbalance levelled at pass granularity, a rank that takes
        passes of a schema re-running its trunk), ``order`` = the sequence every rank walks the schemas in -- schemas with one
        encoder first (their exchanges then overlap the later encodes), the shared ones last.  Host arithmetic on the token
        layout only: the same on every rank, no GPU needed (``bench.py --plan-only``)."""
        if world <= 1:
            return list(range(len(caches))), [None] * len(caches)
        shards, _ = parallel.plan_library([c.plan_items() for c in caches], world)
        members = [sum(1 for sh in shards[k] if sh) for k in range(len(caches))]
        order = sorted(range(len(caches)), key=lambda k: (members[k] > 1, k))
        return order, shards

    def get_schema(self, name: str) -> Optional[Schema]:
        return self.schemas[name].schema if name in self.schemas else None

    def remove_schema(self, name: str):
        if name not in self.schemas:
            raise ValueError(f"There is no such schema named {name} in the cache")
        del self.schemas[name]
        self.prompt_cache.reset()
        # The module stores go back to torch's caching allocator.  The reference also calls
        # torch.cuda.empty_cache() here (:376-378); on ROCm that hands multi-GB blocks back to the driver and the
        # next add_schema pays hipMalloc for them again (measured: 0.27 s -> 0.87 s per persona encode).
        gc.collect()

    def remove_all_schemas(self):
        self.schemas = {}
        self.prompt_cache.reset()
        gc.collect()

    def process(self, prompt: Prompt, no_cache: bool = False, return_full_position_ids: bool = False
                ) -> Tuple[List[int], List[int], float, Optional[KVCache]]:
        """Prompt -> (new token ids, their position ids, gather time in ms, staged KV | None).

        Request assembly follows the reference step by step (:411-474): explicit-stack DFS over
        (module reference, schema module) pairs, arguments fill the first positions of their
        parameter, trailing text continues after the schema."""
        # cache_time: the reference brackets request assembly + PromptCache.update with device events (:391-394, :507-509).
        # When the staging is left to the first forward, or nothing is staged at all (no_cache: pure host assembly), there
        # is no device work in here to bracket: host wall-clock then.
        host_timed = self.prompt_cache.defer_gather or no_cache
        t_host = time.perf_counter()
        if not host_timed:
            start = torch.cuda.Event(enable_timing=True)
            end = torch.cuda.Event(enable_timing=True)
            start.record()

        if prompt.schema not in self.schemas:
            raise ValueError(f"There is no such layout named {prompt.schema} in the cache")
        cached = self.schemas[prompt.schema]
        cached.wait_exchange()
        schema = cached.schema

        used: List[TokenSequence] = []
        arg_ids: List[List[int]] = []
        arg_pos: List[List[int]] = []
        stack: List[Tuple[ModuleRef, Module]] = [(prompt, schema)]
        while stack:
            ref, module = stack.pop()
            used.extend(module.token_sequences())
            params = module.parameters()
            for arg in ref.args:
                param = next((p for p in params if p.name == arg.name), None)
                if param is None:
                    raise ValueError(f"There is no such parameter named {arg.name} in the module {module.name}")
                ids = self.lm.encode(arg.value)
                if len(ids) > param.length:
                    raise ValueError(
                        f"The argument {arg.name} is too long. It should be at most {param.length} characters long")
                arg_ids.append(ids)
                arg_pos.append(param.position_ids()[:len(ids)])
            for m in ref.modules:
                sub = module.select(m.name)
                if sub is None:
                    raise ValueError(f"There is no such module named @{m.name} in the module @{module.name}")
                stack.append((m, sub))

        if len(prompt.text) > 0:
            ids = self.lm.encode(prompt.text)
            arg_ids.append(ids)
            arg_pos.append(list(range(len(schema), len(schema) + len(ids))))

        input_ids = list(itertools.chain(*arg_ids))
        position_ids = list(itertools.chain(*arg_pos))

        if no_cache:
            # everything re-ordered by position, positions re-packed to range(N) (:476-493)
            pairs = sorted(zip([p for s in used for p in s.position_ids()] + position_ids,
                               [t for s in used for t in s.token_ids()] + input_ids))
            ids_sorted = tuple(t for _, t in pairs)
            return ids_sorted, list(range(len(pairs))), (time.perf_counter() - t_host) * 1e3, None

        seq_caches = []
        for s in used:
            if len(s) == 0:
                continue
            sc = cached.get_cache_l1(s)
            if sc is None:
                raise ValueError(f"segment {s!r} of schema {schema.name} has no cached KV")
            sc.inc_usage_counter()
            seq_caches.append(sc)
        self.prompt_cache.update(seq_caches)
        cache = self.prompt_cache.cache
        if host_timed:
            if self.prompt_cache.arena.pending is None:      # update() launched the copy after all (host-tier segments)
                torch.cuda.synchronize()
            cache_time = (time.perf_counter() - t_host) * 1e3
        else:
            end.record()
            torch.cuda.synchronize()
            cache_time = start.elapsed_time(end)
        if type(self.lm).read_k_hook is not LanguageModel.read_k_hook or type(self.lm).read_v_hook is not LanguageModel.read_v_hook:
            for i in range(len(cache)):          # (looking at the views carries out a deferred staging first; identity hooks do not)
                cache[i] = (self.lm.read_k_hook(cache[i][0]), self.lm.read_v_hook(cache[i][1]))
        if return_full_position_ids:
            # positions of the cached keys in the order they are STAGED (most-used first, PromptCache.update), not in the
            # DFS order of `used`: an ALiBi model reads position_ids[:S] as the positions of arena rows [0, S).  (The
            # reference returns the DFS order at :517-519 while staging the sorted order -- the two diverge as soon as
            # the usage counters do; not reproduced.)
            position_ids = [p for c in self.prompt_cache.staged for p in c.token_sequence.position_ids()] + position_ids
        return input_ids, position_ids, cache_time, cache
