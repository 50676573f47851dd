"""KV arena: the HBM layout shared by the staged prompt cache, the encode pass and decode.

Layout (fp16):  ``[B][n_layers][2][n_kv_heads][cap][head_dim]`` -- for B = 1 this is exactly the staged
buffer of the C-ABI (``pc_kv_gather``'s ``dst``), i.e. the reference's per-layer
``torch.empty(num_head, max_ctx_length, head_dim, dtype=torch.half)`` pairs
(``promptcache/cache_engine.py:104-107``) fused into one allocation.  Layer ``i``'s K plane with a
``[:, :length, :]`` slice is what the reference hands to the model as ``past_key_values[i][0]``
(``cache_engine.py:161-165``).

New tokens are appended IN PLACE at rows ``[length, length+q)`` (the reference re-``cat``s the whole
past in every layer of every call, ``promptcache/model/llama2.py:361-364``).
"""
from __future__ import annotations

import weakref
from collections.abc import Sequence
from typing import List, Optional, Tuple

import torch

# live arenas by the address of their buffer: a caller that hands back plain views (the reference's GenerationEngine
# rebuilds the list with k.unsqueeze(0), generation_engine.py:101-102) gets the SAME arena object again -- with its
# residual tail and the hipGraphs keyed on it -- instead of a fresh reconstruction per call
_LIVE: "weakref.WeakValueDictionary[int, KVArena]" = weakref.WeakValueDictionary()


class StagingPlan:
    """What ``PromptCache.update`` decided to stage and has not copied yet: ``(module store address, rows, first staged row)``
    per segment in staging order.  The first forward over the arena either consumes it inside its attention launches (the
    <= 16-row cached prefill reads every staged row from its module store and writes it to the arena as it goes: the K/V
    crosses the chip once) or the arena materialises it with one ``pc_kv_gather`` -- whichever comes first."""

    __slots__ = ("segs", "total", "keep", "_arr")

    def __init__(self, segs: List[Tuple[int, int, int]], total: int, keep: list):
        self.segs = segs          # [(src ptr, len, dst row)]
        self.total = total        # staged rows once the plan is carried out (= the arena's length)
        self.keep = keep          # the stores behind the pointers, kept alive until the plan is carried out
        self._arr = None

    def seg_array(self, dtype):
        """The segments as a numpy array of ``pc_kv_seg`` records (src, dst_row, len)."""
        if self._arr is None:
            import numpy as np
            self._arr = np.array([(p, off, ln) for p, ln, off in self.segs], dtype=dtype)
        return self._arr


class KVArena:
    def __init__(self, batch: int, n_layers: int, n_kv_heads: int, cap: int, head_dim: int, device,
                 dtype=torch.float16):
        self.B, self.L, self.Hkv, self.cap, self.D = batch, n_layers, n_kv_heads, cap, head_dim
        self.buf = torch.empty((batch, n_layers, 2, n_kv_heads, cap, head_dim), device=device, dtype=dtype)
        _LIVE[self.buf.data_ptr()] = self
        self.length = 0
        # optional second buffer of the same shape: the fp16 residuals (value - fp16(value)) of the rows a schema-encode
        # pass appended, so that later rows of that pass -- and scaffolds encoded as suffixes over this pass's prefix --
        # see those keys / values in split precision.  Never stored, never staged: the module KV is `buf`.
        self.lo: Optional[torch.Tensor] = None
        self.lo_len = 0               # rows [0, lo_len) of `lo` hold valid residuals
        # residual TAIL of a generation: fp16 residuals of the rows appended since the staged cache ended -- the prompt's
        # own tokens and every decoded token -- [B][L][2][Hkv][tail_cap][D], row r = key index tail_base + r.  Valid
        # while tail_base + tail_len == length.  The prefill writes rows [0, q), each decode step appends one.
        self.tail_lo: Optional[torch.Tensor] = None
        self.tail_base = -1
        self.tail_len = 0
        self.pending: Optional[StagingPlan] = None     # a staging that has not been carried out yet (see StagingPlan)
        self.row_tab: Optional[torch.Tensor] = None    # pc_kv_row[cap]: the plan expanded per staged row (device scratch)

    def materialize(self) -> None:
        """Carry out a pending staging with one ``pc_kv_gather`` launch (no-op without one)."""
        plan, self.pending = self.pending, None
        if plan is not None and plan.segs:
            from .. import _native
            _native.kv_gather([s[0] for s in plan.segs], [s[1] for s in plan.segs], [s[2] for s in plan.segs], self.buf,
                              self.L, self.Hkv, self.D, self.cap)

    def row_table(self) -> torch.Tensor:
        if self.row_tab is None:
            self.row_tab = torch.empty(self.cap * 16, dtype=torch.uint8, device=self.buf.device)
        return self.row_tab

    def with_lo(self) -> "KVArena":
        if self.lo is None:
            self.lo = torch.empty_like(self.buf)
        return self

    def ensure_tail(self, rows: int) -> None:
        """Room for ``rows`` residual rows.  Capacities come in buckets (320, 512, then powers of two): the captured
        forwards hold the buffer's address, so a slightly longer prompt must not move it."""
        if self.tail_lo is None or self.tail_lo.shape[4] < rows:
            cap = 320 if rows <= 320 else 512
            while cap < rows:
                cap *= 2
            rows = cap
            self.tail_lo = torch.empty((self.B, self.L, 2, self.Hkv, rows, self.D), device=self.buf.device, dtype=self.buf.dtype)
            self.tail_base, self.tail_len = -1, 0

    def tail_planes(self, layer: int):
        """(k_lo, v_lo, batch_stride, head_stride) of the residual tail for one layer."""
        t = self.tail_lo
        cap = t.shape[4]
        return (t[:, layer, 0], t[:, layer, 1], self.L * 2 * self.Hkv * cap * self.D, cap * self.D)

    def lo_planes(self, layer: int):
        """(k_lo, v_lo, batch_stride, head_stride, row0) for pc_rope_append_ex / pc_attn_fwd_ex."""
        return (self.lo[:, layer, 0], self.lo[:, layer, 1], self.batch_stride, self.head_stride, 0)

    # strides in elements
    @property
    def batch_stride(self) -> int:
        return self.L * 2 * self.Hkv * self.cap * self.D

    @property
    def head_stride(self) -> int:
        return self.cap * self.D

    def k_plane(self, layer: int) -> torch.Tensor:
        return self.buf[:, layer, 0]

    def v_plane(self, layer: int) -> torch.Tensor:
        return self.buf[:, layer, 1]

    def views(self, length: Optional[int] = None) -> "StagedKV":
        """Per-layer ``(K, V)`` views ``[B, Hkv, length, D]`` (no copy), as an indexable sequence."""
        n = self.length if length is None else length
        return StagedKV(self, n)

    def grown(self, new_cap: int) -> "KVArena":
        """A larger arena holding the same ``length`` rows (rare path: generation ran past ``cap``)."""
        self.materialize()
        a = KVArena(self.B, self.L, self.Hkv, new_cap, self.D, self.buf.device, self.buf.dtype)
        a.buf[:, :, :, :, :self.length].copy_(self.buf[:, :, :, :, :self.length])
        if self.lo is not None:
            a.with_lo().lo[:, :, :, :, :self.length].copy_(self.lo[:, :, :, :, :self.length])
            a.lo_len = self.lo_len
        a.length = self.length
        a.tail_lo, a.tail_base, a.tail_len = self.tail_lo, self.tail_base, self.tail_len   # indexed by key, not by row capacity
        return a


class StagedKV(Sequence):
    """``past_key_values`` as the reference's callers index it (``[layer][0|1]``), plus the arena it
    aliases so the model can append in place.  Entries are ``[B, Hkv, length, D]`` views (``[Hkv, length, D]`` for the
    unbatched form ``CacheEngine.process`` returns).

    The 2 x n_layers views are built when somebody first LOOKS at them -- indexing, iterating, comparing, copying -- which is
    what a caller that inspects or rebuilds the list does (the reference's GenerationEngine, generation_engine.py:101-102); at
    that moment a staging the arena still owes (``KVArena.pending``) is carried out too.  Handing the object itself to the
    model does neither: the model finds the arena through ``.arena`` and may stage inside its first attention launches -- and
    the engines' own hot path never pays for 64 tensor views per call.

    A ``collections.abc.Sequence`` (not a ``list`` subclass: a list's C fast paths -- ``copy()``, ``+``, ``==``, pickling,
    ``PySequence_Fast`` consumers -- read the underlying storage directly and would see an empty list without carrying out a
    pending staging); every read accessor, inherited mixins included, goes through ``__getitem__`` / ``__iter__``."""

    def __init__(self, arena: KVArena, length: int, batched: bool = True):
        self.arena = arena
        self._arena: Optional[KVArena] = None      # the arena once an entry was replaced (see __setitem__)
        self.length = length
        self._batched = batched
        self._items: Optional[list] = None

    def _look(self) -> list:
        if self._items is not None and self.arena is None:
            return self._items
        a = self.arena
        if a.pending is not None:
            a.materialize()
        if self._items is None:
            n, buf = self.length, a.buf
            if self._batched:
                self._items = [(buf[:, i, 0, :, :n], buf[:, i, 1, :, :n]) for i in range(a.L)]
            else:
                self._items = [(buf[0, i, 0, :, :n], buf[0, i, 1, :, :n]) for i in range(a.L)]
        return self._items

    def __len__(self):
        return self._live_arena.L

    def __getitem__(self, i):
        return self._look()[i]

    def __setitem__(self, i, v):
        # an assigned entry is foreign data: from here on the model must not take the arena shortcut (`.arena` reads as None and
        # arena_from_past() inspects the items -- a replaced layer fails its stride test and the whole list is copied into a
        # fresh arena, replaced entries included)
        self._look()[i] = v
        if self.arena is not None:
            self._arena, self.arena = self.arena, None

    @property
    def _live_arena(self) -> KVArena:
        return self.arena if self.arena is not None else self._arena

    def __iter__(self):
        return iter(self._look())

    def __eq__(self, other):
        if isinstance(other, StagedKV):
            other = other._look()
        return self._look() == other

    __hash__ = None

    def __add__(self, other):
        return self._look() + list(other)

    def __radd__(self, other):
        return list(other) + self._look()

    def copy(self) -> list:
        return list(self._look())

    def __reduce__(self):
        return (list, (self._look(),))       # pickles as the plain list of views it stands for

    def __repr__(self):
        a = self._live_arena
        return f"StagedKV(layers={a.L}, length={self.length}, pending={a.pending is not None}, replaced={self.arena is None})"

    def unbatched(self) -> "StagedKV":
        """``[Hkv, length, D]`` views: what ``CacheEngine.process`` returns (``cache_engine.py:161-165``)."""
        return StagedKV(self._live_arena, self.length, batched=False)


def arena_from_past(past, n_layers: int, n_kv_heads: int, head_dim: int) -> Optional[Tuple[KVArena, int]]:
    """Recover the arena behind ``past_key_values``.

    ``StagedKV`` carries it explicitly.  A plain list of views (the reference's ``GenerationEngine``
    rebuilds the list with ``k.unsqueeze(0)``, ``generation_engine.py:101-102``) is recognised by its
    strides: K ``[B, Hkv, S, D]`` with strides ``(B-stride, cap*D, D, 1)`` and V exactly one K-plane
    further.  Returns ``None`` for foreign tensors (the caller then copies them into a fresh arena).
    """
    if past is None:
        return None
    arena = getattr(past, "arena", None)
    if arena is not None:
        return arena, past.length
    try:
        k0, v0 = past[0][0], past[0][1]
        if k0.dim() == 3:
            k0, v0 = k0.unsqueeze(0), v0.unsqueeze(0)
        B, Hkv, S, D = k0.shape
        if Hkv != n_kv_heads or D != head_dim or k0.dtype != torch.float16 or len(past) != n_layers:
            return None
        if k0.stride(3) != 1 or k0.stride(2) != D or k0.stride(1) % D != 0:
            return None
        cap = k0.stride(1) // D
        plane = Hkv * cap * D
        es = k0.element_size()
        if cap < S or v0.data_ptr() - k0.data_ptr() != plane * es or v0.stride() != k0.stride():
            return None
        for i in range(1, n_layers):
            ki = past[i][0]
            if ki.data_ptr() - k0.data_ptr() != 2 * plane * es * i:
                return None
        if B > 1 and k0.stride(0) != n_layers * 2 * plane:
            return None
        st = k0.untyped_storage()
        need_end = k0.storage_offset() + B * n_layers * 2 * plane
        if need_end * es > st.nbytes():
            return None
        live = _LIVE.get(k0.data_ptr())
        if live is not None and (live.B, live.L, live.Hkv, live.cap, live.D) == (B, n_layers, Hkv, cap, D) and \
                live.buf.dtype == k0.dtype:
            return live, S
        a = KVArena.__new__(KVArena)
        a.B, a.L, a.Hkv, a.cap, a.D = B, n_layers, Hkv, cap, D
        a.buf = torch.empty(0, dtype=k0.dtype, device=k0.device).set_(
            st, k0.storage_offset(), (B, n_layers, 2, Hkv, cap, D),
            (n_layers * 2 * plane, 2 * plane, plane, cap * D, D, 1))
        a.length = S
        a.lo, a.lo_len = None, 0
        a.tail_lo, a.tail_base, a.tail_len = None, -1, 0
        a.pending, a.row_tab = None, None
        return a, S
    except Exception:
        return None
