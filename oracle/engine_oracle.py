"""CPU restatement of the reference's cache-engine data path on top of ``llama_oracle``.
TEST INFRASTRUCTURE ONLY (see ``llama_oracle.py`` header for the import rules and the parity pinning).

Follows ``promptcache/cache_engine.py``:
  ``SchemaCache._process``  :185-308   one forward per encode path, slice per-segment KV, later paths
                                       overwrite earlier entries (``cache_l1[id(tc)] = ...`` :296)
  ``PromptCache.update``    :115-156   concatenate the used segments into fp16 staged buffers
  ``CacheEngine.process``   :388-522   (integer part is done by the caller with ``promptcache_amd.pml``,
                                       itself pinned against the reference by tests/golden/pml_layout.json)
and ``promptcache/generation_engine.py:94-168`` (prefill over the staged KV, greedy decode).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from .llama_oracle import LlamaOracle, greedy_decode, kv_gather, slice_segment_kv


def encode_schema(model: LlamaOracle, jobs: Sequence[dict]) -> Dict[int, list]:
    """``jobs``: per encode path ``{"token_ids", "position_ids", "targets": [TokenSequence, ...]}`` in the
    reference's path order (targets expose ``.offset`` and ``len()``).  Returns
    ``{id(token_sequence): per-layer [(K, V) fp32 [H, len, D]]}`` with last-writer-wins semantics, keyed
    like the reference's ``cache_l1`` (cache_engine.py:296)."""
    lib: Dict[int, list] = {}
    for job in jobs:
        ids = np.asarray([job["token_ids"]])
        pos = np.asarray([job["position_ids"]])
        _, present = model.forward(ids, pos)
        plist = list(job["position_ids"])
        for tc in job["targets"]:
            st = plist.index(tc.offset)  # cache_engine.py:278
            lib[id(tc)] = slice_segment_kv(present, 0, st, st + len(tc))
    return lib


def cached_prefill(model: LlamaOracle, lib, used: Sequence, input_ids: List[int],
                   position_ids: List[int], max_ctx: int, want_attn0: bool = False):
    """Gather ``used`` segments (fp16 staging) and run the new tokens over them."""
    staged, S = kv_gather([lib[id(u)] for u in used], max_ctx)
    past = [(k[None], v[None]) for k, v in staged]   # generation_engine.py:101-102
    out = model.forward(np.asarray([input_ids]), np.asarray([position_ids]), past=past, want_attn0=want_attn0)
    return staged, S, out


def generate_greedy(model: LlamaOracle, logits, present, position_ids: List[int], steps: int,
                    use_full_position_ids: bool = False) -> List[int]:
    """``use_full_position_ids`` (MPT, generation_engine.py:127-129): every step passes the prompt's full position list
    plus ``range(offset, offset + loop_index)`` -- the first decoded token sits at ``offset``, not ``offset + 1``."""
    if not use_full_position_ids:
        return greedy_decode(model, logits, present, max(position_ids) + 1, steps)
    offset = max(position_ids) + 1
    out: List[int] = []
    for i in range(steps):
        tok = int(np.argmax(logits[0, -1]))
        out.append(tok)
        if i == steps - 1:
            break
        pos = list(position_ids) + list(range(offset, offset + i + 1))
        logits, present = model.forward(np.array([[tok]]), np.array([pos]), past=present)
    return out
