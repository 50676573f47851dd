"""CPU oracle for the prompt-cache prefill path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference's algorithm for the hot path
(module-KV gather -> position-id-aware RoPE -> masked attention over staged KV
-> Llama layer stack).  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``prompt-cache_amd/``) never imports anything from
``oracle/`` and has no CPU fallback.

Parity pinning: the reference repository holds no golden vector / KAT for this
path, so this restatement is pinned against outputs of the
reference implementation itself, imported in the build container by
``oracle/gen_golden.py`` (fixtures committed under ``tests/golden/``;
``tests/test_oracle_golden.py`` re-checks them on every run).

Every function cites the reference lines it follows (paths relative to the
reference checkout, e.g. ``promptcache/model/llama2.py``).

Numerics: fp32 everywhere, exactly like the reference CPU path, with the one
reference quirk kept: staged / stored module KV is rounded to fp16
(``cache_engine.py:105-106`` allocates the staged buffer as ``torch.half`` even
on CPU; the ``copy_`` at ``:148-149`` rounds).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32


@dataclass
class OracleConfig:
    vocab_size: int
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    # Optional model constant: the reference evaluates ``1/theta**(arange(0,D,2)/D)`` with torch's fp32
    # pow (llama2.py:121), which differs from numpy's fp32 pow by 1 ulp in some entries.  Callers that
    # hold the reference-evaluated table (fixtures, the product's own torch-CPU evaluation of the same
    # formula) pass it here so both sides rotate with identical constants.
    inv_freq: Optional[np.ndarray] = None

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


# --------------------------------------------------------------------------
# RoPE  (promptcache/model/llama2.py:114-147, :195-210)
# --------------------------------------------------------------------------

def rope_inv_freq(head_dim: int, theta: float) -> np.ndarray:
    """``inv_freq = 1 / theta ** (arange(0, D, 2) / D)`` (llama2.py:121): fp32 exponents, the power
    evaluated in float64 and rounded once to fp32 (= torch's fp32 result for theta = 1e4; see
    ``OracleConfig.inv_freq`` for bit-identical constants in the general case)."""
    ex = np.arange(0, head_dim, 2, dtype=F32) / F32(head_dim)
    pw = np.power(np.float64(theta), ex.astype(np.float64)).astype(F32)
    return (F32(1.0) / pw).astype(F32)


def rope_cos_sin(position_ids: np.ndarray, head_dim: int, theta: float,
                 inv_freq: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
    """cos/sin rows for the *supplied* position ids.

    The reference builds a table of ``max(position_ids)+1`` rows
    (llama2.py:357, :129-137: ``t`` fp32, ``freqs = outer(t, inv_freq)``,
    ``emb = cat(freqs, freqs)``) and indexes it with ``position_ids``
    (llama2.py:206-207).  Building only the indexed rows is the same numbers.
    """
    t = position_ids.astype(F32)
    inv = rope_inv_freq(head_dim, theta) if inv_freq is None else np.asarray(inv_freq, dtype=F32)
    freqs = t[..., None] * inv[None, :]  # fp32 product, as einsum("i,j->ij")
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb, dtype=F32), np.sin(emb, dtype=F32)


def rotate_half(x: np.ndarray) -> np.ndarray:
    """llama2.py:195-199 (half-split pairing: dim i with i + D/2)."""
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def apply_rope(x: np.ndarray, cos: np.ndarray, sin: np.ndarray) -> np.ndarray:
    """``x*cos + rotate_half(x)*sin`` (llama2.py:208-209). x: [B,H,q,D]; cos/sin: [B,q,D]."""
    return x * cos[:, None, :, :] + rotate_half(x) * sin[:, None, :, :]


# --------------------------------------------------------------------------
# Mask  (llama2.py:62-76, :798-819)
# --------------------------------------------------------------------------

def causal_mask(q_len: int, past_len: int) -> np.ndarray:
    """Additive mask [q, past+q]: zeros over ALL past columns, lower-triangular over the new
    tokens in *input order* (index order, not position order).  ``finfo.min`` fill."""
    m = np.full((q_len, q_len), np.finfo(F32).min, dtype=F32)
    idx = np.arange(q_len)
    m[idx[None, :] <= idx[:, None]] = 0.0
    if past_len > 0:
        m = np.concatenate([np.zeros((q_len, past_len), dtype=F32), m], axis=1)
    return m


# --------------------------------------------------------------------------
# Layer pieces
# --------------------------------------------------------------------------

def rmsnorm(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    """llama2.py:103-108."""
    x = x.astype(F32)
    var = np.mean(x * x, axis=-1, keepdims=True, dtype=F32)
    return w * (x * (F32(1.0) / np.sqrt(var + F32(eps))))


def silu(x: np.ndarray) -> np.ndarray:
    return x / (F32(1.0) + np.exp(-x))


def softmax_rows(s: np.ndarray) -> np.ndarray:
    """fp32 softmax over the last dim (llama2.py:387)."""
    m = s.max(axis=-1, keepdims=True)
    e = np.exp(s - m, dtype=F32)
    return e / e.sum(axis=-1, keepdims=True, dtype=F32)


def attention_core(q: np.ndarray, k: np.ndarray, v: np.ndarray, past_len: int,
                   n_rep: int = 1) -> np.ndarray:
    """``softmax(q k^T / sqrt(D) + mask) v`` (llama2.py:368-388).

    q: [B,H,q,D] (RoPE applied); k, v: [B,Hkv,S+q,D] (k RoPE applied).  Returns [B,H,q,D].
    ``repeat_kv`` (llama2.py:247-256) is a head broadcast.
    """
    B, H, ql, D = q.shape
    if n_rep > 1:
        k = np.repeat(k, n_rep, axis=1)
        v = np.repeat(v, n_rep, axis=1)
    s = np.matmul(q, np.swapaxes(k, 2, 3)) / F32(np.sqrt(D))
    if ql > 1:  # llama2.py:802 -- no causal mask is built for q_len == 1
        s = s + causal_mask(ql, past_len)[None, None]
    p = softmax_rows(s.astype(F32))
    return np.matmul(p, v)


class LlamaOracle:
    """Llama forward with explicit position ids and a legacy tuple KV cache.

    Follows ``LlamaForCausalLM.forward`` (llama2.py:986-1076) ->
    ``LlamaModel.forward`` (:822-951) -> ``LlamaDecoderLayer.forward`` (:600-654)
    -> ``LlamaAttention.forward`` (:315-410) / ``LlamaMLP.forward`` (:242).

    ``weights`` keys (fp32 numpy): ``embed`` [V,hid]; per layer i:
    ``l{i}.ln1``, ``l{i}.wq/wk/wv/wo`` ([out,in], nn.Linear layout), ``l{i}.ln2``,
    ``l{i}.gate/up/down``; ``norm``; ``lm_head`` [V,hid].
    """

    def __init__(self, cfg: OracleConfig, weights: Dict[str, np.ndarray]):
        self.cfg = cfg
        self.w = {k: np.asarray(v, dtype=F32) for k, v in weights.items()}

    def _lin(self, x: np.ndarray, key: str) -> np.ndarray:
        """``nn.Linear`` without bias: ``x @ W^T`` in fp32.  (One overridable place: ``oracle/llmint8_oracle.py`` swaps in
        the LLM.int8 linear for the decoder-layer projections.)"""
        return x @ self.w[key].T

    def forward(self, input_ids: np.ndarray, position_ids: np.ndarray,
                past: Optional[Sequence[Tuple[np.ndarray, np.ndarray]]] = None,
                n_layers: Optional[int] = None, want_attn0: bool = False):
        """input_ids, position_ids: [B,q] int.  past: per layer (K,V) [B,Hkv,S,D] (any float dtype;
        upcast to fp32 by ``cat`` type promotion exactly like the reference CPU path).

        Returns ``(logits [B,q,V] fp32, present list of (K,V) fp32 [B,Hkv,S+q,D])`` and, if
        ``want_attn0``, additionally layer 0's attention output before ``o_proj`` ([B,q,H*D]).
        """
        c, w = self.cfg, self.w
        B, ql = input_ids.shape
        H, Hkv, D = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        L = c.num_hidden_layers if n_layers is None else n_layers
        past_len = 0 if past is None else past[0][0].shape[2]
        x = w["embed"][input_ids]  # llama2.py:869
        cos, sin = rope_cos_sin(position_ids, D, c.rope_theta, c.inv_freq)
        present = []
        attn0 = None
        for i in range(L):
            res = x
            h = rmsnorm(x, w[f"l{i}.ln1"], c.rms_norm_eps)
            q = self._lin(h, f"l{i}.wq").reshape(B, ql, H, D).transpose(0, 2, 1, 3)
            k = self._lin(h, f"l{i}.wk").reshape(B, ql, Hkv, D).transpose(0, 2, 1, 3)
            v = self._lin(h, f"l{i}.wv").reshape(B, ql, Hkv, D).transpose(0, 2, 1, 3)
            q = apply_rope(q, cos, sin)
            k = apply_rope(k, cos, sin)
            if past is not None:  # llama2.py:361-364
                k = np.concatenate([past[i][0].astype(F32), k], axis=2)
                v = np.concatenate([past[i][1].astype(F32), v], axis=2)
            present.append((k, v))
            a = attention_core(q, k, v, past_len, H // Hkv)
            a = a.transpose(0, 2, 1, 3).reshape(B, ql, H * D)
            if i == 0:
                attn0 = a
            x = res + self._lin(a, f"l{i}.wo")
            res = x
            h = rmsnorm(x, w[f"l{i}.ln2"], c.rms_norm_eps)
            x = res + self._lin(silu(self._lin(h, f"l{i}.gate")) * self._lin(h, f"l{i}.up"), f"l{i}.down")
        x = rmsnorm(x, w["norm"], c.rms_norm_eps)
        logits = (x @ w["lm_head"].T).astype(F32)  # llama2.py:1050-1051, all q rows
        if want_attn0:
            return logits, present, attn0
        return logits, present


# --------------------------------------------------------------------------
# Cache engine pieces  (promptcache/cache_engine.py)
# --------------------------------------------------------------------------

def slice_segment_kv(present, batch_row: int, st: int, ed: int):
    """``SchemaCache._process`` slice-and-store (cache_engine.py:283-296): per layer
    ``K[j, :, st:ed, :]`` / ``V[j, :, st:ed, :]``.  Stored at the model dtype (fp32 on the CPU path)."""
    return [(k[batch_row, :, st:ed, :].copy(), v[batch_row, :, st:ed, :].copy()) for k, v in present]


def kv_gather(segments: Sequence[Sequence[Tuple[np.ndarray, np.ndarray]]], max_ctx: int):
    """``PromptCache.update`` on a fresh engine (cache_engine.py:115-156): concatenate the used
    segments, in order, into per-layer ``[H, max_ctx, D]`` **fp16** buffers (``:104-107``); returns
    the ``[:, :length, :]`` views (``:161-165``) and the staged length.

    The retention logic of ``:121-129`` is not reproduced (SURVEY.md section 7 "hard parts": it
    compares a sorted list with an unsorted one); a fresh engine stages exactly the used segments in
    input order, which is what every parity fixture exercises.
    """
    n_layers = len(segments[0])
    H, _, D = segments[0][0][0].shape
    staged = [(np.zeros((H, max_ctx, D), np.float16), np.zeros((H, max_ctx, D), np.float16))
              for _ in range(n_layers)]
    off = 0
    for seg in segments:
        ln = seg[0][0].shape[1]
        if off + ln > max_ctx:
            raise ValueError("staged KV exceeds max_ctx_length")
        for i in range(n_layers):
            staged[i][0][:, off:off + ln, :] = seg[i][0].astype(np.float16)  # copy_ rounds to half
            staged[i][1][:, off:off + ln, :] = seg[i][1].astype(np.float16)
        off += ln
    return [(k[:, :off, :], v[:, :off, :]) for k, v in staged], off


def greedy_decode(model: LlamaOracle, logits: np.ndarray, present, position_offset: int, steps: int) -> List[int]:
    """``GenerationEngine.generate`` greedy branch (generation_engine.py:123-168): argmax of the
    last row, then q_len=1 steps at position ``position_offset + i``."""
    out: List[int] = []
    for i in range(steps):
        tok = int(np.argmax(logits[0, -1]))
        out.append(tok)
        if i == steps - 1:
            break
        # generation_engine.py:132 -- position for loop index i+1 is position_offset + (i+1)
        logits, present = model.forward(np.array([[tok]]), np.array([[position_offset + i + 1]]), past=present)
    return out
