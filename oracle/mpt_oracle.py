"""CPU oracle for the MPT adapter of the prompt-cache path.  TEST INFRASTRUCTURE ONLY (same rules as
``oracle/llama_oracle.py``).

numpy restatement of ``promptcache/model/mpt.py`` as the reference's ``Mpt`` adapter runs it
(``promptcache/model/__init__.py:261-288``, ``use_full_position_ids = True``): ALiBi attention whose bias row is
gathered at the POSITION IDS of all keys (cached and new), bias-free LayerNorms, GELU MLP.  Pinned against the
reference ``MptForCausalLM`` run in the build container (``oracle/gen_golden.py`` -> ``tests/golden/model_mpt_*.npz``).

Interface difference from the other oracles: ``position_ids`` covers every key, i.e. has ``past_len + q_len`` entries
per row (``mpt.py:172``: ``position_bias[:, :, position_ids]`` must match the key length).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from .falcon_oracle import gelu
from .llama_oracle import F32, causal_mask, softmax_rows


@dataclass
class MptOracleConfig:
    vocab_size: int
    hidden_size: int
    num_hidden_layers: int
    num_attention_heads: int
    layer_norm_epsilon: float = 1e-5
    alibi_bias_max: int = 8

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    num_key_value_heads = property(lambda self: self.num_attention_heads)


def alibi_slopes(num_heads: int, alibi_bias_max: int = 8) -> np.ndarray:
    """``build_mpt_alibi_tensor`` (mpt.py:90-110): slopes 1 / 2**(i * bias_max / n), i = 1..n.  The reference only
    works for power-of-two head counts (its ``slopes.view(1, num_heads, 1, 1)`` at :104 raises before the
    interleaving branch for any other count), so that is all this restates."""
    n2 = 2 ** math.ceil(math.log2(num_heads))
    if n2 != num_heads:
        raise ValueError(f"MPT ALiBi slopes need a power-of-two head count (got {num_heads}): mpt.py:104")
    base = np.arange(1, n2 + 1, dtype=F32) * F32(alibi_bias_max / n2)
    return (F32(1.0) / np.power(F32(2.0), base)).astype(F32)


def layernorm_nobias(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    """``LayerNorm`` with ``bias = None`` (mpt.py:205-215)."""
    x = x.astype(F32)
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = np.mean(xc * xc, axis=-1, keepdims=True, dtype=F32)
    return xc * (F32(1.0) / np.sqrt(var + F32(eps))) * w


class MptOracle:
    """``MptForCausalLM.forward`` (mpt.py:650-700) -> ``MptModel.forward`` (:464-583) -> ``MptBlock.forward`` (:227-266)
    -> ``MptAttention.forward`` (:132-190) / ``MptMLP.forward`` (:196-203).

    ``weights`` keys: ``embed``; per layer ``l{i}.ln1``, ``l{i}.wqkv`` [3*hid, hid] (q | k | v, :144), ``l{i}.wo``,
    ``l{i}.ln2``, ``l{i}.w1`` (up_proj), ``l{i}.w2`` (down_proj); ``lnf``; ``lm_head``.
    """

    def __init__(self, cfg: MptOracleConfig, weights: Dict[str, np.ndarray]):
        self.cfg = cfg
        self.w = {k: np.asarray(v, dtype=F32) for k, v in weights.items()}
        self.slopes = alibi_slopes(cfg.num_attention_heads, cfg.alibi_bias_max)

    def forward(self, input_ids: np.ndarray, position_ids: np.ndarray,
                past: Optional[Sequence[Tuple[np.ndarray, np.ndarray]]] = None,
                n_layers: Optional[int] = None, want_attn0: bool = False):
        """position_ids: [B, past_len + q] -- the position id of EVERY key (with no past: of the q new tokens)."""
        c, w = self.cfg, self.w
        B, ql = input_ids.shape
        H, D, hid = c.num_attention_heads, c.head_dim, c.hidden_size
        L = c.num_hidden_layers if n_layers is None else n_layers
        past_len = 0 if past is None else past[0][0].shape[2]
        assert position_ids.shape[1] == past_len + ql, "MPT needs the position id of every key (full position ids)"
        x = w["embed"][input_ids]                                                           # :497
        max_len = int(position_ids.max()) + 1                                               # :522
        # alibi[h, t] = (t - (max_len - 1)) * slope[h]  (:97, :109), gathered at the keys' position ids (:172)
        bias = (position_ids.astype(F32)[:, None, None, :] - F32(max_len - 1)) * self.slopes[None, :, None, None]
        mask = causal_mask(ql, past_len)[None, None] if True else None                      # :525-529 (also for q == 1: all visible)
        scale = F32(1.0 / math.sqrt(D))                                                     # :139-140
        present = []
        attn0 = None
        for i in range(L):
            h = layernorm_nobias(x, w[f"l{i}.ln1"], c.layer_norm_epsilon)                   # :240
            qkv = h @ w[f"l{i}.wqkv"].T                                                     # :143
            q, k, v = (t.reshape(B, ql, H, D).transpose(0, 2, 1, 3) for t in np.split(qkv, 3, axis=2))   # :144-147
            if past is not None:                                                            # :149-153
                k = np.concatenate([past[i][0].astype(F32), k], axis=2)
                v = np.concatenate([past[i][1].astype(F32), v], axis=2)
            present.append((k, v))
            s = np.matmul(q, np.swapaxes(k, 2, 3)) * scale + bias                           # :157, :174
            s = np.where(mask < 0, np.finfo(F32).min, s)                                    # masked_fill :177
            p = softmax_rows(s.astype(F32))                                                 # :180
            a = np.matmul(p, v).transpose(0, 2, 1, 3).reshape(B, ql, hid)                   # :183-184
            if i == 0:
                attn0 = a
            x = x + a @ w[f"l{i}.wo"].T                                                     # :185, :254
            h = layernorm_nobias(x, w[f"l{i}.ln2"], c.layer_norm_epsilon)                   # :256
            x = x + gelu(h @ w[f"l{i}.w1"].T) @ w[f"l{i}.w2"].T                             # :197-201
        x = layernorm_nobias(x, w["lnf"], c.layer_norm_epsilon)                             # :568
        logits = (x @ w["lm_head"].T).astype(F32)
        if want_attn0:
            return logits, present, attn0
        return logits, present
