"""CPU oracle for the Falcon adapter of the prompt-cache path.  TEST INFRASTRUCTURE ONLY (see
``oracle/llama_oracle.py`` for the rules: only ``tests/``, ``smoke()`` and ``bench.py``'s cpu_baseline may import it).

numpy restatement of ``promptcache/model/falcon.py`` for the falcon-7b architecture the reference's ``Falcon``
adapter loads (``promptcache/model/__init__.py:206-258``): ``multi_query`` (one shared K/V head), ``parallel_attn``
(attention and MLP both read the single ``input_layernorm`` output), rotary positions at explicit position ids, no
linear biases, no alibi.  Pinned against the reference implementation run in the build container
(``oracle/gen_golden_falcon.py`` -> ``tests/golden/model_falcon_*.npz``, re-checked by ``tests/test_oracle_golden.py``).

Same interface as ``LlamaOracle`` (``forward(ids, pos, past) -> (logits, present)``), so ``oracle/engine_oracle.py``
drives either.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
from scipy.special import erf

from .llama_oracle import F32, apply_rope, attention_core, rope_cos_sin


@dataclass
class FalconOracleConfig:
    vocab_size: int
    hidden_size: int
    num_hidden_layers: int
    num_attention_heads: int
    layer_norm_epsilon: float = 1e-5
    rope_theta: float = 10000.0
    inv_freq: Optional[np.ndarray] = None      # see OracleConfig.inv_freq

    num_key_value_heads = 1

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


def layernorm(x: np.ndarray, w: np.ndarray, b: np.ndarray, eps: float) -> np.ndarray:
    """``torch.nn.LayerNorm`` (falcon.py:757, :1020): biased variance over the last dim, fp32."""
    x = x.astype(F32)
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = np.mean(xc * xc, axis=-1, keepdims=True, dtype=F32)
    return xc * (F32(1.0) / np.sqrt(var + F32(eps))) * w + b


def gelu(x: np.ndarray) -> np.ndarray:
    """``nn.GELU()`` (falcon.py:726): the exact erf form."""
    return (F32(0.5) * x * (F32(1.0) + erf(x / np.sqrt(F32(2.0))).astype(F32))).astype(F32)


class FalconOracle:
    """``FalconForCausalLM.forward`` (falcon.py:1274-1340) -> ``FalconModel.forward`` (:1070-1222) ->
    ``FalconDecoderLayer.forward`` (:761-816) -> ``FalconAttention.forward`` (:416-539) / ``FalconMLP`` (:730-733).

    ``weights`` keys (fp32 numpy): ``embed``; per layer ``l{i}.ln_w``, ``l{i}.ln_b``, ``l{i}.wqkv`` [(H+2)D, hid]
    (H query heads, then the key head, then the value head: ``_split_heads`` multi-query branch :393-396),
    ``l{i}.wo`` (``dense``), ``l{i}.w1`` (``dense_h_to_4h``), ``l{i}.w2`` (``dense_4h_to_h``); ``lnf_w``, ``lnf_b``;
    ``lm_head``.
    """

    def __init__(self, cfg: FalconOracleConfig, weights: Dict[str, np.ndarray]):
        self.cfg = cfg
        self.w = {k: np.asarray(v, dtype=F32) for k, v in weights.items()}

    def forward(self, input_ids: np.ndarray, position_ids: np.ndarray,
                past: Optional[Sequence[Tuple[np.ndarray, np.ndarray]]] = None,
                n_layers: Optional[int] = None, want_attn0: bool = False):
        """past / present: per layer (K, V) [B, 1, S, D] (the standard format ``_convert_cache_to_standard_format``
        hands back, :1218).  The mask is index-order causal with every past column visible
        (``_prepare_attn_mask`` :1031-1060); position ids only steer the rotation (:443-444, :117-130)."""
        c, w = self.cfg, self.w
        B, ql = input_ids.shape
        H, D = c.num_attention_heads, c.head_dim
        L = c.num_hidden_layers if n_layers is None else n_layers
        past_len = 0 if past is None else past[0][0].shape[2]
        x = w["embed"][input_ids]                                            # :1112
        cos, sin = rope_cos_sin(position_ids, D, c.rope_theta, c.inv_freq)   # :117-130 gathered at position_ids
        present = []
        attn0 = None
        for i in range(L):
            h = layernorm(x, w[f"l{i}.ln_w"], w[f"l{i}.ln_b"], c.layer_norm_epsilon)          # :779
            fused = (h @ w[f"l{i}.wqkv"].T).reshape(B, ql, H + 2, D)                          # :428, :393-396
            q = fused[:, :, :H].transpose(0, 2, 1, 3)
            k = fused[:, :, H:H + 1].transpose(0, 2, 1, 3)
            v = fused[:, :, H + 1:].transpose(0, 2, 1, 3)
            q = apply_rope(q, cos, sin)                                                       # :444, :155
            k = apply_rope(k, cos, sin)
            if past is not None:                                                              # :446-452
                k = np.concatenate([past[i][0].astype(F32), k], axis=2)
                v = np.concatenate([past[i][1].astype(F32), v], axis=2)
            present.append((k, v))
            a = attention_core(q, k, v, past_len, H)      # scaled_dot_product_attention with the float mask, :477-479
            a = a.transpose(0, 2, 1, 3).reshape(B, ql, H * D)
            if i == 0:
                attn0 = a
            attn_out = a @ w[f"l{i}.wo"].T                                                    # :493
            mlp_out = gelu(h @ w[f"l{i}.w1"].T) @ w[f"l{i}.w2"].T                             # :798 parallel_attn, :731-732
            x = x + (mlp_out + attn_out)                                                      # :810-813
        x = layernorm(x, w["lnf_w"], w["lnf_b"], c.layer_norm_epsilon)                        # :1210
        logits = (x @ w["lm_head"].T).astype(F32)                                             # :1311
        if want_attn0:
            return logits, present, attn0
        return logits, present
