"""CPU restatement of LLM.int8() -- what ``load_in_8bit=True`` means in the reference's GPU runs.  TEST INFRASTRUCTURE ONLY
(see ``llama_oracle.py`` for the import rules).

Where the algorithm lives.  The reference loads every GPU model through ``transformers`` with ``load_in_8bit=True``
(``demo.py:27-29``, ``eval.py:36-42``, ``config/llm_config_*.json:5``); transformers swaps each decoder-layer ``nn.Linear`` for
``bitsandbytes.nn.Linear8bitLt(has_fp16_weights=False, threshold=6.0)`` (``llm_int8_threshold`` default 6.0, ``lm_head``
skipped).  ``bitsandbytes`` is a third-party dependency (``requirements.txt`` pins ``bitsandbytes==0.41.1``) that is neither
vendored under /root/reference nor installed here, and it has no CPU path, so it cannot be run: PARITY IS UNPINNED against
bitsandbytes itself.  What is restated is its PUBLISHED algorithm -- Dettmers, Lewis, Belkada, Zettlemoyer, "LLM.int8(): 8-bit
Matrix Multiplication for Transformers at Scale" (NeurIPS 2022), section 3 (vector-wise quantisation, eq. 3-5, and the
mixed-precision decomposition, eq. 6-8), in the order bitsandbytes' ``MatMul8bitLt.forward`` applies it:

  weights (once, at load)    CB = round(W * 127 / absmax_row(W))  int8,  SCB = absmax_row(W)            vector-wise, per output row
  per call, X = fp16(input)  outlier columns  O = {k : |X[t, k]| >= threshold for some row t}            eq. 6, threshold 6.0
                             X0 = X with every entry |x| >= threshold zeroed;  SCA = absmax_row(X0)
                             CA = round(X0 * 127 / SCA) int8, columns in O zeroed                        vector-wise, per token row
                             Y  = (CA . CB^T as int32) * SCA[t] * SCB[n] / 127^2                         eq. 3 (dequantised in fp32)
                                  + X[:, O] . fp16(CB[:, O] * SCB / 127)^T                               eq. 8: the fp16 part
Rounding is round-half-to-even (``torch.round`` / ``__float2int_rn``).  bitsandbytes rounds Y to fp16; the build keeps it in
fp32 (its residual stream is fp32) -- the one deliberate difference, and it only removes a rounding.

Anchored on the reference's call sites above and on self-consistency properties (tests/test_llmint8_cpu.py): integers
reproduced exactly by the product's quantiser, zero threshold == plain vector-wise int8, an outlier column passes through in
fp16, the weight quantiser is the one of ``int8_oracle.py``.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from .int8_oracle import LINEAR_KEYS, quantize_rows_int8
from .llama_oracle import LlamaOracle, OracleConfig

F32 = np.float32
THRESHOLD = 6.0
INV_127SQ = F32(6.200012e-05)        # bitsandbytes' constant for 1 / (127 * 127) in int8_mm_dequant


def quantize_activations(x: np.ndarray, threshold: float = THRESHOLD):
    """``int8_vectorwise_quant(A.to(fp16), threshold)``: -> (CA int8 [T, K], SCA fp32 [T], outlier column indices, X fp16)."""
    x16 = np.asarray(x, dtype=F32).astype(np.float16)
    a = x16.astype(F32)
    outl = (np.abs(a) >= F32(threshold)) if threshold > 0 else np.zeros(a.shape, bool)
    cols = np.flatnonzero(outl.any(axis=0))
    a0 = np.where(outl, F32(0), a)
    sca = np.abs(a0).max(axis=1).astype(F32)
    ok = sca > 0
    inv = np.where(ok, F32(127.0) / np.where(ok, sca, F32(1)), F32(0)).astype(F32)
    ca = np.clip(np.rint(a0 * inv[:, None]), -127, 127).astype(np.int8)
    ca[:, cols] = 0
    return ca, sca, cols, x16


def linear(x: np.ndarray, cb: np.ndarray, scale: np.ndarray, threshold: float = THRESHOLD) -> np.ndarray:
    """LLM.int8 linear: ``x`` [T, K] (any float), ``cb`` int8 [N, K], ``scale`` = absmax_row(W) / 127 (the product's and
    ``int8_oracle.quantize_rows_int8``'s convention: SCB = 127 * scale).  Returns fp32 [T, N]."""
    ca, sca, cols, x16 = quantize_activations(x, threshold)
    out32 = ca.astype(np.int64) @ cb.T.astype(np.int64)                       # exact
    scb = (scale.astype(F32) * F32(127.0)).astype(F32)
    y = out32.astype(F32) * (sca[:, None] * scb[None, :]) * INV_127SQ
    if cols.size:
        sub_b = (cb[:, cols].astype(F32) * scale.astype(F32)[:, None]).astype(np.float16)      # CB * SCB / 127 in fp16
        y = y + x16[:, cols].astype(F32) @ sub_b.astype(F32).T
    return y.astype(F32)


class LlamaInt8Oracle(LlamaOracle):
    """The Llama oracle with every decoder-layer projection replaced by the LLM.int8 linear (embeddings, norms, lm_head
    stay fp32: ``llm_int8_skip_modules`` default)."""

    def __init__(self, cfg: OracleConfig, weights: Dict[str, np.ndarray], threshold: float = THRESHOLD):
        super().__init__(cfg, weights)
        self.threshold = threshold
        self.q: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}
        for k, v in self.w.items():
            if v.ndim == 2 and k.startswith("l") and k.split(".")[-1] in LINEAR_KEYS:
                self.q[k] = quantize_rows_int8(v)

    def _lin(self, x: np.ndarray, key: str) -> np.ndarray:
        if key not in self.q:
            return super()._lin(x, key)
        cb, scale = self.q[key]
        shp = x.shape
        return linear(x.reshape(-1, shp[-1]), cb, scale, self.threshold).reshape(*shp[:-1], cb.shape[0])
