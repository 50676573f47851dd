"""CPU restatement of the weight-only int8 mode (TEST INFRASTRUCTURE ONLY -- see ``llama_oracle.py`` for the import rules).

The reference loads its GPU models with ``load_in_8bit=True`` (``config/llm_config_*.json:5``, ``eval.py:36-42``,
``demo.py:28``), i.e. through bitsandbytes' LLM.int8 (``bitsandbytes`` is a pinned dependency, ``requirements.txt``, that is
NOT in ``/root/reference`` nor installed here).  What is restated is the published weight quantiser of that scheme --
row-wise absmax int8 (Dettmers et al., "LLM.int8()", 2022, section 3.1: ``X_i8 = round(127 / max|X_row| * X_row)``, dequantised with ``max|X_row| / 127``) -- and
NOT its activation path: the build keeps activations in split-precision fp16 (weight-only int8), which is strictly closer
to the fp32 reference than int8 activations are.  PARITY UNPINNED against bitsandbytes itself (no copy of it to run);
pinned are (a) the quantiser, bit-exact between this numpy version (IEEE fp32 divisions) and ``_native.quantize_rows_int8``, and (b) the
forward over the DEQUANTISED weights ``q * scale`` (fp32) through the ordinary oracle, which is what the W8 kernels must
reproduce to the usual 1e-2.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def quantize_rows_int8(w: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    wf = w.astype(np.float32)
    amax = np.abs(wf).max(axis=1)
    ok = amax > 0
    safe = np.where(ok, amax, np.float32(1.0)).astype(np.float32)
    scale = np.where(ok, safe / np.float32(127.0), np.float32(1.0)).astype(np.float32)
    inv = np.where(ok, np.float32(127.0) / safe, np.float32(0.0)).astype(np.float32)
    q = np.clip(np.rint(wf * inv[:, None]), -127, 127).astype(np.int8)      # rint: round half to even, as torch.round
    return q, scale


def dequantize(q: np.ndarray, scale: np.ndarray) -> np.ndarray:
    return q.astype(np.float32) * scale[:, None].astype(np.float32)


LINEAR_KEYS = ("wq", "wk", "wv", "wo", "gate", "up", "down")      # the Llama oracle's per-layer names: "l{i}.wq", ...
LINEAR_KEYS_FUSED = ("wqkv", "wo", "w1", "w2")                    # the Falcon / MPT oracles' ("l{i}.wqkv", ...)


def dequantized_weights(weights: Dict[str, np.ndarray], keys=LINEAR_KEYS_FUSED) -> Dict[str, np.ndarray]:
    """The same for any of the oracles' weight dicts: per-layer 2-D tensors named ``l{i}.<key>`` with ``key in keys``."""
    return {k: (dequantize(*quantize_rows_int8(v)) if v.ndim == 2 and k.startswith("l") and k.split(".")[-1] in keys
                else v.astype(np.float32)) for k, v in weights.items()}


def dequantized_llama_weights(weights: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """fp32 weight dict with every decoder-layer linear replaced by ``dequantize(quantize(w))`` (embeddings, norms and
    lm_head stay as they are: LLM.int8 skips ``lm_head``, ``llm_int8_skip_modules`` default)."""
    out = {}
    for k, v in weights.items():
        if v.ndim == 2 and k.startswith("l") and k.split(".")[-1] in LINEAR_KEYS:
            out[k] = dequantize(*quantize_rows_int8(v))
        else:
            out[k] = v.astype(np.float32)
    return out
