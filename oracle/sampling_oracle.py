"""numpy restatement of the reference's sampling front end.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported
by tests/, never by the product.

Follows ``promptcache/generation_engine.py``:
  * chain construction and thresholds                 :32-42   (temperature, repetition penalty, top-p, top-k, in that order)
  * application to the last row of logits             :150-155 (history ids only when repetition_penalty > 1)
  * greedy rule                                       :159     (temperature < 1e-5 or top_p < 1e-8 -> argmax)
The four processors are ``transformers`` classes (pinned 4.34.0 in the reference's requirements.txt:13, 5.x installed
here); their published semantics are restated below.  Pinned by tests/golden/sampling_chain.npz, which
``oracle/gen_golden.py --sampling-only`` produced by running the REFERENCE's own ``GenerationParameters`` chain.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

NEG_INF = -np.inf


def process_logits(logits: np.ndarray, history: Optional[Sequence[int]], temperature: float, repetition_penalty: float,
                   top_p: float, top_k: int) -> np.ndarray:
    """One row of logits [V] -> processed logits [V] (fp32, filtered entries = -inf)."""
    x = np.asarray(logits, dtype=np.float32).copy()
    if temperature >= 1e-5 and temperature != 1.0:                 # TemperatureLogitsWarper
        x = x / np.float32(temperature)
    if repetition_penalty > 1.0:                                   # RepetitionPenaltyLogitsProcessor (CTRL rule)
        idx = np.unique(np.asarray(history, dtype=np.int64))
        v = x[idx]
        x[idx] = np.where(v < 0, v * np.float32(repetition_penalty), v / np.float32(repetition_penalty))
    if 1e-8 <= top_p < 1.0:                                        # TopPLogitsWarper (min_tokens_to_keep = 1)
        order = np.argsort(x, kind="stable")                       # ascending, as torch.sort(descending=False)
        sx = x[order]
        e = np.exp(sx - sx.max())
        cum = np.cumsum((e / e.sum()).astype(np.float32), dtype=np.float32)
        remove = cum <= np.float32(1.0 - top_p)
        remove[-1] = False                                         # always keep the most probable token
        x[order[remove]] = NEG_INF
    if top_k > 0:                                                  # TopKLogitsWarper
        k = min(int(top_k), x.shape[0])
        kth = np.sort(x)[-k]
        x[x < kth] = NEG_INF
    return x


def is_greedy(temperature: float, top_p: float) -> bool:
    return temperature < 1e-5 or top_p < 1e-8
