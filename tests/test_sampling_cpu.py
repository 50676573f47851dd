"""The sampling front end of ``GenerationEngine`` (reference generation_engine.py:32-42, :149-168) on the CPU: the product's
logits-processor chain and the numpy oracle against the fixture the REFERENCE's own chain produced
(tests/golden/sampling_chain.npz, made by ``oracle/gen_golden.py --sampling-only``)."""
import os

import numpy as np
import torch

from oracle import sampling_oracle as so
from promptcache_amd.generation_engine import GenerationParameters

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampling_chain.npz")


def _cases():
    z = np.load(GOLD)
    for i, (t, rp, tp, tk) in enumerate(z["params"]):
        yield float(t), float(rp), float(tp), int(tk), z["logits"][i], z["history"][i], z["processed"][i]


def test_processor_chain_and_oracle_match_reference_fixture():
    n = 0
    for t, rp, tp, tk, logits, hist, want in _cases():
        params = GenerationParameters(temperature=t, repetition_penalty=rp, top_p=tp, top_k=tk)
        chain = params.get_logits_processor()
        h = torch.as_tensor([hist.tolist()]) if rp > 1.0 else None
        got = chain(h, torch.from_numpy(logits[None].copy()))[0].numpy()
        orc = so.process_logits(logits, hist, t, rp, tp, tk)
        keep = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), keep) and np.array_equal(np.isfinite(orc), keep), (t, rp, tp, tk)
        np.testing.assert_allclose(got[keep], want[keep], rtol=0, atol=0)
        np.testing.assert_allclose(orc[keep], want[keep], rtol=1e-6, atol=1e-6)
        assert params.greedy == so.is_greedy(t, tp)
        n += 1
    assert n == 10


def test_greedy_rule_and_defaults():
    assert GenerationParameters(temperature=0.0).greedy and GenerationParameters(top_p=0.0).greedy
    assert not GenerationParameters().greedy
    assert len(GenerationParameters().get_logits_processor()) == 0          # all defaults: empty chain (:32-42)
    assert len(GenerationParameters(temperature=0.7, repetition_penalty=1.1, top_p=0.9, top_k=3).get_logits_processor()) == 4
