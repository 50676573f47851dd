"""Pins the CPU oracle (numpy restatement) against outputs of the REFERENCE implementation captured by
oracle/gen_golden.py.  fp32 vs fp32: tolerances are fp32 round-off of different summation orders."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import engine_oracle as eo
from tests import helpers as H


@pytest.mark.parametrize("case", H.MODEL_CASES + H.FALCON_CASES + H.MPT_CASES)
def test_oracle_reproduces_reference_outputs(case):
    g = H.load_case(case)
    shape, schema, jobs, prompt, used, ids, pos = H.layout_for_case(g)
    # integer layout first (bit-exact)
    assert ids == g["input_ids"].tolist()
    pos = H.full_positions(g, used, pos)
    assert pos == g["position_ids"].tolist()
    assert [[u.offset, len(u)] for u in used] == g["seg_table"].tolist()
    model, _ = H.oracle_for_case(g, shape)
    lib = eo.encode_schema(model, jobs)
    # stored module KV (fp32 on the reference CPU path): sampled rows
    objs = {id(t): t for j in jobs for t in j["targets"]}
    mods = sorted(lib.keys(), key=lambda k: (objs[k].offset, len(objs[k])))
    assert [[objs[k].offset, len(objs[k])] for k in mods] == g["mod_table"].tolist()
    for j, key in enumerate(mods):
        np.testing.assert_allclose(lib[key][0][0][:, 0], g["mod_k_first"][j], atol=2e-5, rtol=1e-4)
        np.testing.assert_allclose(lib[key][-1][1][:, -1], g["mod_v_last"][j], atol=2e-5, rtol=1e-4)
    staged, S, (logits, present, attn0) = eo.cached_prefill(model, lib, used, ids, pos, int(g["max_ctx"]), want_attn0=True)
    assert S == int(g["S"])
    rows = g["staged_rows"]
    got_k = np.stack([k[:, rows] for k, _ in staged])
    got_v = np.stack([v[:, rows] for _, v in staged])
    # fp16 staging of fp32 values that agree to ~1e-6: allow one fp16 ulp on a handful of rounding ties
    assert np.mean(got_k.view(np.uint16) != g["staged_k"].view(np.uint16)) < 1e-2
    np.testing.assert_allclose(got_k.astype(np.float32), g["staged_k"].astype(np.float32), atol=2e-3, rtol=1e-3)
    np.testing.assert_allclose(got_v.astype(np.float32), g["staged_v"].astype(np.float32), atol=2e-3, rtol=1e-3)
    np.testing.assert_allclose(attn0[0], g["attn0"], atol=1e-4, rtol=1e-3)
    np.testing.assert_allclose(present[0][0][0, :, S:], g["new_k0"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(logits[0], g["logits_cached"], atol=2e-3, rtol=1e-3)
    assert np.abs(logits[0] - g["logits_cached"]).max() < 2e-3
    toks = eo.generate_greedy(model, logits, present, pos, len(g["greedy"]), use_full_position_ids=H.is_mpt(g))
    assert toks == g["greedy"].tolist()


@pytest.mark.parametrize("case", ["tiny_trip", "mid_mha_doc", "falcon_mid_doc", "mpt_mid_doc"])
def test_oracle_nocache_path(case):
    from promptcache_amd.pml import Prompt
    g = H.load_case(case)
    shape, schema, jobs, prompt, used, ids, pos = H.layout_for_case(g)
    # no_cache re-packing (cache_engine.py:476-493): all tokens sorted by position, positions = range(N)
    seg_tokens = []
    for seq in used:
        seg_tokens += list(zip(seq.position_ids(), seq.token_ids()))
    pairs = sorted(seg_tokens + list(zip(pos, ids)))
    nids = [t for _, t in pairs]
    assert nids == g["nocache_ids"].tolist()
    assert list(range(len(pairs))) == g["nocache_pos"].tolist()
    model, _ = H.oracle_for_case(g, shape)
    logits, _ = model.forward(np.asarray([nids]), np.asarray([list(range(len(nids)))]))
    np.testing.assert_allclose(logits[0, -1], g["logits_nocache_last"], atol=2e-3, rtol=1e-3)


def test_union_free_schema_cached_equals_nocache_up_to_kv_rounding():
    """SURVEY.md section 7 invariant: with every module of a union-free schema selected, the cached path
    equals the no-cache path up to the fp16 rounding of staged KV."""
    g = H.load_case("mid_mha_doc")
    assert np.abs(g["logits_cached"][-1] - g["logits_nocache_last"]).max() < 5e-3


def test_fixture_key_sets_match_the_generator_manifest():
    """Every committed .npz holds exactly the arrays the committed oracle/gen_golden.py writes (tests/golden/manifest.json
    is written by the same run): a stale fixture fails here instead of looking like a changed reference."""
    with open(os.path.join(H.GOLD, "manifest.json")) as f:
        man = json.load(f)
    assert "lxml STAND-IN" in man["_provenance"]["pml_layout.json, pml_recover.json, pml/*.xml"]   # (how the fixtures were made)
    man = {k: v for k, v in man.items() if not k.startswith("_")}
    files = sorted(os.path.basename(p) for p in glob.glob(os.path.join(H.GOLD, "*.npz")))
    assert files == sorted(man)
    for fn in files:
        with np.load(os.path.join(H.GOLD, fn), allow_pickle=False) as z:
            assert sorted(z.files) == man[fn], fn
    model_keys = {tuple(v) for k, v in man.items() if k.startswith("model_")}
    assert len(model_keys) == 1, "every model fixture carries the same arrays"
