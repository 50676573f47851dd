"""The product's PML front-end against the REFERENCE on 80 random schemas and 114 random prompts (tests/golden/pml_fuzz.json: generated
and laid out by oracle/gen_golden.py with the reference imported; the texts are seeded random PML authored by that script -- nested
modules, unions with and without a scaffold member, parameters, raw whitespace, escapes, the odd malformed character).  Integer work:
bit-exact -- schema length, encode paths, per-scaffold token / position CRCs, owned segments, no-cache prompt assembly; where the
reference raises (over-long parameter scaffolds, foreign tags in a union, unknown modules, over-long arguments) the product raises
the same exception type with the same message; where the reference's own encode would die (``select()`` returning None,
cache_engine.py:268) the product's plan dies the same way."""
import json
import os
import zlib

import numpy as np
import pytest

from tests import helpers as H
from promptcache_amd import pml

with open(os.path.join(H.GOLD, "pml_fuzz.json")) as f:
    CASES = json.load(f)


def crc(xs):
    return zlib.crc32(np.asarray(list(xs), dtype=np.int64).tobytes())


@pytest.mark.parametrize("i", range(len(CASES)))
def test_random_schema_and_prompts_match_the_reference(i):
    c = CASES[i]
    fmt = H.llama_formatter()
    lm = H.TokOnlyLM()
    if "error" in c:
        with pytest.raises(Exception) as ei:
            pml.Schema(fmt(c["schema"]), lm, max_tokens=c["max_tokens"])
        assert type(ei.value).__name__ == c["error"]["type"] and str(ei.value) == c["error"]["message"]
        return
    sc = pml.Schema(fmt(c["schema"]), lm, max_tokens=c["max_tokens"])
    assert len(sc) == c["length"]
    paths = sc.encode_paths()
    assert [str(p) for p in paths] == [p["path"] for p in c["paths"]]
    for p, exp in zip(paths, c["paths"]):
        sf = sc.get_scaffold(p)
        assert len(sf.token_ids()) == exp["n"]
        assert crc(sf.token_ids()) == exp["ids_crc"] and crc(sf.position_ids()) == exp["pos_crc"]
        sel = sf.select(p)
        if exp["targets"] is None:
            assert sel is None            # (the reference's encode of this schema raises AttributeError; so does SchemaCache._plan)
        else:
            assert [[t.offset, len(t)] for t in sel.all_token_sequences()] == exp["targets"]
    # prompts go through the PRODUCT's CacheEngine.process(no_cache=True) (no model needed: a no_cache schema is not encoded),
    # as the goldens went through the reference's (cache_engine.py:388-493)
    from promptcache_amd import CacheEngine

    class NoModelLM(H.TokOnlyLM):
        device = "cpu"
        use_full_position_ids = False

        def get_cache_shape(self):
            return 1, 1, 8

    for pr in c["prompts"]:
        eng = CacheEngine(64, NoModelLM(), target_device="cpu")
        eng.add_schema(fmt(c["schema"]), max_tokens=c["max_tokens"], no_cache=True)
        if "error" in pr:
            with pytest.raises(Exception) as ei:
                eng.process(pml.Prompt(pr["prompt"], [fmt]), no_cache=True)
            assert type(ei.value).__name__ == pr["error"]["type"] and str(ei.value) == pr["error"]["message"], (pr, ei.value)
            continue
        prompt = pml.Prompt(pr["prompt"], [fmt])
        assert prompt.text == pr["text"]
        ids, pos, _, _ = eng.process(prompt, no_cache=True)
        assert len(ids) == pr["nocache_n"] and list(pos) == list(range(len(ids)))
        assert crc(ids) == pr["nocache_ids_crc"] and crc(pos) == pr["nocache_pos_crc"]
