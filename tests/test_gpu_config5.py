"""BASELINE config 5's workload -- the schema LIBRARY encode (SURVEY section 8d row 5 over promptcache/cache_engine.py:185-308;
the reference loops ``add_schema`` over its schema files, eval.py:172-181) -- at the 7b layer shape, one rank:

* the bench's own library (``bench.py`` ``encode_library``: five persona-structured + three flat document schemas, 148 encode
  passes, ~97 k scaffold tokens, ~30 k cached tokens) goes through ``CacheEngine.add_schemas``; every stored segment must equal,
  bit for bit, what a per-schema ``add_schema`` encode stores (the library call only adds a schedule on top);
* a persona segment (a non-default union member: encoded as a SUFFIX over the schema's trunk, packed into a ragged batch) and a
  flat-document segment (encoded inside one long whole-scaffold pass) are compared with ``engine_oracle.encode_schema`` -- the
  reference's per-scaffold encode in fp32 on the host, rounded to fp16 as ``cache_engine.py:283-296`` stores it.

Four layers of the 7b shape: the oracle finishes in seconds; depth is the subject of tests/test_gpu_fullsize.py.
"""
import dataclasses

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _library_texts():
    """The same call sequence as bench.py's ``encode_library`` leg and ``bench.py --plan-only``."""
    from promptcache_amd import synth
    texts = [synth.persona_like(name=f"lib-persona-{i}", system_len=200 + 40 * i, seed=20 + i)[0] for i in range(5)]
    texts += [synth.flat_docs(f"lib-docs-{i}", 30, lens, 8, seed=30 + i)[0]
              for i, lens in enumerate([(306, 76, 800, 800, 800), (1500, 1200), (400,) * 6])]
    return texts


def _stores(engine, name):
    sc = engine.schemas[name]
    return sorted(((c.token_sequence.offset, len(c), c.store) for c in sc.cache_l1.values()), key=lambda t: (t[0], t[1]))


def test_config5_library_7b_shape():
    from oracle import engine_oracle as eo
    from oracle.llama_oracle import LlamaOracle, OracleConfig
    from promptcache_amd import CacheEngine
    from promptcache_amd.model import Llama2
    from promptcache_amd.model.config import SHAPES
    from promptcache_amd.model.weights import random_weights_device
    shape = dataclasses.replace(SHAPES["llama2-7b"], num_hidden_layers=4, vocab_size=8192)
    w = random_weights_device(shape, "cuda:0", torch.float16, seed=31)
    lm = Llama2(name="lib7b", shape=shape, weights=w, device="cuda:0")
    fmt = lm.get_formatter()
    texts = [fmt(t) for t in _library_texts()]

    lib = CacheEngine(4096, lm)
    lib.add_schemas(texts)
    names = list(lib.schemas)
    assert len(names) == 8
    passes = sum(lib.schemas[n].encode_stats["total_passes"] for n in names)
    scaffold_tokens = sum(len(j["token_ids"]) for n in names for j in lib.schemas[n]._plan())
    cached = sum(lib.schemas[n].encode_stats["cached_tokens"] for n in names)
    computed = sum(lib.schemas[n].encode_stats["computed_tokens"] for n in names)
    print(f"[config 5] 8 schemas, {passes} passes, {scaffold_tokens} scaffold tokens, {computed} computed, {cached} cached")
    assert passes > 120 and scaffold_tokens > 80_000 and cached > 25_000          # the workload BASELINE config 5 names

    # (a) the library call stores what per-schema add_schema calls store, bit for bit
    solo = CacheEngine(4096, lm)
    n_seg = 0
    for text, name in zip(texts, names):
        solo.add_schema(text)
        a, b = _stores(lib, name), _stores(solo, name)
        assert [(o, n) for o, n, _ in a] == [(o, n) for o, n, _ in b]
        for (_, _, x), (_, _, y) in zip(a, b):
            assert torch.equal(x.view(torch.int16), y.view(torch.int16))
        n_seg += len(a)
        solo.remove_schema(name)                                   # (the library stays; one solo schema resident at a time)
    assert n_seg == sum(len(lib.schemas[n].cache_l1) for n in names)

    # (b) two segments against the reference's encode, restated on the host
    cfg = OracleConfig(vocab_size=shape.vocab_size, hidden_size=shape.hidden_size, intermediate_size=shape.intermediate_size,
                       num_hidden_layers=shape.num_hidden_layers, num_attention_heads=shape.num_attention_heads,
                       num_key_value_heads=shape.num_key_value_heads, rms_norm_eps=shape.rms_norm_eps,
                       rope_theta=shape.rope_theta, inv_freq=lm.hf_model.inv_freq_cpu.numpy())
    oracle = LlamaOracle(cfg, {k: v.float().cpu().numpy() for k, v in w.items()})

    def check(schema_name, want):
        """``want(path string, TokenSequence)`` picks the segment; the oracle runs that segment's scaffold up to the segment's last
        token (under the causal mask nothing behind a token reaches its K / V, cache_engine.py:243-296)."""
        sc = lib.get_schema(schema_name)
        for p in sc.encode_paths():
            sf = sc.get_scaffold(p)
            for tc in sf.select(p).all_token_sequences():
                if want(str(p), tc):
                    pos = list(sf.position_ids())
                    end = pos.index(tc.offset) + len(tc)
                    job = dict(token_ids=sf.token_ids()[:end], position_ids=pos[:end], targets=[tc])
                    ref = eo.encode_schema(oracle, [job])[id(tc)]
                    store = lib.schemas[schema_name].cache_l1[id(tc)].store            # [L][2][Hkv][len][D] fp16
                    worst, top = 0.0, 0.0
                    for li, (k, v) in enumerate(ref):
                        for plane, r in ((0, k), (1, v)):
                            r16 = r.astype(np.float16).astype(np.float32)            # stored rounded to fp16 (cache_engine.py:283-296)
                            got = store[li, plane].float().cpu().numpy()
                            worst = max(worst, float(np.abs(got - r16).max()))
                            top = max(top, float(np.abs(r16).max()))
                    print(f"[config 5] {schema_name} path '{p}' segment at {tc.offset} (+{len(tc)}), scaffold rows run by the oracle "
                          f"{end}: max|dKV| {worst:.2e} (max|KV| {top:.2f})")
                    return worst, top, end
        raise AssertionError("no such segment")

    # persona: a member of the FOURTH union (its scaffold = system + preamble + the member, positions with a gap in between; the
    # product encodes it as a suffix over the trunk, packed into a ragged batch)
    w1, t1, rows1 = check("lib-persona-0", lambda path, tc: "occupation-3" in path and len(tc) > 100)
    # documents: the last long module of a union-free schema (one whole-scaffold pass of ~2.7 k rows)
    w2, t2, rows2 = check("lib-docs-1", lambda path, tc: len(tc) > 1000 and tc.offset > 1000)
    assert rows1 > 500 and rows2 > 2500
    # fp16 stores of values of O(1): half an fp16 ulp at the top of the range is 2^-11 * max|KV|; allow two ulps of drift
    assert w1 <= 2.0 ** -9 * max(t1, 1.0) and w2 <= 2.0 ** -9 * max(t2, 1.0), (w1, t1, w2, t2)
