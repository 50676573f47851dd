"""Shared helpers for the parity tests: golden loading, layout via the product's PML front-end,
oracle construction on the golden's inputs."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODEL_CASES = ["tiny_trip", "tiny_trip2", "mid_trip", "mid_mha_doc", "tiny_personalike"]
FALCON_CASES = ["falcon_tiny_trip", "falcon_mid_doc"]     # reference Falcon adapter: multi-query cache (L, 1, D)
MPT_CASES = ["mpt_tiny_trip", "mpt_mid_doc"]              # reference Mpt adapter: ALiBi at the keys' position ids


def is_mpt(g) -> bool:
    return str(g["shape_name"]).startswith("mpt")


def full_positions(g, used, pos):
    """Position ids handed to the model: MPT wants one per KEY (staged segments in staging order, then the new tokens,
    cache_engine.py:517-519); everyone else the new tokens only."""
    if not is_mpt(g):
        return list(pos)
    return [p for u in used for p in u.position_ids()] + list(pos)


def is_falcon(g) -> bool:
    return str(g["shape_name"]).startswith("falcon")


def shape_for_case(g):
    from promptcache_amd.model.config import FALCON_SHAPES, MPT_SHAPES, SHAPES
    name = str(g["shape_name"])
    return (FALCON_SHAPES if name.startswith("falcon") else MPT_SHAPES if name.startswith("mpt") else SHAPES)[name]


def load_case(name):
    z = np.load(os.path.join(GOLD, f"model_{name}.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


class TokOnlyLM:
    """encode-only adapter: enough to lay out a schema / assemble a prompt on the CPU."""

    def __init__(self, vocab=32000):
        from promptcache_amd.model.tokenizer import StandInTokenizer
        self.hf_tokenizer = StandInTokenizer(vocab)
        self.unk_token_id = 0
        self.eos_token_id = 2

    def encode(self, text):
        return self.hf_tokenizer.encode(text)


def llama_formatter():
    from promptcache_amd.model import _llama_formatter
    return _llama_formatter()


def formatter_for_case(g):
    from promptcache_amd.model import _falcon_formatter, _mpt_formatter
    return _falcon_formatter() if is_falcon(g) else _mpt_formatter() if is_mpt(g) else llama_formatter()


def assemble(schema, prompt, lm):
    """Integer part of CacheEngine.process: (used TokenSequence objects, new ids, new positions)."""
    used, ids, pos = [], [], []
    stack = [(prompt, schema)]
    while stack:
        ref, module = stack.pop()
        used += list(module.token_sequences())
        for arg in ref.args:
            prm = [p for p in module.parameters() if p.name == arg.name][0]
            a = lm.encode(arg.value)
            ids += a
            pos += prm.position_ids()[:len(a)]
        for m in ref.modules:
            stack.append((m, module.select(m.name)))
    if len(prompt.text) > 0:
        t = lm.encode(prompt.text)
        ids += t
        pos += list(range(len(schema), len(schema) + len(t)))
    return used, ids, pos


def layout_for_case(g):
    """Schema, encode jobs (reference path order), prompt assembly for a model golden."""
    from promptcache_amd.pml import Prompt, Schema
    shape = shape_for_case(g)
    lm = TokOnlyLM(shape.vocab_size)
    fmt = formatter_for_case(g)
    mt = int(g["max_tokens"])
    schema = Schema(fmt(str(g["schema_text"])), lm, max_tokens=None if mt < 0 else mt)
    jobs = []
    for p in schema.encode_paths():
        sf = schema.get_scaffold(p)
        jobs.append(dict(path=str(p), token_ids=sf.token_ids(), position_ids=sf.position_ids(),
                         targets=sf.select(p).all_token_sequences()))
    prompt = Prompt(str(g["prompt_text"]), [fmt])
    used, ids, pos = assemble(schema, prompt, lm)
    return shape, schema, jobs, prompt, used, ids, pos


def oracle_for_case(g, shape):
    from oracle.llama_oracle import LlamaOracle, OracleConfig
    from promptcache_amd.model.weights import make_falcon_weights_np, make_mpt_weights_np, make_weights_np
    if is_mpt(g):
        from oracle.mpt_oracle import MptOracle, MptOracleConfig
        w16 = make_mpt_weights_np(shape, int(g["seed"]), float(g["scale"]))
        cfg = MptOracleConfig(vocab_size=shape.vocab_size, hidden_size=shape.hidden_size,
                              num_hidden_layers=shape.num_hidden_layers, num_attention_heads=shape.num_attention_heads,
                              layer_norm_epsilon=shape.layer_norm_epsilon, alibi_bias_max=shape.alibi_bias_max)
        return MptOracle(cfg, {k: v.astype(np.float32) for k, v in w16.items()}), w16
    if is_falcon(g):
        from oracle.falcon_oracle import FalconOracle, FalconOracleConfig
        w16 = make_falcon_weights_np(shape, int(g["seed"]), float(g["scale"]))
        cfg = FalconOracleConfig(vocab_size=shape.vocab_size, hidden_size=shape.hidden_size,
                                 num_hidden_layers=shape.num_hidden_layers, num_attention_heads=shape.num_attention_heads,
                                 layer_norm_epsilon=shape.layer_norm_epsilon, rope_theta=shape.rope_theta, inv_freq=g["inv_freq"])
        return FalconOracle(cfg, {k: v.astype(np.float32) for k, v in w16.items()}), w16
    w16 = make_weights_np(shape, int(g["seed"]), float(g["scale"]))
    cfg = OracleConfig(vocab_size=shape.vocab_size, hidden_size=shape.hidden_size,
                       intermediate_size=shape.intermediate_size, num_hidden_layers=shape.num_hidden_layers,
                       num_attention_heads=shape.num_attention_heads, num_key_value_heads=shape.num_key_value_heads,
                       rms_norm_eps=shape.rms_norm_eps, rope_theta=shape.rope_theta, inv_freq=g["inv_freq"])
    return LlamaOracle(cfg, {k: v.astype(np.float32) for k, v in w16.items()}), w16


def oracle_blas():
    """Context manager: the BLAS pool at the width the host oracle runs fastest at.  numpy's OpenBLAS defaults to 64 threads on
    the 256-CPU GPU boxes, where an sgemm of a few hundred rows runs 4-5x SLOWER than on 16 (tools/blas_probe.py: 0.54 vs 2.6
    TFLOP/s at 130 rows) -- the full-depth parity tests spend their time exactly there."""
    import os
    from threadpoolctl import threadpool_limits
    return threadpool_limits(limits=min(16, os.cpu_count() or 16), user_api="blas")
