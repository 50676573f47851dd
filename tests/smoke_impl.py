"""`__graft_entry__.smoke()` body: one small pass of the whole hot path on cuda:0 (schema encode ->
gather -> cached prefill -> greedy decode through the HIP kernels), checked against the numpy oracle."""
import numpy as np
import torch

from oracle import engine_oracle as eo
from oracle.llama_oracle import LlamaOracle, OracleConfig
from tests import helpers as H


def run():
    from promptcache_amd import CacheEngine, GenerationEngine, GenerationParameters, Prompt
    from promptcache_amd.model import Llama2
    from promptcache_amd.model.config import SHAPES
    from promptcache_amd.model.weights import make_weights_np
    g = H.load_case("tiny_trip")
    shape = SHAPES["tiny"]
    w16 = make_weights_np(shape, 0, 4.0)
    lm = Llama2(name="smoke", shape=shape, weights=w16, device="cuda:0")
    eng = CacheEngine(256, lm)
    eng.add_schema(lm.get_formatter()(str(g["schema_text"])))
    prompt = Prompt(str(g["prompt_text"]), [lm.get_formatter()])
    ids, pos, cache_ms, cache = eng.process(prompt)
    out = lm(input_ids=torch.tensor([ids], device="cuda"), position_ids=torch.tensor([pos], device="cuda"),
             past_key_values=cache, use_cache=True)
    shape_, schema, jobs, _, used, ids_o, pos_o = H.layout_for_case(g)
    model, _ = H.oracle_for_case(g, shape)
    lib = eo.encode_schema(model, jobs)
    _, S, (logits, present) = eo.cached_prefill(model, lib, used, ids_o, pos_o, 256)
    err = float(np.abs(out.logits[0].cpu().numpy() - logits[0]).max())
    assert ids == ids_o and pos == pos_o and err < 1e-2, f"smoke parity failed: max|dlogit|={err}"
    ids2, pos2, _, cache2 = eng.process(prompt)
    params = GenerationParameters(temperature=0.0, max_new_tokens=3, stop_token_ids=[], stop_str=[])
    outs = list(GenerationEngine(lm).generate(ids2, pos2, params, cache2, stream_interval=1))
    toks = eo.generate_greedy(model, logits, present, pos_o, 3)
    assert outs[-1].new_text == lm.decode(toks), (outs[-1].new_text, toks)
    print(f"smoke ok: S={S} q={len(ids)} gather {cache_ms:.3f} ms  max|dlogit| vs oracle {err:.2e}  greedy {toks}")
