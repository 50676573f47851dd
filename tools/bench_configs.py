"""TTFT for the BASELINE.json configs 2-4 at full model depth (context numbers for DESIGN.md; bench.py carries the
headline persona line).  Same recipe as the reference's eval (eval.py:200-215): cache_time + first-forward time, cached
vs no_cache, synthetic PML with the structure of each config (promptcache_amd/synth.py), random weights at true shapes.

    python tools/bench_configs.py [--configs 2,3,4]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "prompt-cache_amd"))
from promptcache_amd import CacheEngine, Prompt, synth  # noqa: E402
from promptcache_amd.model import Falcon, Llama2, Mpt  # noqa: E402


def ttft(lm, eng, prompt, no_cache, reps=5):
    ts = []
    for _ in range(reps):
        eng.prompt_cache.reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ids, pos, _, cache = eng.process(prompt, no_cache=no_cache, return_full_position_ids=lm.use_full_position_ids)
        lm(input_ids=torch.tensor([list(ids)], device=lm.device), position_ids=torch.tensor([pos], device=lm.device),
           past_key_values=cache, use_cache=True)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts[1:])), len(ids), (0 if cache is None else cache[0][0].shape[1])


CACHED_ONLY = False


def run(cfg):
    if cfg == 2:
        lm = Llama2("llama2-7b", random_init=True)
        cases = [("code_generation_game-like, max_ctx 5000", synth.flat_docs("game", 30, (306, 76, 800, 800, 800, 800, 800), 12), 5000)]
    elif cfg == 3:
        lm = Llama2("codellama-7b", random_init=True)
        rng = np.random.default_rng(0)
        cases = [(f"squad-like entry {i}", synth.flat_docs(f"squad{i}", 20, (int(rng.integers(100, 400)),), int(rng.integers(10, 30)), seed=i + 1), 1024)
                 for i in range(4)]
    elif cfg == 6:       # not a BASELINE config: the MPT adapter (ALiBi, full position ids)
        lm = Mpt("mpt-7b", random_init=True)
        cases = [("mpt-7b, persona-structured schema", synth.persona_like(), 4096)]
    elif cfg == 5:       # not a BASELINE config: the Falcon adapter (multi-query KV: 8 KiB per token instead of 512 KiB)
        lm = Falcon("falcon-7b", random_init=True)
        cases = [("falcon-7b, persona-structured schema", synth.persona_like(), 4096),
                 ("falcon-7b, game-like schema", synth.flat_docs("game", 30, (306, 76, 800, 800, 800, 800, 800), 12), 5000)]
    else:
        lm = Llama2("llama2-13b", random_init=True)
        cases = [("longbench-like 8k context", synth.flat_docs("longbench", 10, (8000,), 255), 9186)]
    fmt = lm.get_formatter()
    for name, (sp, pp), max_ctx in cases:
        eng = CacheEngine(max_ctx, lm)
        t0 = time.perf_counter(); eng.add_schema(fmt(sp)); torch.cuda.synchronize(); enc = time.perf_counter() - t0
        prompt = Prompt(pp, [fmt])
        c, q, S = ttft(lm, eng, prompt, False)
        n, nq, _ = ttft(lm, eng, prompt, True) if not CACHED_ONLY else (float("nan"), 0, 0)
        print(json.dumps({"config": cfg, "model": lm.config.name, "case": name, "staged_tokens": S, "new_tokens": q,
                          "ttft_cached_ms": round(c, 3), "ttft_no_cache_ms": round(n, 3), "speedup": round(n / c, 2),
                          "cached_prefill_tokens_per_s": round((S + q) / c * 1e3), "encode_s": round(enc, 3)}), flush=True)
        del eng
    del lm
    torch.cuda.empty_cache()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3,4")
    ap.add_argument("--cached-only", action="store_true", help="skip the no_cache leg (for profiling the cached path)")
    a = ap.parse_args()
    CACHED_ONLY = a.cached_only
    for c in [int(x) for x in a.configs.split(",")]:
        run(c)
