"""Greedy decode after the cached prefill of the persona prompt (device-side loop, one hipGraph replay per token), for
rocprofv3 --kernel-trace: python tools/decode_profile.py [new_tokens]; tools/decode_stats.py turns the trace into per-token rows."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import CacheEngine, GenerationEngine, GenerationParameters, Prompt, synth  # noqa: E402
from promptcache_amd.model import Llama2  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
lm = Llama2("llama2-7b", device="cuda:0", random_init=True, seed=0)
eng = CacheEngine(4096, lm)
fmt = lm.get_formatter()
sp, pp = synth.persona_like()
eng.add_schema(fmt(sp))
prompt = Prompt(pp, [fmt])
gen = GenerationEngine(lm)
for rep in range(2):
    eng.prompt_cache.reset()
    ids, pos, _, cache = eng.process(prompt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 0
    for out in gen.generate(ids, pos, GenerationParameters(temperature=0.0, max_new_tokens=N, stop_token_ids=[]), cache):
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {n} tokens in {dt * 1e3:.1f} ms ({n / dt:.1f} tok/s incl. the prefill)", flush=True)
