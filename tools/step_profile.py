"""Cached prefill of the persona prompt with a question of QL words (q = QL + 4 new tokens), 12 timed steps, for
rocprofv3 --kernel-trace --stats.  python tools/step_profile.py QL"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import CacheEngine, Prompt, synth  # noqa: E402
from promptcache_amd.model import Llama2  # noqa: E402

QL = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lm = Llama2("llama2-7b", device="cuda:0", random_init=True, seed=0)
eng = CacheEngine(4096, lm)
fmt = lm.get_formatter()
sp, pp = synth.persona_like(question_len=QL)
eng.add_schema(fmt(sp))
prompt = Prompt(pp, [fmt])
for i in range(14):
    eng.prompt_cache.reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ids, pos, _, cache = eng.process(prompt)
    o = lm(input_ids=torch.tensor([ids], device="cuda"), position_ids=torch.tensor([pos], device="cuda"), past_key_values=cache, use_cache=True)
    torch.cuda.synchronize()
    if i >= 11:
        print(f"q={len(ids)} ttft {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
