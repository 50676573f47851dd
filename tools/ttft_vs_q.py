"""TTFT of the cached prefill as the number of new tokens q grows (7b shape, S = 1725 staged): shows where the
weight-streaming path (q <= 64) hands over to the dense path.  python tools/ttft_vs_q.py [q,q,...] [document tokens]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "prompt-cache_amd"))
from promptcache_amd import CacheEngine, Prompt, synth
from promptcache_amd.model import Llama2

lm = Llama2("llama2-7b", random_init=True)
fmt = lm.get_formatter()
qs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,12,16,17,24,32,33,48,64,65,96,128,192,256").split(",")]
for q in qs:
    sp, pp = synth.flat_docs(f"s{q}", 4, (int(sys.argv[2]) if len(sys.argv) > 2 else 1700,), max(q - 2, 1))
    eng = CacheEngine(4096, lm)
    eng.add_schema(fmt(sp))
    prompt = Prompt(pp, [fmt])
    ts = []
    for _ in range(7):
        eng.prompt_cache.reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ids, pos, _, cache = eng.process(prompt)
        lm(input_ids=torch.tensor([list(ids)], device=lm.device), position_ids=torch.tensor([pos], device=lm.device),
           past_key_values=cache, use_cache=True)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"q={len(ids):4d} S={cache[0][0].shape[1]:5d}  ttft {np.median(ts[2:]):.3f} ms", flush=True)
    del eng
