"""How much of a cached-prefill step is the hipGraph launch itself?  One replay of the captured forward against two replays
enqueued back to back (the second one's submission hides under the first one's execution): t(2) - t(1) = the forward's GPU time,
2 t(1) - t(2) = what a lone replay exposes in front of it.  python tools/replay_gap.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import CacheEngine, Prompt, synth  # noqa: E402
from promptcache_amd.model import Llama2  # noqa: E402

lm = Llama2("llama2-7b", device="cuda:0", random_init=True, seed=0)
eng = CacheEngine(4096, lm)
fmt = lm.get_formatter()
sp, pp = synth.persona_like()
eng.add_schema(fmt(sp))
prompt = Prompt(pp, [fmt])
for _ in range(4):
    eng.prompt_cache.reset()
    ids, pos, _, cache = eng.process(prompt)
    lm(input_ids=torch.tensor([ids]), position_ids=torch.tensor([pos]), past_key_values=cache, use_cache=True)
torch.cuda.synchronize()
m = lm.hf_model
ent = list(m._graphs.values())[-1]
g = ent[0]
res = {}
for n in (1, 2, 3):
    ts = []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    res[n] = float(np.median(ts[5:]))
    print(f"{n} replay(s): {res[n]:.1f} us")
print(f"GPU time of one forward {res[2] - res[1]:.1f} us (third: {res[3] - res[2]:.1f}); exposed in front of a lone replay {2 * res[1] - res[2]:.1f} us")
