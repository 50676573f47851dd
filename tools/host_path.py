"""Where the HOST time of one timed step goes (request assembly + the call into the captured forward): cProfile over 200 steps of
bench.py's step body, plus wall-clock splits.  python tools/host_path.py"""
import cProfile
import os
import pstats
import sys
import time

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
pc = eng.prompt_cache


def step():
    pc.reset()
    t0 = time.perf_counter()
    ids, pos, _, cache = eng.process(prompt)
    t1 = time.perf_counter()
    out = lm(input_ids=torch.tensor([ids]), position_ids=torch.tensor([pos]), past_key_values=cache, use_cache=True)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2


for _ in range(10):
    step()
ts = [step() for _ in range(200)]
med = lambda k: sorted(t[k] for t in ts)[len(ts) // 2] * 1e6   # noqa: E731
print(f"median us: process {med(0):.1f} | lm() call returns after {med(1):.1f} | wait for the GPU {med(2):.1f} | total {med(0) + med(1) + med(2):.1f}")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
