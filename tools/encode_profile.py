"""Schema encode of the bench's persona-structured schema, twice (warm), for rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import CacheEngine, synth  # noqa: E402
from promptcache_amd.model import Llama2  # noqa: E402

lm = Llama2("llama2-7b", device="cuda:0", random_init=True, seed=0)
eng = CacheEngine(4096, lm)
text = lm.get_formatter()(synth.persona_like()[0])
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.add_schema(text)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = eng.schemas["persona"].encode_stats
    print(f"encode {i}: {dt * 1e3:.1f} ms  {st}", flush=True)
    if i < 2:
        eng.remove_schema("persona")
