"""Wall time of the BASELINE config-5 stand-in (8-schema library through CacheEngine.add_schemas) on one GPU, three times in a row.
python tools/library_encode_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
import __graft_entry__ as ge  # noqa: E402
ge.build()
from promptcache_amd import CacheEngine, synth  # noqa: E402
from promptcache_amd.model import Llama2  # noqa: E402

lm = Llama2("llama2-7b", device="cuda:0", random_init=True, seed=0)
eng = CacheEngine(4096, lm)
fmt = lm.get_formatter()
lib = [synth.persona_like(name=f"lib-persona-{i}", system_len=200 + 40 * i, seed=20 + i)[0] for i in range(5)]
lib += [synth.flat_docs(f"lib-docs-{i}", 30, lens, 8, seed=30 + i)[0] for i, lens in enumerate([(306, 76, 800, 800, 800), (1500, 1200), (400,) * 6])]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.add_schemas([fmt(t) for t in lib])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    names = [n for n in eng.schemas if n.startswith("lib-")]
    per = []
    print(f"rep {rep}: {dt:.3f} s  computed {sum(eng.schemas[n].encode_stats['computed_tokens'] for n in names)}", flush=True)
    for n in names:
        eng.remove_schema(n)
# one schema at a time
for t in lib:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.add_schema(fmt(t))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nm = list(eng.schemas)[-1]
    print(f"  {nm}: {dt * 1e3:.1f} ms  {eng.schemas[nm].encode_stats}", flush=True)
