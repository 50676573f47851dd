"""Schema-encode throughput probe (persona-structured schema at the 7b shape): wall time per add_schema call with
and without trunk reuse (SchemaCache.share_trunk), after a warm-up call."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "prompt-cache_amd"))
from promptcache_amd import CacheEngine, synth
from promptcache_amd.cache_engine import SchemaCache
from promptcache_amd.model import Llama2
lm = Llama2("llama2-7b", device="cuda:0", random_init=True, seed=0)
fmt = lm.get_formatter()
sp, pp = synth.persona_like()
eng = CacheEngine(4096, lm)
for share in (True, True, False, False, True):
    SchemaCache.share_trunk = share
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.add_schema(fmt(sp))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = eng.schemas["persona"].encode_stats
    print(f"share_trunk={share}: {dt*1e3:.1f} ms  {st['encoded_tokens']/dt:.0f} scaffold tok/s  passes={st['passes']} "
          f"shared={st['trunk_shared_passes']} computed_tokens={st['computed_tokens']} of {st['encoded_tokens']}")
    eng.remove_schema("persona")
