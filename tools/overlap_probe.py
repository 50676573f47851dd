"""Feasibility probe: how much of the KV gather (0.32 ms at the persona shape) hides under the forward's launches when it runs
on a side stream CONCURRENTLY with the hipGraph replay of the cached prefill?  (Timing only: the overlapped variant re-gathers
already staged segments, so its result is still correct.)   python tools/overlap_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import CacheEngine, Prompt, synth, _native  # noqa: E402
from promptcache_amd.model import Llama2  # noqa: E402

lm = Llama2("llama2-7b", device="cuda:0", random_init=True, seed=0)
eng = CacheEngine(4096, lm)
fmt = lm.get_formatter()
sp, pp = synth.persona_like()
eng.add_schema(fmt(sp))
prompt = Prompt(pp, [fmt])
pc = eng.prompt_cache
ids, pos, _, cache = eng.process(prompt)
it = torch.tensor([ids], device="cuda"); pt = torch.tensor([pos], device="cuda")
for _ in range(3):
    lm(input_ids=it, position_ids=pt, past_key_values=cache, use_cache=True)
torch.cuda.synchronize()
a = pc.arena
ptrs, lens, offs, off = [], [], [], 0
for m in pc.staged:
    ptrs.append(m.store.data_ptr()); lens.append(len(m)); offs.append(off); off += len(m)
main, side = torch.cuda.current_stream(), torch.cuda.Stream()


def seq():
    _native.kv_gather(ptrs, lens, offs, a.buf, a.L, a.Hkv, a.D, a.cap)
    lm(input_ids=it, position_ids=pt, past_key_values=cache, use_cache=True)


def ovl():
    side.wait_stream(main)
    _native.kv_gather(ptrs, lens, offs, a.buf, a.L, a.Hkv, a.D, a.cap, stream=side.cuda_stream)
    lm(input_ids=it, position_ids=pt, past_key_values=cache, use_cache=True)
    main.wait_stream(side)


def ovl_rev():
    side.wait_stream(main)
    lm(input_ids=it, position_ids=pt, past_key_values=cache, use_cache=True)       # the graph replay first ...
    _native.kv_gather(ptrs, lens, offs, a.buf, a.L, a.Hkv, a.D, a.cap, stream=side.cuda_stream)   # ... then the gather beside it
    main.wait_stream(side)


def only():
    lm(input_ids=it, position_ids=pt, past_key_values=cache, use_cache=True)


for name, fn in (("forward only", only), ("gather then forward", seq), ("gather || forward", ovl), ("forward || gather", ovl_rev), ("gather then forward", seq)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    print(f"{name:22s}: {best:.3f} ms", flush=True)
