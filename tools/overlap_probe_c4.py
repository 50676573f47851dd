"""Feasibility probe for BASELINE config 4's shape (13b, 8.3 k staged tokens, 259 new rows: the forward runs at a quarter of the
HBM roof): does the KV gather (2.5 ms) hide under the forward when it runs on a side stream?  (Timing only: the overlapped
variant re-gathers already staged segments.)   python tools/overlap_probe_c4.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
import bench  # noqa: E402
from promptcache_amd import CacheEngine, Prompt, _native  # noqa: E402
from promptcache_amd.model import Llama2  # noqa: E402

name, max_ctx, max_tokens, entries, label = bench.config_workload(4)
lm = Llama2(name, device="cuda:0", random_init=True, seed=0)
eng = CacheEngine(max_ctx, lm)
fmt = lm.get_formatter()
eng.add_schema(fmt(entries[0][0]), max_tokens=max_tokens)
prompt = Prompt(entries[0][1], [fmt])
pc = eng.prompt_cache
ids, pos, _, cache = eng.process(prompt)
it = torch.tensor([list(ids)], device="cuda"); pt = torch.tensor([pos], device="cuda")
print("q", len(ids), "S", len(pc), flush=True)
a = pc.arena
ptrs, lens, offs, off = [], [], [], 0
for m in pc.staged:
    ptrs.append(m.store.data_ptr()); lens.append(len(m)); offs.append(off); off += len(m)
main, side = torch.cuda.current_stream(), torch.cuda.Stream()
S = len(pc)


def fwd():
    a.length = S
    a.tail_base, a.tail_len = -1, 0
    lm(input_ids=it, position_ids=pt, past_key_values=pc.cache, use_cache=True)


def seq():
    _native.kv_gather(ptrs, lens, offs, a.buf, a.L, a.Hkv, a.D, a.cap)
    fwd()


def ovl():
    side.wait_stream(main)
    _native.kv_gather(ptrs, lens, offs, a.buf, a.L, a.Hkv, a.D, a.cap, stream=side.cuda_stream)
    fwd()
    main.wait_stream(side)


for nm, fn in (("forward only", fwd), ("gather then forward", seq), ("gather || forward", ovl), ("gather then forward", seq), ("gather || forward", ovl)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    print(f"{nm:22s}: {best:.3f} ms", flush=True)
