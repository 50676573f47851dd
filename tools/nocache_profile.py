"""No-cache prefill of T random tokens (7b shape), 4 passes, for rocprofv3 --kernel-trace --stats.
python tools/nocache_profile.py T [model]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd.model import Llama2  # noqa: E402

T = int(sys.argv[1])
lm = Llama2(sys.argv[2] if len(sys.argv) > 2 else "llama2-7b", device="cuda:0", random_init=True, seed=0)
g = torch.Generator().manual_seed(0)
ids = torch.randint(3, 31000, (1, T), generator=g).to("cuda:0")
pos = torch.arange(T, device="cuda:0").view(1, T)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.inference_mode():
        out = lm(input_ids=ids, position_ids=pos, past_key_values=None, use_cache=True)
    torch.cuda.synchronize()
    print(f"T={T} pass {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
