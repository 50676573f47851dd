"""Seconds of ONE schema-encode forward (many_rows, kv_only: what SchemaCache._process launches) by its row count, 1 GPU, true
layer shape -- the curve `bench.py --plan-only` prices every rank's forwards on (a rank of an 8-GPU encode
runs forwards of a few hundred rows, far below the large-M rate).  Two shapes of forward: a single scaffold of `rows` tokens
(B = 1: trunks, whole scaffolds) and a ragged suffix batch over a 300-row shared prefix (B = rows / 192 batch rows of 192 tokens
each: the suffix groups).  Writes profiles-style JSON to stdout.   python tools/encode_rate_curve.py [model]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd.model import Llama2  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "llama2-7b"
lm = Llama2(model, device="cuda:0", random_init=True, seed=0)
m = lm.hf_model
g = torch.Generator().manual_seed(1)


def best(fn, reps=4):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts[1:])


points, batched = [], []
for rows in (48, 96, 160, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192):
    ids = torch.randint(3, 32000, (1, rows), generator=g).cuda()
    pos = torch.arange(rows).unsqueeze(0).cuda()
    t = best(lambda: m(input_ids=ids, position_ids=pos, use_cache=True, many_rows=True, kv_only=True))
    points.append({"rows": rows, "seconds": t, "tokens_per_s": rows / t})
    print(f"B=1 rows={rows}: {t * 1e3:.2f} ms  {rows / t:.0f} tok/s", file=sys.stderr, flush=True)
# ragged suffix batches over a shared prefix (the trunk read in place)
trunk_ids = torch.randint(3, 32000, (1, 300), generator=g).cuda()
trunk = m(input_ids=trunk_ids, position_ids=torch.arange(300).unsqueeze(0).cuda(), use_cache=True, many_rows=True, kv_only=True).past_key_values.arena
for B in (1, 2, 4, 8, 16, 32):
    ids = torch.randint(3, 32000, (B, 192), generator=g).cuda()
    pos = (300 + torch.arange(192)).unsqueeze(0).expand(B, 192).contiguous().cuda()
    pre = [300 - 3 * (b % 5) for b in range(B)]
    t = best(lambda: m(input_ids=ids, position_ids=pos, use_cache=True, many_rows=True, kv_only=True, shared_prefix=(trunk, pre)))
    batched.append({"rows": B * 192, "batch_rows": B, "seconds": t, "tokens_per_s": B * 192 / t})
    print(f"suffix batch B={B} x 192: {t * 1e3:.2f} ms  {B * 192 / t:.0f} tok/s", file=sys.stderr, flush=True)
print(json.dumps({"model": model, "points": points, "suffix_batches_over_a_300_row_prefix": batched,
                  "how": "best of 3 warm calls of LlamaHIP.__call__(many_rows=True, kv_only=True) per row count (host launch time "
                         "included: that is what a rank pays per forward); tools/encode_rate_curve.py"}))
