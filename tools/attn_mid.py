"""Attention of a mid-size question over a long staged cache (config 4's cached step: 13b, q = 259 over 8258 staged keys), event-timed.
python tools/attn_mid.py [H S q]        (PC_ATTN_NSPLIT forces the split count)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

H, S, q = (int(a) for a in (sys.argv[1:4] + ["40", "8258", "259"][len(sys.argv) - 1:]))
n.load()
dev = "cuda:0"
Hkv, D, L = H, 128, 4
cap = S + q + 64
arena = torch.randn((L, 2, Hkv, cap, D), device=dev).half()
q16 = torch.randn((q, H * D), device=dev).half()
q16l = (torch.randn((q, H * D), device=dev) * 2 ** -11).half()
lo = torch.zeros((2, Hkv, 320, D), device=dev).half()
mt = (q + 15) // 16
ah = torch.empty((mt, H * D // 32, 64, 8), dtype=torch.float16, device=dev)
al = torch.empty_like(ah)
ws = torch.empty(max(n.attn_workspace_bytes(1, H, D, q, S + q), 4) // 4 * 4, dtype=torch.float32, device=dev)
past_dev = torch.tensor([S, 0], dtype=torch.int32, device=dev)
kvlo = None if os.environ.get("NO_KVLO") else (lo[0], lo[1], Hkv * 320 * D, 320 * D, -1)


def run(i):
    li = i % L
    n.attn_fwd(q16, q * H * D, H * D, arena[li, 0], arena[li, 1], 2 * Hkv * cap * D, cap * D, None, 0, 0, 1, H, Hkv, D, q, S,
               1.0 / D ** 0.5, ws, past_len_dev=past_dev, out_frag=(ah, al), q_lo=q16l, kv_lo=kvlo)


for i in range(3):
    run(i)
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(16):
        run(i)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 16 * 1e3)
print(f"H={H} S={S} q={q} nsplit={os.environ.get('PC_ATTN_NSPLIT', 'auto')}: {best:.1f} us per launch pair", flush=True)
