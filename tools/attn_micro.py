"""Cached-prefill attention launch in a loop (persona shape by default) for rocprofv3 --kernel-trace --stats.
python tools/attn_micro.py [S q tail reps fused]     (fused = 1: pc_attn with arrival counters, the single-launch merge)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

S, q, tail, reps, fused = (int(a) for a in (sys.argv[1:6] + ["1725", "12", "1", "300", "1"][len(sys.argv) - 1:]))
n.load()
dev = "cuda:0"
H = Hkv = 32
D = 128
L = 8
cap = max(4096, S + q + 64)
arena = torch.randn((L, 2, Hkv, cap, D), device=dev).half()
q16 = torch.randn((q, H * D), device=dev).half()
q16l = (torch.randn((q, H * D), device=dev) * 2 ** -11).half()
lo = torch.zeros((2, Hkv, 320, D), device=dev).half()
mt = (q + 15) // 16
ah = torch.empty((mt, H * D // 32, 64, 8), dtype=torch.float16, device=dev)
al = torch.empty_like(ah)
ws = torch.empty(max(n.attn_workspace_bytes(1, H, D, q, S + q), 4) // 4, dtype=torch.float32, device=dev)
ctr = torch.zeros(H, dtype=torch.int32, device=dev) if fused else None
past_dev = torch.tensor([S, 0], dtype=torch.int32, device=dev)
kvlo = (lo[0], lo[1], Hkv * 320 * D, 320 * D, -1) if tail else None


def step(i):
    li = i % L
    n.attn_fwd(q16, q * H * D, H * D, arena[li, 0], arena[li, 1], 2 * Hkv * cap * D, cap * D, None, 0, 0, 1, H, Hkv, D, q, S,
               1.0 / D ** 0.5, ws, past_len_dev=past_dev, out_frag=(ah, al), q_lo=q16l, kv_lo=kvlo, counters=ctr)


for i in range(10):
    step(i)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(reps):
        step(i)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(5):
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / reps * 1e3)
kv_bytes = 2 * Hkv * (S + q) * D * 2
print(f"attn S={S} q={q} tail={tail} fused={fused} nstream={os.environ.get('PC_ATTN_SMALL_WG', '-')}: {best:.2f} us per layer-launch "
      f"({kv_bytes / best / 1e3:.0f} GB/s of K/V)", flush=True)
