"""attention (defer_merge) + pc_gemm_part_rows against attention + merge launch + o_proj at the persona shape (7b: 32 heads, 1725 staged
keys, 12 live rows in a 16-row graph bucket): event-timed pairs, K slices 2 / 4 / 8.   python tools/part_rows_micro.py [H S q_len]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

H, S, q_len = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 1725, 16)
D, dev = 128, "cuda:0"
n.load()
K = N = H * D
L = 8                                                                     # distinct weights / caches so that nothing stays in L2 / MALL
q = torch.randn((q_len, K), device=dev).half(); ql = (torch.randn((q_len, K), device=dev) * 1e-3).half()
cap = S + q_len + 8
kvs = [torch.randn((1, 2, H, cap, D), device=dev).half() for _ in range(L)]
wfs = [n.to_weight_frags((0.03 * torch.randn((N, K), device=dev)).half()) for _ in range(L)]
ws = torch.empty(n.attn_workspace_bytes(1, H, D, q_len, S + q_len) // 4, dtype=torch.float32, device=dev)
ah, al = (torch.zeros((1, K // 32, 64, 8), dtype=torch.float16, device=dev) for _ in range(2))
y = torch.zeros((q_len, N), dtype=torch.float32, device=dev)
sc = torch.empty(n.gemm_skinny_ks_scratch_bytes(N, 8) // 4, dtype=torch.float32, device=dev)
ctr = torch.zeros(N // 16, dtype=torch.int32, device=dev)
rows_dev = torch.tensor([12 if q_len == 16 else q_len], dtype=torch.int32, device=dev)


def attn(i, defer):
    kv = kvs[i % L]
    return n.attn_fwd(q, q_len * K, K, kv[:, 0], kv[:, 1], 2 * H * cap * D, cap * D, None, 0, 0, 1, H, H, D, q_len, S, 1.0 / np.sqrt(D), ws,
                      out_frag=(ah, al), q_lo=ql, defer_merge=defer)


def three(i):
    attn(i, False)
    n.gemm_skinny(wfs[i % L], ah, al, q_len, N, K, n.EPI_ADD, y=y, ldy=N, rows_dev=rows_dev)


def part(ksl):
    def f(i):
        ns = attn(i, True)
        n.gemm_part_rows(wfs[i % L], ws, ws[H * ns * q_len * D:], ns, H, D, N, q_len, y, N, ksl, sc, ctr, rows_dev=rows_dev)
    return f


def timed(f, reps=64):
    for i in range(8):
        f(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            f(i)
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


print(f"H={H} S={S} q_len={q_len}: attention + merge + o_proj (three launches, graph replay) {timed(three):.2f} us per layer", flush=True)
for ksl in (8, 4, 2):
    print(f"   attention (partials) + pc_gemm_part_rows, {ksl} K slices: {timed(part(ksl)):.2f} us", flush=True)
