"""In-graph time of the N = hidden projections with their residual add at M rows (7b shapes): the one-tile-per-workgroup launch
(gemm_skinny EPI_ADD) against pc_gemm_skinny_ks (K split across workgroups, reduction inside the launch) over (tiles, slices).
python tools/ks_micro.py [M]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12


def timeit(fn, iters=128, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


for N, K in ((4096, 4096), (4096, 11008)):
    ws = [n.to_weight_frags(torch.randn(N, K, device=DEV).half() * 0.05) for _ in range(24)]
    hi, lo = n.to_act_frags(torch.randn(M, K, device=DEV))
    y = torch.zeros((M, N), dtype=torch.float32, device=DEV)
    i = [0]

    def base():
        i[0] = (i[0] + 1) % len(ws)
        n.gemm_skinny(ws[i[0]], hi, lo, M, N, K, n.EPI_ADD, y=y, ldy=N)
    t = timeit(base)
    print(f"M={M} K={K} EPI_ADD one tile per workgroup: {t:.2f} us  {N * K * 2 / t / 1e3:.0f} GB/s", flush=True)
    for T, S in ((1, 2), (2, 2), (2, 4), (4, 4), (4, 8), (8, 8), (1, 1), (2, 1)):
        scratch = torch.empty(n.gemm_skinny_ks_scratch_bytes(N, S) // 4, dtype=torch.float32, device=DEV)
        ctr = torch.zeros(N // 16, dtype=torch.int32, device=DEV)

        def ks():
            i[0] = (i[0] + 1) % len(ws)
            n.gemm_skinny_ks(ws[i[0]], hi, lo, M, N, K, y, N, S, T, scratch, ctr)
        t = timeit(ks)
        print(f"M={M} K={K} ks T={T} S={S} ({N // 16 // T * S} wgs): {t:.2f} us  {N * K * 2 / t / 1e3:.0f} GB/s", flush=True)
