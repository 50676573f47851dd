"""How much faster does a weight-streaming launch run when its weights are already on the die (Infinity Cache / L2)?
In-graph time per launch of the 7b-shape projections with the launches cycling over `copies` weight images:
1 copy = every launch re-reads the image the previous launch just read; 24 copies = every launch streams from HBM.
python tools/mall_probe.py [M]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12


def timeit(fn, iters=96, warm=6):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


for name, N, K, epi in (("o_proj", 4096, 4096, 1), ("down", 4096, 11008, 1), ("qkv(store)", 12288, 4096, 0),
                        ("gate|up", 22016, 4096, 2)):
    for copies in (1, 2, 4, 24):
        ws = [n.to_weight_frags(torch.randn(N, K, device=DEV).half() * 0.05) for _ in range(copies)]
        hi, lo = n.to_act_frags(torch.randn(M, K, device=DEV))
        y = torch.zeros((M, N), dtype=torch.float32, device=DEV)
        KSo = (N // 2 + 31) // 32
        oh = torch.zeros(((M + 15) // 16) * KSo * 512, dtype=torch.float16, device=DEV)
        ol = torch.zeros_like(oh)
        i = [0]

        def fn():
            i[0] = (i[0] + 1) % len(ws)
            if epi == 2:
                n.gemm_skinny(ws[i[0]], hi, lo, M, N, K, epi, of_hi=oh, of_lo=ol)
            else:
                n.gemm_skinny(ws[i[0]], hi, lo, M, N, K, epi, y=y, ldy=N)
        t = timeit(fn)
        print(f"{name:11s} M={M} copies={copies:2d} ({copies * N * K * 2 / 1e6:7.1f} MB live): {t:6.2f} us  "
              f"{N * K * 2 / t / 1e3:5.0f} GB/s", flush=True)
        del ws
        torch.cuda.empty_cache()
