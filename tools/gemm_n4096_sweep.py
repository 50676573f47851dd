"""In-graph time of the N = hidden projections (o_proj K = 4096, down_proj K = 11008) for M rows.
python tools/gemm_n4096_sweep.py M epi(0 store|1 add) kslices     (env PC_GEMM_T / PC_GEMM_U select the launch shape)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

DEV = "cuda:0"
M, epi, kq = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])


def timeit(fn, iters=128, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for N, K in ((4096, 4096), (4096, 11008)):
    ws = [n.to_weight_frags(torch.randn(N, K, device=DEV).half() * 0.05) for _ in range(24)]
    hi, lo = n.to_act_frags(torch.randn(M, K, device=DEV))
    y = torch.zeros((kq, M, N), dtype=torch.float32, device=DEV)
    i = [0]

    def fn():
        i[0] = (i[0] + 1) % len(ws)
        n.gemm_skinny(ws[i[0]], hi, lo, M, N, K, epi, y=y, ldy=N, kslices=kq)
    t = timeit(fn)
    print(f"M={M} N={N} K={K} epi={epi} kq={kq} T={os.environ.get('PC_GEMM_T', '-')} U={os.environ.get('PC_GEMM_U', '-')}: "
          f"{t:.2f} us  {N * K * 2 / t / 1e3:.0f} GB/s", flush=True)
