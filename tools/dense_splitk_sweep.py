"""pc_gemm_dense at few rows: the N = hidden projections with and without the split-K form (pc_gemm_dense_ws), and the
row-split weight-streaming kernel (gemm_rows, what <= PC_MID_MAX_ROWS uses) on the same shapes.
python tools/dense_splitk_sweep.py            (PC_DENSE_KS / PC_DENSE_BN force the slices / the tile width)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

n.load()
dev = "cuda:0"


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


ws = torch.empty(34 << 20, dtype=torch.uint8, device=dev)
for M in (130, 257, 300, 400, 512, 640, 800, 1000, 1300, 1737):
    for name, N, K in (("o", 4096, 4096), ("down", 4096, 11008)):
        x2 = torch.randn((2, M, K), device=dev).half()
        x2[1] *= 2.0 ** -11
        w = (0.02 * torch.randn((N, K), device=dev)).half()
        y = torch.zeros((M, N), dtype=torch.float32, device=dev)
        t0 = timeit(lambda: n.gemm_dense(x2[0], x2[1], w, M, N, K, n.EPI_ADD, y=y))
        t1 = timeit(lambda: n.gemm_dense(x2[0], x2[1], w, M, N, K, n.EPI_ADD, y=y, workspace=ws))
        line = f"M={M:5d} {name:5s} plain {t0:7.1f} us  ws {t1:7.1f} us"
        if M <= 512:
            wf = n.to_weight_frags(w)
            hi, lo = n.to_act_frags((x2[0].float() + x2[1].float()))
            t2 = timeit(lambda: n.gemm_skinny(wf, hi, lo, M, N, K, n.EPI_ADD, y=y, ldy=N))
            line += f"  rows-kernel {t2:7.1f} us"
        print(line + f"   ({2.0 * 2 * M * N * K / min(t0, t1) / 1e6:6.1f} TF best dense)", flush=True)
