"""Few hundred rows: the row-split weight-streaming kernel (gemm_rows, <= PC_MID_MAX_ROWS) against pc_gemm_dense (+ split-K
workspace) on the four projections of the 7b layer.  python tools/dense_vs_rows.py"""
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
hid, inter = 4096, 11008
for M in (65, 100, 130, 200, 259, 400, 512):
    tot_r = tot_d = 0.0
    for name, N, K, epi in (("qkv", 3 * hid, hid, n.EPI_STORE), ("o", hid, hid, n.EPI_ADD), ("gate|up", 2 * inter, hid, n.EPI_SILU),
                            ("down", hid, inter, n.EPI_ADD)):
        x2 = torch.randn((2, M, K), device=dev).half()
        x2[1] *= 2.0 ** -11
        w = (0.02 * torch.randn((N, K), device=dev)).half()
        y = torch.zeros((M, N), dtype=torch.float32, device=dev)
        oh = torch.empty((M, N // 2), dtype=torch.float16, device=dev)
        ol = torch.empty_like(oh)
        wf = n.to_weight_frags(w)
        hi, lo = n.to_act_frags((x2[0].float() + x2[1].float()))
        KSo = (N // 2 + 31) // 32
        fh = torch.zeros(((M + 15) // 16) * KSo * 512, dtype=torch.float16, device=dev)
        fl = torch.zeros_like(fh)
        if epi == n.EPI_SILU:
            td = timeit(lambda: n.gemm_dense(x2[0], x2[1], w, M, N, K, epi, out_hi=oh, out_lo=ol))
            tr = timeit(lambda: n.gemm_skinny(wf, hi, lo, M, N, K, epi, of_hi=fh, of_lo=fl))
        else:
            td = timeit(lambda: n.gemm_dense(x2[0], x2[1], w, M, N, K, epi, y=y, workspace=ws))
            slabs = torch.zeros((4, M, N), dtype=torch.float32, device=dev)       # the product's form: 4 K-slices -> slabs
            tr = timeit(lambda: n.gemm_skinny(wf, hi, lo, M, N, K, n.EPI_STORE, y=slabs, ldy=N, kslices=4)) if epi == n.EPI_ADD \
                else timeit(lambda: n.gemm_skinny(wf, hi, lo, M, N, K, epi, y=y, ldy=N))
        tot_r += tr; tot_d += td
        print(f"M={M:4d} {name:8s} rows-kernel {tr:7.1f} us   dense {td:7.1f} us", flush=True)
        del wf, w
    print(f"M={M:4d} layer projections: rows-kernel {tot_r:7.1f} us   dense {tot_d:7.1f} us", flush=True)
