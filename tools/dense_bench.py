"""Event-time pc_gemm_dense against the round-1 formulation (hipBLASLt on [hi; lo]-stacked rows + separate fp32 passes)
at the encode shapes of the 7b layer.  python tools/dense_bench.py [M ...]"""
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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [800, 3000, 6000]
    hid, inter = 4096, 11008
    shapes = [("qkv", 3 * hid, hid, n.EPI_STORE), ("o", hid, hid, n.EPI_ADD), ("gate|up", 2 * inter, hid, n.EPI_SILU),
              ("down", hid, inter, n.EPI_ADD)]
    for M in Ms:
        tot_a = tot_b = 0.0
        for name, N, K, epi in shapes:
            x2 = torch.randn((2, M, K), device=dev).half()
            x2[1] *= 2.0 ** -11
            w = (0.02 * torch.randn((N, K), device=dev)).half()
            y = torch.zeros((M, N if epi != n.EPI_SILU else N), dtype=torch.float32, device=dev)
            oh = torch.empty((M, N // 2), dtype=torch.float16, device=dev)
            ol = torch.empty_like(oh)
            if epi == n.EPI_SILU:
                fa = lambda: n.gemm_dense(x2[0], x2[1], w, M, N, K, epi, out_hi=oh, out_lo=ol)
                fa1 = lambda: n.gemm_dense(x2[0], None, w, M, N, K, epi, out_hi=oh, out_lo=ol)

                def fb():
                    g = torch.mm(x2.view(2 * M, K), w.t(), out_dtype=torch.float32)
                    n.silu_mul_split(g[:M], g[M:], oh, ol, M, N // 2)
            else:
                fa = lambda: n.gemm_dense(x2[0], x2[1], w, M, N, K, epi, y=y)
                fa1 = lambda: n.gemm_dense(x2[0], None, w, M, N, K, epi, y=y)

                def fb():
                    g = torch.mm(x2.view(2 * M, K), w.t(), out_dtype=torch.float32)
                    if epi == n.EPI_ADD:
                        n.add3(y, g[:M], g[M:], M * N)
            ta, ta1, tb = timeit(fa), timeit(fa1), timeit(fb)
            fl = 2.0 * 2 * M * N * K
            print(f"M={M:5d} {name:8s} N={N:6d} K={K:6d}  dense(hi+lo) {ta:8.1f} us {fl / ta / 1e6:7.1f} TF | hi only {ta1:8.1f} us "
                  f"{fl / 2 / ta1 / 1e6:7.1f} TF | hipBLASLt stacked + epilogue {tb:8.1f} us {fl / tb / 1e6:7.1f} TF")
            tot_a += ta
            tot_b += tb
        print(f"M={M}: layer projections {tot_a:.0f} us (dense) vs {tot_b:.0f} us (round-1 formulation)")


if __name__ == "__main__":
    main()
