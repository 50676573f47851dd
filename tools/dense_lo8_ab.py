"""pc_gemm_dense at the encode shapes of the 7b layer, interleaved in one process: both planes in fp16 (rounds 2-4) | the residual plane
on the int8 MFMA (round 5: pc_gemm_dense_lo8; with and without its pc_quant_rows_i8 launch) | hi plane only (the floor).
    python tools/dense_lo8_ab.py [M ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

n.load()
dev = "cuda:0"


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def best(fns, reps=10, rounds=5):
    for f in fns:
        f(); f()
    torch.cuda.synchronize()
    out = [1e30] * len(fns)
    for _ in range(rounds):
        for i, f in enumerate(fns):
            out[i] = min(out[i], timed(f, reps))
    return out


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [800, 3000, 6000]
    hid, inter = 4096, 11008
    shapes = [("qkv", 3 * hid, hid, n.EPI_STORE), ("o", hid, hid, n.EPI_ADD), ("gate|up", 2 * inter, hid, n.EPI_SILU),
              ("down", hid, inter, n.EPI_ADD)]
    for M in Ms:
        tot = [0.0, 0.0, 0.0, 0.0]
        for name, N, K, epi in shapes:
            x2 = torch.randn((2, M, K), device=dev).half()
            x2[1] *= 2.0 ** -11
            w = (0.02 * torch.randn((N, K), device=dev)).half()
            w8, w8s = n.quantize_rows_int8(w)
            y = torch.randn((M, N), dtype=torch.float32, device=dev)
            oh = torch.empty((M, N // 2), dtype=torch.float16, device=dev)
            ol = torch.empty_like(oh)
            codes = torch.empty((M, K), dtype=torch.int8, device=dev)
            sc = torch.empty(M, dtype=torch.float32, device=dev)
            n.quant_rows_i8(x2[1], M, K, codes, sc)
            out = dict(out_hi=oh, out_lo=ol) if epi == n.EPI_SILU else dict(y=y)
            f16 = lambda: n.gemm_dense(x2[0], x2[1], w, M, N, K, epi, **out)
            f8 = lambda: n.gemm_dense_lo8(x2[0], codes, sc, w, w8, w8s, M, N, K, epi, **out)

            def f8q():
                n.quant_rows_i8(x2[1], M, K, codes, sc)
                n.gemm_dense_lo8(x2[0], codes, sc, w, w8, w8s, M, N, K, epi, **out)
            f1 = lambda: n.gemm_dense(x2[0], None, w, M, N, K, epi, **out)
            t16, t8, t8q, t1 = best([f16, f8, f8q, f1])
            fl = 2.0 * M * N * K
            print(f"M={M:5d} {name:8s} N={N:6d} K={K:6d}  fp16 lo {t16:8.1f} us | int8 lo {t8:8.1f} us (x{t16 / t8:.3f}) | + quantiser {t8q:8.1f} us "
                  f"(x{t16 / t8q:.3f}) | hi only {t1:8.1f} us   [{2 * fl / t16 / 1e6:6.0f} / {2 * fl / t8 / 1e6:6.0f} / {fl / t1 / 1e6:6.0f} TF of 2-plane / 2-plane / 1-plane work]",
                  flush=True)
            for i, t in enumerate((t16, t8, t8q, t1)):
                tot[i] += t
        print(f"M={M}: layer projections fp16 lo {tot[0]:.0f} us | int8 lo {tot[1]:.0f} us | with quantisers {tot[2]:.0f} us | hi only {tot[3]:.0f} us", flush=True)


if __name__ == "__main__":
    main()
