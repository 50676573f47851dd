"""A/B of pc_gemm_dense's two main loops inside one process: PC_DENSE_PP=1 (round 5: the two wave groups of a workgroup one
barrier apart, LDS-DMA pieces between the MFMAs) against PC_DENSE_PP=0 (rounds 2-4: lockstep, one barrier per K-step), interleaved,
at the encode shapes of the 7b layer, hi + lo planes and hi only; the outputs of the two must be bit-identical.
    python tools/dense_pp_ab.py [M ...]"""
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


def ab(fn, reps=10, rounds=5):
    best = {}
    for pp in ("1", "0"):
        os.environ["PC_DENSE_PP"] = pp
        fn(); fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for pp in ("1", "0"):
            os.environ["PC_DENSE_PP"] = pp
            t = timed(fn, reps)
            best[pp] = min(best.get(pp, 1e30), t)
    return best["1"], best["0"]


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [300, 800, 3000, 6000]
    hid, inter = 4096, 11008
    shapes = [("qkv", 3 * hid, hid, n.EPI_STORE), ("o", hid, hid, n.EPI_ADD), ("gate|up", 2 * inter, hid, n.EPI_SILU),
              ("down", hid, inter, n.EPI_ADD)]
    for M in Ms:
        tot = {"1": 0.0, "0": 0.0}
        for name, N, K, epi in shapes:
            x2 = torch.randn((2, M, K), device=dev).half()
            x2[1] *= 2.0 ** -11
            w = (0.02 * torch.randn((N, K), device=dev)).half()
            y0 = torch.randn((M, N), dtype=torch.float32, device=dev)
            y = y0.clone()
            oh = torch.empty((M, N // 2), dtype=torch.float16, device=dev)
            ol = torch.empty_like(oh)
            for planes, lo in ((2, x2[1]), (1, None)):
                if epi == n.EPI_SILU:
                    fn = lambda: n.gemm_dense(x2[0], lo, w, M, N, K, epi, out_hi=oh, out_lo=ol)
                else:
                    fn = lambda: n.gemm_dense(x2[0], lo, w, M, N, K, epi, y=y)
                outs = {}
                for pp in ("1", "0"):                      # bit-identical results
                    os.environ["PC_DENSE_PP"] = pp
                    y.copy_(y0); oh.zero_(); ol.zero_()
                    fn()
                    torch.cuda.synchronize()
                    outs[pp] = (y.clone(), oh.clone(), ol.clone())
                same = all(torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a.view(torch.int16),
                                       b.view(torch.int32) if b.dtype == torch.float32 else b.view(torch.int16))
                           for a, b in zip(outs["1"], outs["0"]))
                t1, t0 = ab(fn)
                fl = 2.0 * planes * M * N * K
                print(f"M={M:5d} {name:8s} N={N:6d} K={K:6d} planes={planes}  ping-pong {t1:8.1f} us {fl / t1 / 1e6:7.1f} TF | lockstep "
                      f"{t0:8.1f} us {fl / t0 / 1e6:7.1f} TF | x{t0 / t1:.3f}  bit-identical={same}", flush=True)
                if planes == 2:
                    tot["1"] += t1; tot["0"] += t0
        print(f"M={M}: layer projections (hi + lo) ping-pong {tot['1']:.0f} us vs lockstep {tot['0']:.0f} us  (x{tot['0'] / tot['1']:.3f})", flush=True)


if __name__ == "__main__":
    main()
