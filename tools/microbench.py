"""Kernel micro-benchmarks (run on the GPU box): achieved GB/s of kv_gather, time and GB/s|TFLOP/s of
attention at the BASELINE shapes.  HIP-event timing on torch's current stream (where the C-ABI launches)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "prompt-cache_amd"))
from promptcache_amd import _native as n  # noqa: E402

DEV = "cuda"
PERSONA = [275, 1, 1, 1, 1, 1, 84, 1, 1, 174, 1, 1, 256, 1, 1, 155, 1, 1, 267, 1, 1, 265, 1, 1, 232]
GAME = [306, 2, 2, 2, 2, 76, 800, 800, 800, 800, 800]


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def bench_gather(name, lens, L, Hkv, D, max_ctx):
    segs = [torch.randn((L, 2, Hkv, ln, D), device=DEV).half() for ln in lens]
    dst = torch.empty((L, 2, Hkv, max_ctx, D), dtype=torch.float16, device=DEV)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(int).tolist()
    ptrs = [s.data_ptr() for s in segs]
    S = sum(lens)
    byts = 2 * S * L * 2 * Hkv * D * 2
    med, best = timeit(lambda: n.kv_gather(ptrs, lens, offs, dst, L, Hkv, D, max_ctx))
    # torch reference: one big D2D copy of the same byte count
    a = torch.empty(byts // 4, dtype=torch.float16, device=DEV)
    b = torch.empty_like(a)
    cmed, _ = timeit(lambda: b.copy_(a))
    print(f"gather {name}: S={S} nseg={len(lens)} bytes={byts/1e9:.3f} GB  med {med*1e3:.1f} us  "
          f"{byts/med/1e6:.0f} GB/s (best {byts/best/1e6:.0f})  | torch D2D copy same bytes {byts/cmed/1e6:.0f} GB/s")


def bench_attn(name, H, Hkv, D, q_len, past):
    cap = past + q_len
    q = torch.randn((1, q_len, H, D), device=DEV).half()
    k = torch.randn((1, Hkv, cap, D), device=DEV).half()
    v = torch.randn((1, Hkv, cap, D), device=DEV).half()
    out = torch.empty((1, q_len, H * D), dtype=torch.float16, device=DEV)
    ws = torch.empty(max(n.attn_workspace_bytes(1, H, D, q_len, cap), 4) // 4, dtype=torch.float32, device=DEV)
    sc = 1.0 / np.sqrt(D)
    fn = lambda: n.attn_fwd(q, q_len * H * D, H * D, k, v, Hkv * cap * D, cap * D, out, q_len * H * D, H * D,
                            1, H, Hkv, D, q_len, past, sc, ws)
    med, best = timeit(fn)
    byts = 2 * Hkv * cap * D * 2 + 2 * H * q_len * D * 2
    flops = 4 * H * D * q_len * (past + (q_len + 1) / 2)
    print(f"attn {name}: H={H} q={q_len} past={past}  med {med*1e3:.1f} us (best {best*1e3:.1f})  "
          f"{byts/med/1e6:.0f} GB/s  {flops/med/1e9:.1f} TFLOP/s")


def bench_gemm(name, M, N, K, epi, kq=1):
    ncopy = max(2, int(700e6 / (N * K * 2)) + 1)          # rotate through > 256 MB (Infinity Cache) of weights
    ws = [n.to_weight_frags(torch.randn(N, K, device=DEV).half() * 0.05) for _ in range(ncopy)]
    x = torch.randn(M, K, device=DEV)
    hi, lo = n.to_act_frags(x)
    mt = (M + 15) // 16
    y = torch.zeros((kq, M, N), dtype=torch.float32, device=DEV)
    oh = torch.empty((mt, max(N // 64, 1), 64, 8), dtype=torch.float16, device=DEV)
    ol = torch.empty_like(oh)
    i = [0]

    def fn():
        i[0] = (i[0] + 1) % ncopy
        if epi == 2:
            n.gemm_skinny(ws[i[0]], hi, lo, M, N, K, 2, of_hi=oh, of_lo=ol)
        else:
            n.gemm_skinny(ws[i[0]], hi, lo, M, N, K, epi, y=y, ldy=N, kslices=kq)
    med, best = timeit(fn, iters=40)
    print(f"gemm {name}: M={M} N={N} K={K} epi={epi} kq={kq}  med {med*1e3:.1f} us (best {best*1e3:.1f})  weights {N*K*2/med/1e6:.0f} GB/s")


if __name__ == "__main__":
    if "--gather-only" in sys.argv:      # PMC passes: a handful of launches of the persona gather, nothing else
        segs = [torch.randn((32, 2, 32, ln, 128), device=DEV).half() for ln in PERSONA]
        dst = torch.empty((32, 2, 32, 4096, 128), dtype=torch.float16, device=DEV)
        offs = np.concatenate([[0], np.cumsum(PERSONA)[:-1]]).astype(int).tolist()
        for _ in range(5):
            n.kv_gather([s_.data_ptr() for s_ in segs], PERSONA, offs, dst, 32, 32, 128, 4096)
        torch.cuda.synchronize()
        print("gather-only done: algorithmic bytes per launch", 2 * sum(PERSONA) * 32 * 2 * 32 * 128 * 2)
        sys.exit(0)
    if "--gemm-only" in sys.argv:        # PMC passes: the kernel of bench.py's roofline_gemm (norm-fused gate|up, 12 rows, 7b)
        hid, inter, T = 4096, 11008, 12
        ncopy = 5                                             # 5 x 180 MB > the 256 MB Infinity Cache
        ws = [n.to_weight_frags(torch.randn(2 * inter, hid, device=DEV).half() * 0.05) for _ in range(ncopy)]
        x = torch.randn(T, hid, device=DEV)
        g = torch.ones(hid, dtype=torch.float16, device=DEV)
        oh = torch.empty((1, inter // 32, 64, 8), dtype=torch.float16, device=DEV); ol = torch.empty_like(oh)
        for i in range(10):
            n.gemm_skinny_norm(ws[i % ncopy], x, g, 1e-5, T, 2 * inter, hid, n.EPI_SILU, of_hi=oh, of_lo=ol)
        torch.cuda.synchronize()
        print("gemm-only done: algorithmic weight bytes per launch", 2 * inter * hid * 2)
        sys.exit(0)
    if "--attn-cached-only" in sys.argv:  # PMC passes: the cached-prefill attention of the persona prompt (q = 12 over S = 1725)
        for _ in range(2):
            bench_attn("persona cached", 32, 32, 128, 12, 1725)
        sys.exit(0)
    if "--gemm-w8" in sys.argv:          # int8 weight images vs fp16, the four projection shapes of a 7b layer at 12 rows
        def w8(name, M, N, K, epi, kq=1, norm=False):
            ncopy = max(2, int(700e6 / (N * K)) + 1)
            qs = [n.quantize_rows_int8(torch.randn(N, K, device=DEV).half() * 0.05) for _ in range(2)]
            ws = [(n.to_weight_frags_i8(qs[i % 2][0]), qs[i % 2][1]) for i in range(ncopy)]
            x = torch.randn(M, K, device=DEV); hi, lo = n.to_act_frags(x)
            g = torch.ones(K, dtype=torch.float16, device=DEV)
            mt = (M + 15) // 16
            y = torch.zeros((kq, M, N), dtype=torch.float32, device=DEV)
            oh = torch.empty((mt, max(N // 64, 1), 64, 8), dtype=torch.float16, device=DEV); ol = torch.empty_like(oh)
            i = [0]
            def fn():
                i[0] = (i[0] + 1) % ncopy
                wf, sc = ws[i[0]]
                if norm:
                    n.gemm_skinny_norm(wf, x, g, 1e-5, M, N, K, epi, y=y if epi == 0 else None, ldy=N, of_hi=oh, of_lo=ol, wscale=sc)
                elif epi == 2:
                    n.gemm_skinny(wf, hi, lo, M, N, K, 2, of_hi=oh, of_lo=ol, wscale=sc)
                else:
                    n.gemm_skinny(wf, hi, lo, M, N, K, epi, y=y, ldy=N, kslices=kq, wscale=sc)
            med, best = timeit(fn, iters=40)
            print(f"w8 {name}{' norm' if norm else ''}: M={M} N={N} K={K} epi={epi} kq={kq}  med {med*1e3:.1f} us (best {best*1e3:.1f})  int8 weights {N*K/med/1e6:.0f} GB/s")
        M = 12
        w8("gate_up", M, 22016, 4096, 2); w8("gate_up", M, 22016, 4096, 2, norm=True)
        w8("qkv", M, 12288, 4096, 0); w8("o", M, 4096, 4096, 0, 4); w8("down", M, 4096, 11008, 0, 4)
        bench_gemm("gate_up fp16", M, 22016, 4096, 2); bench_gemm("down fp16", M, 4096, 11008, 0, 4)
        sys.exit(0)
    if "--attn-only" in sys.argv:        # PMC passes: a few launches of the large-q (MFMA-bound) attention shapes
        for _ in range(2):
            bench_attn("nocache 4404", 32, 32, 128, 4404, 0)
            bench_attn("nocache 1737", 32, 32, 128, 1737, 0)
        sys.exit(0)
    if "--gemm-msweep" in sys.argv:      # how the weight-streaming rate holds up as the row tiles (MT = ceil(M/16)) grow
        for M in (12, 16, 24, 32, 48, 64):
            bench_gemm("qkv", M, 12288, 4096, 0)
            bench_gemm("gate_up", M, 22016, 4096, 2)
            bench_gemm("down", M, 4096, 11008, 0, 4)
        sys.exit(0)
    if "--gemm-rows" in sys.argv:        # 65..512 rows: the row-split kernel, 7b and 13b projection shapes
        for M in (96, 128, 259, 512):
            bench_gemm("qkv-7b", M, 12288, 4096, 0)
            bench_gemm("gate_up-7b", M, 22016, 4096, 2)
            bench_gemm("down-7b", M, 4096, 11008, 0, 4)
            bench_gemm("o-7b", M, 4096, 4096, 0, 4)
        for M in (259,):
            bench_gemm("qkv-13b", M, 15360, 5120, 0)
            bench_gemm("gate_up-13b", M, 27648, 5120, 2)
            bench_gemm("down-13b", M, 5120, 13824, 0, 4)
            bench_gemm("o-13b", M, 5120, 5120, 0, 4)
        sys.exit(0)
    if "--gemm" in sys.argv:
        M = 12
        bench_gemm("qkv", M, 12288, 4096, 0)
        bench_gemm("o", M, 4096, 4096, 1)
        for kq in (2, 4, 8):
            bench_gemm("o", M, 4096, 4096, 0, kq)
            bench_gemm("down", M, 4096, 11008, 0, kq)
        bench_gemm("gate_up", M, 22016, 4096, 2)
        bench_gemm("down", M, 4096, 11008, 1)
        bench_gemm("lm_head", M, 32000, 4096, 0)
        bench_gemm("qkv M=1", 1, 12288, 4096, 0)
        bench_gemm("qkv M=32", 32, 12288, 4096, 0)
        sys.exit(0)
    print(torch.cuda.get_device_name(0))
    bench_gather("persona-7b", PERSONA, 32, 32, 128, 4096)
    bench_gather("game-7b", GAME, 32, 32, 128, 5000)
    bench_gather("longbench-13b", [30, 8000, 20], 40, 40, 128, 9186)
    bench_attn("persona cached", 32, 32, 128, 12, 1725)
    bench_attn("game cached", 32, 32, 128, 14, 4390)
    bench_attn("13b 8k cached", 40, 40, 128, 260, 8000)
    bench_attn("decode", 32, 32, 128, 1, 1737)
    bench_attn("encode 456", 32, 32, 128, 456, 0)
    bench_attn("encode 815", 32, 32, 128, 815, 0)
    bench_attn("nocache 1737", 32, 32, 128, 1737, 0)
    bench_attn("nocache 4404", 32, 32, 128, 4404, 0)
