"""One projection of the <= 16-row stack in its int8 forms against the fp16 ones (7b shapes), HIP-event times inside a loop that
cycles eight weight copies (nothing found in the MALL): python tools/i8_proj_micro.py [T]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 12
DEV = "cuda"
NW = 8


def bench(fn, reps=40):
    for i in range(8):
        fn(i % NW)
    ts = []
    for i in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(i % NW); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


for name, N, K in (("down_proj", 4096, 11008), ("o_proj", 4096, 4096)):
    g = torch.Generator(device=DEV).manual_seed(1)
    ws = [(0.03 * torch.randn((N, K), device=DEV, generator=g)).half() for _ in range(NW)]
    wf = [n.to_weight_frags(w) for w in ws]
    qs = [n.quantize_rows_int8(w) for w in ws]
    wf8 = [n.to_weight_frags_i8(q) for q, _ in qs]
    wt8 = [q.t().contiguous() for q, _ in qs]
    x = torch.randn((T, K), device=DEV, generator=g).clamp(-5.9, 5.9)
    hi, lo = n.to_act_frags(x)
    mt = (T + 15) // 16
    codes = torch.empty_like(hi)
    img = torch.empty((mt, K // 64, 64, 16), dtype=torch.int8, device=DEV)
    xs = torch.empty(T, dtype=torch.float32, device=DEV)
    flags = torch.zeros(16384, dtype=torch.uint8, device=DEV)
    n.quant_act_i8(hi, True, T, K, codes, xs, flags, None, codes8=img)
    zero = torch.zeros_like(hi)
    corr = torch.zeros((T, N), dtype=torch.float32, device=DEV)
    has = torch.zeros(1, dtype=torch.int32, device=DEV)
    y = torch.zeros((T, N), dtype=torch.float32, device=DEV)
    print(f"== {name}: N={N} K={K} T={T}   fp16 image {N * K * 2 / 1e6:.1f} MB, int8 image {N * K / 1e6:.1f} MB   (us: median / min)")
    print("fp16 weights, EPI_ADD, one tile per workgroup      : %.1f / %.1f" % bench(lambda i: n.gemm_skinny(wf[i], hi, lo, T, N, K, n.EPI_ADD, y=y, ldy=N)))
    print("int8 weights, fp16 activation planes, EPI_ADD      : %.1f / %.1f" % bench(lambda i: n.gemm_skinny(wf8[i], hi, lo, T, N, K, n.EPI_ADD, y=y, ldy=N, wscale=qs[i][1])))
    print("LLM.int8, corr passed in (no flag scan), code plane: %.1f / %.1f" % bench(lambda i: n.gemm_skinny_a8(wf8[i], qs[i][1], codes, zero, xs, corr, has, T, N, K, n.EPI_ADD, y=y, ldy=N)))
    print("LLM.int8, corr passed in, int8 image               : %.1f / %.1f" % bench(lambda i: n.gemm_skinny_a8(wf8[i], qs[i][1], codes, zero, xs, corr, has, T, N, K, n.EPI_ADD, y=y, ldy=N, codes8=img)))
    print("LLM.int8, correction inside the launch, int8 image : %.1f / %.1f" % bench(lambda i: n.gemm_skinny_a8c(wf8[i], qs[i][1], codes, zero, xs, flags, hi, wt8[i], T, N, K, n.EPI_ADD, y=y, ldy=N, codes8=img)))
