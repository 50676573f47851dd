"""Where the time of a weight-streaming launch goes: per-wave wall-clock stamps (entry, K loop done, reduced, done) of one
launch in the middle of an in-graph sequence, for the four projections of a 7b layer at M rows (dev hook pc_dev_gemm_trace).
python tools/gemm_trace.py [M]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12
lib = n.load()
hid, inter, H, D = 4096, 11008, 32, 128
TICK_US = 0.01          # wall_clock64: 100 MHz


def frags(N, K, cnt):
    return [n.to_weight_frags(torch.randn(N, K, device=DEV).half() * 0.05) for _ in range(cnt)]


x = torch.randn(M, hid, device=DEV)
g = torch.ones(hid, dtype=torch.float16, device=DEV)
ah, al = n.to_act_frags(torch.randn(M, H * D, device=DEV))
ch, cl = n.to_act_frags(torch.randn(M, inter, device=DEV))
oh = torch.empty((1, inter // 32, 64, 8), dtype=torch.float16, device=DEV); ol = torch.empty_like(oh)
y = torch.zeros((M, hid), dtype=torch.float32, device=DEV)
cs = torch.zeros((M, D // 2, 2), dtype=torch.float32, device=DEV); cs[..., 0] = 1
q16 = torch.empty((M, H * D), dtype=torch.float16, device=DEV); q16l = torch.empty_like(q16)
arena = torch.zeros((2, H, 4096, D), dtype=torch.float16, device=DEV)
perm = n.rope_row_perm(H, H, D) if hasattr(n, "rope_row_perm") else None

cases = {
    "o_proj (EPI_ADD)": (frags(hid, H * D, 8), lambda w: n.gemm_skinny(w, ah, al, M, hid, H * D, n.EPI_ADD, y=y, ldy=hid), hid * H * D * 2),
    "down (EPI_ADD)": (frags(hid, inter, 4), lambda w: n.gemm_skinny(w, ch, cl, M, hid, inter, n.EPI_ADD, y=y, ldy=hid), hid * inter * 2),
    "gate|up (NORM, SILU)": (frags(2 * inter, hid, 3), lambda w: n.gemm_skinny_norm(w, x, g, 1e-5, M, 2 * inter, hid, n.EPI_SILU, of_hi=oh, of_lo=ol), 2 * inter * hid * 2),
    "q|k|v (NORM, ROPE)": (frags(3 * hid, hid, 4), lambda w: n.gemm_qkv_rope_norm(w, x, g, 1e-5, M, hid, cs, q16, q16l, H * D, arena[0], arena[1], 0, 4096 * D, 1, H, H, D, M, 100, 4096), 3 * hid * hid * 2),
}
trace = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device=DEV)
for name, (ws, fn, nbytes) in cases.items():
    for w in ws:
        fn(w)
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        for i in range(6):
            if i == 4:
                lib.pc_dev_gemm_trace(trace.data_ptr())
            fn(ws[i % len(ws)])
            if i == 4:
                lib.pc_dev_gemm_trace(None)
    for _ in range(3):
        trace.zero_()
        gph.replay()
        torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(-1, 8, 4)
    t = t[(t[:, :, 0] > 0).any(axis=1)]                      # workgroups that stamped
    t0 = t[:, :, 0][t[:, :, 0] > 0].min()
    print(f"{name}: {len(t)} workgroups, {nbytes / 1e6:.1f} MB", flush=True)
    for slot, label in enumerate(("entry", "K loop done", "reduced (LDS barrier)", "done (stores drained)")):
        v = t[:, :, slot].astype(np.float64)
        v = (v[v > 0] - t0) * TICK_US
        if len(v):
            print(f"    {label:24s} min {v.min():6.2f}  p50 {np.median(v):6.2f}  p95 {np.percentile(v, 95):6.2f}  max {v.max():6.2f} us")
    tt = trace.cpu().numpy().reshape(-1, 8, 4)[:len(t)]
    wgdone = (tt[:, :, 2].max(axis=1).astype(np.float64) - t0) * TICK_US            # per workgroup: all waves through the K loop
    print("    workgroup K-loop completion by blockIdx % 8 (XCD): " +
          " ".join(f"{wgdone[x::8].mean():.2f}" for x in range(8)) + "   | by blockIdx // 32: " +
          " ".join(f"{wgdone[i * 32:(i + 1) * 32].mean():.2f}" for i in range(len(wgdone) // 32)))
    kw = (t[:, :, 1].astype(np.float64) - t0) * TICK_US
    print("    K-loop completion by wave index: " + " ".join(f"{kw[:, w].mean():.2f}" for w in range(8)) +
          "   (std " + " ".join(f"{kw[:, w].std():.2f}" for w in range(8)) + ")")
    kl = (t[:, :, 1].astype(np.float64) - t[:, :, 0]) * TICK_US
    kl = kl[(t[:, :, 1] > 0) & (t[:, :, 0] > 0)]
    print(f"    per-wave K loop: p50 {np.median(kl):.2f}  max {kl.max():.2f} us")
