"""Per-wave wall-clock stamps of one cached-prefill attention launch (attn_small_kernel) in the middle of an in-graph sequence.
python tools/attn_trace.py [S q tail gather]     gather = 1: the staging variant (pc_attn gather_rows: rows read from module stores and
written to the arena as they pass)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

S, q, tail, gat = (int(a) for a in (sys.argv[1:5] + ["1725", "12", "1", "0"][len(sys.argv) - 1:]))
lib = n.load()
dev = "cuda:0"
H = Hkv = 32; D = 128; L = 8
cap = max(4096, S + q + 64)
arena = torch.randn((L, 2, Hkv, cap, D), device=dev).half()
q16 = torch.randn((q, H * D), device=dev).half()
q16l = (torch.randn((q, H * D), device=dev) * 2 ** -11).half()
lo = torch.zeros((2, Hkv, 320, D), device=dev).half()
ah = torch.empty(((q + 15) // 16, H * D // 32, 64, 8), dtype=torch.float16, device=dev); al = torch.empty_like(ah)
ws = torch.empty(max(n.attn_workspace_bytes(1, H, D, q, S + q), 4) // 4, dtype=torch.float32, device=dev)
past_dev = torch.tensor([S, 0], dtype=torch.int32, device=dev)
kvlo = (lo[0], lo[1], Hkv * 320 * D, 320 * D, -1) if tail else None
trace = torch.zeros(1024 * 16, dtype=torch.int64, device=dev)


rows = None
if gat:
    lens = [275, 1, 1, 1, 1, 1, 84, 1, 1, 174, 1, 1, 256, 1, 1, 155, 1, 1, 267, 1, 1, 265, 1, 1, 232] if S == 1725 else [S]
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(int)
    stores = [torch.randn((L, 2, Hkv, ln, D), device=dev).half() for ln in lens]
    arr = np.array([(s.data_ptr(), int(o), int(ln)) for s, o, ln in zip(stores, offs, lens)],
                   dtype=np.dtype([("src", "<u8"), ("dst_row", "<i4"), ("len", "<i4")]))
    segs = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
    words = torch.tensor([len(lens), S + q], dtype=torch.int32, device=dev)
    rows = torch.zeros(cap * 16, dtype=torch.uint8, device=dev)
    n.kv_row_table(segs, words[0:1], 64, words[1:2], arena, Hkv, D, cap, rows)


def step(i):
    li = i % L
    n.attn_fwd(q16, q * H * D, H * D, arena[li, 0], arena[li, 1], 2 * Hkv * cap * D, cap * D, None, 0, 0, 1, H, Hkv, D, q, S,
               1.0 / D ** 0.5, ws, past_len_dev=past_dev, out_frag=(ah, al), q_lo=q16l, kv_lo=kvlo,
               gather=None if rows is None else (rows, li * 2 * Hkv, (li * 2 + 1) * Hkv))


for i in range(8):
    step(i)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(6):
        if i == 4:
            lib.pc_dev_attn_trace(trace.data_ptr())
        step(i)
        if i == 4:
            lib.pc_dev_attn_trace(None)
for _ in range(3):
    trace.zero_(); g.replay(); torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(-1, 4, 4)
t = t[(t[:, :, 0] > 0).any(axis=1)]
t0 = t[:, :, 0][t[:, :, 0] > 0].min()
print(f"attn_small_kernel S={S} q={q} tail={tail} staging={gat}: {len(t)} workgroups stamped")
for slot, label in enumerate(("entry", "first tile scored", "key slice done", "done (partial stored)")):
    v = t[:, :, slot].astype(np.float64)
    v = (v[v > 0] - t0) * 0.01
    if len(v):
        print(f"    {label:24s} n={len(v):5d} min {v.min():6.2f}  p50 {np.median(v):6.2f}  p95 {np.percentile(v, 95):6.2f}  max {v.max():6.2f} us")
