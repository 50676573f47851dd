"""What the staging variant of the ring attention pays for, one layer at a BASELINE shape (default config 4: 13b heads, 259 new
rows over 8 000 staged keys): HIP-event time of (a) the plain ring launch on a staged arena, (b) the gather launch with a row
table that marks every row as already staged (table look-ups and entry DMA, no stores), (c) the gather launch that stages all
rows from a module store.  The launches walk over NL layers of K/V (2 GB of store and of arena at the default shape: nothing is
found in the 256 MB MALL, as inside a forward) and the three variants alternate, so clock drift hits them alike.
python tools/ring_gather_micro.py [H D q S]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

H, D, q_len, S = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (40, 128, 259, 8000)
Hkv = H
L = max(2, min(12, int(2.2e9 / (2 * Hkv * S * D * 2))))
DEV = "cuda"
cap = S + q_len + 8
g = torch.Generator(device=DEV).manual_seed(1)
store = torch.randn((L, 2, Hkv, S, D), device=DEV, generator=g).half()
arena = torch.zeros((L, 2, Hkv, cap, D), dtype=torch.float16, device=DEV)
arena[:, :, :, S:S + q_len] = torch.randn((L, 2, Hkv, q_len, D), device=DEV, generator=g).half()
q = torch.randn((1, q_len, H, D), device=DEV, generator=g).half()
ql = (1e-4 * torch.randn((1, q_len, H, D), device=DEV, generator=g)).half()
klo = (1e-4 * torch.randn((1, Hkv, q_len, D), device=DEV, generator=g)).half()
vlo = (1e-4 * torch.randn((1, Hkv, q_len, D), device=DEV, generator=g)).half()
kv_lo = (klo, vlo, Hkv * q_len * D, q_len * D, -1)
ws = torch.empty(max(n.attn_workspace_bytes(1, H, D, q_len, S + q_len), 4) // 4, dtype=torch.float32, device=DEV)
mt = (q_len + 15) // 16
oh = torch.empty((mt, H * D // 32, 64, 8), dtype=torch.float16, device=DEV)
ol = torch.empty_like(oh)
seg_dt = np.dtype([("src", "<u8"), ("dst_row", "<i4"), ("len", "<i4")])


def table(nseg):
    arr = np.zeros(1, dtype=seg_dt)
    arr[0] = (store.data_ptr(), 0, S)
    segs = torch.from_numpy(arr.view(np.uint8).copy()).to(DEV)
    words = torch.tensor([nseg, S + q_len], dtype=torch.int32, device=DEV)
    rows = torch.zeros(cap * 16, dtype=torch.uint8, device=DEV)
    n.kv_row_table(segs, words[0:1], 64, words[1:2], arena, Hkv, D, cap, rows)
    return rows


tab_staged, tab_fresh = table(0), table(1)
n.kv_gather([store.data_ptr()], [S], [0], arena, L, Hkv, D, cap)
variants = [("plain ring launch, staged arena", None), ("gather launch, every row already staged", tab_staged),
            ("gather launch, stages all %d rows" % S, tab_fresh)]
times = {name: [] for name, _ in variants}


def one(li, tab):
    gather = None if tab is None else (tab, li * 2 * Hkv, (li * 2 + 1) * Hkv)
    n.attn_fwd(q, q_len * H * D, H * D, arena[li, 0].unsqueeze(0), arena[li, 1].unsqueeze(0), L * 2 * Hkv * cap * D, cap * D, None, 0, 0,
               1, H, Hkv, D, q_len, S, 1.0 / np.sqrt(D), ws, out_frag=(oh, ol), q_lo=ql, kv_lo=kv_lo, gather=gather)


for it in range(4 + 36):
    for vi, (name, tab) in enumerate(variants):
        li = (3 * it + vi) % L                                   # (a layer nobody touched for the last L - 1 launches)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); one(li, tab); e1.record(); torch.cuda.synchronize()
        if it >= 4:
            times[name].append(e0.elapsed_time(e1) * 1e3)
print(f"H={H} D={D} q={q_len} S={S} layers={L}   (ring kernel + split merge, us: median / min)")
for name, _ in variants:
    print("%-42s: %.1f / %.1f" % (name, float(np.median(times[name])), float(np.min(times[name]))))
