"""pc_gemm_chain at the 7b shapes: in-graph time against the four stand-alone launches, and (PC_CHAIN_TRACE=1) the per-phase
wall-clock stamps of every workgroup of one launch.  python tools/chain_trace.py [M]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd"), os.path.join(ROOT, "tests")]
from promptcache_amd import _native as n  # noqa: E402
from test_gpu_chain import Layer, _chain, _separate, _state, SHAPE_7B  # noqa: E402

n.load()
DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12
hid, inter, H, Hkv, D = (SHAPE_7B[k] for k in ("hid", "inter", "H", "Hkv", "D"))
B, q_len, past, eps = 1, M, 1725, 1e-5
cap = past + q_len + 3
NL = 6
layers = [Layer(n, hid, inter, H, Hkv, D, seed=i) for i in range(NL)]
st = _state(n, M, hid, inter, H, Hkv, D, B, q_len, past, cap, seed=0)
sync = n.chain_sync_state(DEV)


def run(fn, iters=4):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def sep():
    for L in layers:
        _separate(n, L, st[0], st[1], st[2], st[3], st[4], M, hid, inter, H, Hkv, D, B, q_len, past, cap, st[5], st[6], st[7], st[8], True, eps)


def chain():
    for L in layers:
        _chain(n, L, st[0], st[1], st[2], st[3], st[4], M, hid, inter, H, Hkv, D, B, q_len, past, cap, st[5], st[6], st[7], st[8], True, eps, sync)


ts, tc = run(sep) / NL, run(chain) / NL
print(f"M={M}: four launches {ts:.2f} us per layer, chain {tc:.2f} us per layer", flush=True)
assert n.chain_sync_error(sync) == 0
if os.environ.get("PC_CHAIN_TRACE"):
    torch.cuda.synchronize()
    _chain(n, layers[0], st[0], st[1], st[2], st[3], st[4], M, hid, inter, H, Hkv, D, B, q_len, past, cap, st[5], st[6], st[7], st[8], True, eps, sync)
    torch.cuda.synchronize()
    w0 = n.load().pc_chain_sync_err_word() + 32
    tr = sync[w0:w0 + 256 * 32].cpu().numpy().view(np.uint64).reshape(256, 16).astype(np.float64)
    t0 = tr[:, 0].min()
    us = (tr - t0) / 100.0                     # wall_clock64: 100 MHz
    names = ["start", "bodies done", "arrive", "released"]
    for ph in range(4):
        for k in range(4):
            col = us[:, 4 * ph + k] if not (ph == 3 and k >= 2) else None
            if col is None or ph == 3 and k >= 2:
                continue
            c = us[:, (12 if ph == 3 else 4 * ph) + k]
            print(f"phase {ph} {names[k]:12s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us")
    for ph, slot in ((0, 1), (1, 5), (2, 9), (3, 13)):
        c = us[:, slot] - us[:, slot - 1]
        print(f"phase {ph} body duration by wg % 8:", " ".join(f"{np.median(c[x::8]):6.2f}/{c[x::8].max():6.2f}" for x in range(8)))
        order = np.argsort(c)[::-1][:12]
        print("   slowest wgs:", " ".join(f"{int(i)}:{c[i]:.1f}" for i in order))
