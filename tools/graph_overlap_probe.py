"""Can the per-layer KV gather hide under the projections when it sits on a PARALLEL BRANCH of the same hipGraph?
32 'layers' of the four 7b projections (pc_gemm, 12 rows) on the capture stream; the gather of one layer's K/V (persona
segments, 56 MB read+write) on a forked stream, joined before the next layer.  python tools/graph_overlap_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

DEV = "cuda:0"
n.load()
M, hid, inter, H, D, L = 12, 4096, 11008, 32, 128, 32
PERSONA = [275, 1, 1, 1, 1, 1, 84, 1, 1, 174, 1, 1, 256, 1, 1, 155, 1, 1, 267, 1, 1, 265, 1, 1, 232]
NW = 3                                               # weight copies rotated (3 x 405 MB > the 256 MB Infinity Cache)
wq = [n.to_weight_frags(torch.randn(3 * hid, hid, device=DEV).half() * 0.05) for _ in range(NW)]
wo = [n.to_weight_frags(torch.randn(hid, hid, device=DEV).half() * 0.05) for _ in range(NW)]
wg = [n.to_weight_frags(torch.randn(2 * inter, hid, device=DEV).half() * 0.05) for _ in range(NW)]
wd = [n.to_weight_frags(torch.randn(hid, inter, device=DEV).half() * 0.05) for _ in range(NW)]
x = torch.randn(M, hid, device=DEV)
g = torch.ones(hid, dtype=torch.float16, device=DEV)
ah, al = n.to_act_frags(torch.randn(M, H * D, device=DEV))
ch = torch.empty((1, inter // 32, 64, 8), dtype=torch.float16, device=DEV); cl = torch.empty_like(ch)
y = torch.zeros((M, 3 * hid), dtype=torch.float32, device=DEV)
segs = [torch.randn((L, 2, H, ln, D), device=DEV).half() for ln in PERSONA]
dst = torch.empty((L, 2, H, 4096, D), dtype=torch.float16, device=DEV)
offs = np.concatenate([[0], np.cumsum(PERSONA)[:-1]]).astype(int).tolist()


def layer(i):
    k = i % NW
    n.gemm_skinny_norm(wq[k], x, g, 1e-5, M, 3 * hid, hid, n.EPI_STORE, y=y, ldy=3 * hid)
    n.gemm_skinny(wo[k], ah, al, M, hid, H * D, n.EPI_ADD, y=x, ldy=hid)
    n.gemm_skinny_norm(wg[k], x, g, 1e-5, M, 2 * inter, hid, n.EPI_SILU, of_hi=ch, of_lo=cl)
    n.gemm_skinny(wd[k], ch, cl, M, hid, inter, n.EPI_ADD, y=x, ldy=hid)


def gather_layer(li, stream=None):
    n.kv_gather([s[li].data_ptr() for s in segs], PERSONA, offs, dst[li], 1, H, D, 4096, stream=stream)


def gather_all():
    n.kv_gather([s.data_ptr() for s in segs], PERSONA, offs, dst, L, H, D, 4096)


def timeit(build, reps=5):
    build()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        build()
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(reps):
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def layers_only():
    for i in range(L):
        layer(i)


def serial():
    gather_all()
    for i in range(L):
        layer(i)


side = torch.cuda.Stream()


def forked():
    main = torch.cuda.current_stream()
    gather_layer(0)                                   # layer 0's K/V must be there first
    for i in range(L):
        if i + 1 < L:
            side.wait_stream(main)
            gather_layer(i + 1, stream=side.cuda_stream)      # the next layer's K/V beside this layer's projections
        layer(i)
        main.wait_stream(side)


a, b, c, d = timeit(layers_only), timeit(gather_all), timeit(serial), timeit(forked)
print(f"projections of 32 layers {a:.3f} ms | gather (all layers, one launch) {b:.3f} ms | serial {c:.3f} ms | "
      f"gather of layer l+1 on a parallel graph branch beside layer l {d:.3f} ms")
