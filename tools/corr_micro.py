"""pc_outlier_corr alone: T rows, K input columns of which NOUT are outlier columns, N outputs; in-graph time per launch.
python tools/corr_micro.py [T K N NOUT]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

T, K, N, NOUT = (int(a) for a in (sys.argv[1:5] + ["12", "11008", "4096", "460"][len(sys.argv) - 1:]))
n.load()
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
x = torch.randn((T, K), generator=g).clamp(-5, 5)
cols = torch.randperm(K, generator=g)[:NOUT]
x[torch.randint(0, T, (NOUT,), generator=g), cols] = 9.0
xh, _ = n.to_act_frags(x.to(dev))
codes = torch.zeros_like(xh)
xs = torch.zeros(T, dtype=torch.float32, device=dev)
flags = torch.zeros((2, 16384), dtype=torch.uint8, device=dev)
n.quant_act_i8(xh, True, T, K, codes, xs, flags[0], flags[1])
wq = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).to(dev)
wt = wq.t().contiguous()
wsc = (0.001 + 0.001 * torch.rand(N, generator=g)).to(dev)
corr = torch.zeros((T, N), dtype=torch.float32, device=dev)
has = torch.zeros(1, dtype=torch.int32, device=dev)
print("flagged columns:", int(flags[0].ne(0).sum()))


def fn():
    n.outlier_corr(flags[0], K, xh, codes, True, xs, wt, wsc, None, T, N, corr, has)


for _ in range(3):
    fn()
torch.cuda.synchronize()
gph = torch.cuda.CUDAGraph()
with torch.cuda.graph(gph):
    for _ in range(32):
        fn()
gph.replay(); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gph.replay(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 32 * 1e3)
print(f"T={T} K={K} N={N} outlier columns {NOUT}: {best:.1f} us per launch", flush=True)
