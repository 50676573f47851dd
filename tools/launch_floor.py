"""Per-launch floor of the weight-streaming GEMM inside a captured graph: the same launch with K = 64 / 512 / full.
The difference between the rows is the incremental streaming rate; the K = 64 row is ramp + reduction + epilogue."""
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, "prompt-cache_amd"); sys.path.insert(0, ".")
import torch
from promptcache_amd import _native as n
DEV = "cuda:0"
def timeit(fn, iters=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (N, K, epi, kq) in ((4096, 64, 0, 1), (4096, 512, 0, 1), (4096, 4096, 0, 4), (22016, 64, 2, 1), (22016, 512, 2, 1), (22016, 4096, 2, 1)):
    M = 12
    ncopy = max(2, int(600e6 / (N * K * 2)) + 1)
    ws = [n.to_weight_frags(torch.randn(N, K, device=DEV).half() * 0.05) for _ in range(min(ncopy, 64))]
    x = torch.randn(M, K, device=DEV); hi, lo = n.to_act_frags(x)
    y = torch.zeros((kq, M, N), dtype=torch.float32, device=DEV)
    oh = torch.empty((1, max(N // 64, 1), 64, 8), dtype=torch.float16, device=DEV); ol = torch.empty_like(oh)
    i = [0]
    def fn():
        i[0] = (i[0] + 1) % len(ws)
        if epi == 2: n.gemm_skinny(ws[i[0]], hi, lo, M, N, K, 2, of_hi=oh, of_lo=ol)
        else: n.gemm_skinny(ws[i[0]], hi, lo, M, N, K, 0, y=y, ldy=N, kslices=kq)
    t = timeit(fn)
    print(f"N={N} K={K} epi={epi} kq={kq}: {t:.2f} us per launch inside a graph  ({N*K*2/1e6:.1f} MB -> {N*K*2/t/1e3:.0f} GB/s)")
