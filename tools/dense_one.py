"""One pc_gemm_dense shape in a loop (for rocprofv3 --pmc passes): python tools/dense_one.py M N K epi two reps"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

M, N, K, epi, two, reps = (int(a) for a in sys.argv[1:7])
n.load()
dev = "cuda:0"
x2 = torch.randn((2, M, K), device=dev).half()
x2[1] *= 2.0 ** -11
w = (0.02 * torch.randn((N, K), device=dev)).half()
y = torch.zeros((M, N), dtype=torch.float32, device=dev)
oh = torch.empty((M, max(N // 2, 4)), dtype=torch.float16, device=dev)
ol = torch.empty_like(oh)
for _ in range(reps):
    if epi == n.EPI_SILU:
        n.gemm_dense(x2[0], x2[1] if two else None, w, M, N, K, epi, out_hi=oh, out_lo=ol)
    else:
        n.gemm_dense(x2[0], x2[1] if two else None, w, M, N, K, epi, y=y)
torch.cuda.synchronize()
