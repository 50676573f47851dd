"""One projection shape on the row-split weight-streaming kernel (65..512 rows), in a loop: for rocprofv3 --pmc.
python tools/rows_one.py M N K epi reps"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

M, N, K, epi, reps = (int(a) for a in sys.argv[1:6])
n.load()
dev = "cuda:0"
ws = [n.to_weight_frags((0.02 * torch.randn((N, K), device=dev)).half()) for _ in range(3)]
hi, lo = n.to_act_frags(torch.randn((M, K), device=dev))
y = torch.zeros((4, M, N), dtype=torch.float32, device=dev)
KSo = (N // 2 + 31) // 32
fh = torch.zeros(((M + 15) // 16) * KSo * 512, dtype=torch.float16, device=dev)
fl = torch.zeros_like(fh)
for i in range(reps):
    if epi == 2:
        n.gemm_skinny(ws[i % 3], hi, lo, M, N, K, 2, of_hi=fh, of_lo=fl)
    else:
        n.gemm_skinny(ws[i % 3], hi, lo, M, N, K, 0, y=y, ldy=N, kslices=4)
torch.cuda.synchronize()
