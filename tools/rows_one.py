"""ONE gemm_rows_kernel shape in a loop (for tools/pmc_rows.sh / rocprofv3): python tools/rows_one.py M N K epi [launches] [kslices]
epi 0: plain store (K slices -> slabs), 2: gate|up with the SiLU epilogue (N = 2 x intermediate)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

M, N, K, epi = (int(a) for a in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 12
ksl = int(sys.argv[6]) if len(sys.argv) > 6 else 1
n.load()
dev = "cuda:0"
ws = [n.to_weight_frags((0.02 * torch.randn((N, K), device=dev)).half()) for _ in range(3)]
hi, lo = n.to_act_frags(torch.randn((M, K), device=dev))
y = torch.zeros((ksl, M, N), dtype=torch.float32, device=dev)
KSo = (N // 2 + 31) // 32
fh = torch.zeros(((M + 15) // 16) * KSo * 512, dtype=torch.float16, device=dev)
fl = torch.zeros_like(fh)
for i in range(iters):
    if epi == 2:
        n.gemm_skinny(ws[i % 3], hi, lo, M, N, K, 2, of_hi=fh, of_lo=fl)
    else:
        n.gemm_skinny(ws[i % 3], hi, lo, M, N, K, 0, y=y, ldy=N, kslices=ksl)
torch.cuda.synchronize()
print("done", M, N, K, epi, iters, ksl)
