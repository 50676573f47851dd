"""Event-timed gemm_rows_kernel launches at the projection shapes of a long question (65..512 rows), as _forward_skinny issues
them (o_proj / down_proj with 4 K slices into slabs, gate|up with the SiLU epilogue, q|k|v as a plain store here).
python tools/rows_bench.py [model: 7b|13b] [rows ...]      (PC_ROWS_KSL: K slices per shape; PC_ROWS_VARIANT: wide-panel candidates of a -DPC_DEV_ROWS_VARIANTS build)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import _native as n  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "13b"
rows = [int(a) for a in sys.argv[2:]] or [256, 259]
hid, inter, heads = {"7b": (4096, 11008, 32), "13b": (5120, 13824, 40)}[model]
n.load()
dev = "cuda:0"
kslices = [int(v) for v in os.environ.get("PC_ROWS_KSL", "1,4,1,4").split(",")]      # K slices of q|k|v, o_proj, gate|up (SiLU: 1), down_proj
shapes = [("q|k|v", 3 * hid, hid, 0, kslices[0]), ("o_proj", hid, hid, 0, kslices[1]), ("gate|up", 2 * inter, hid, 2, 1), ("down", hid, inter, 0, kslices[3])]
for M in rows:
    total = 0.0
    line = []
    for name, N, K, epi, ksl in shapes:
        ws = [n.to_weight_frags((0.02 * torch.randn((N, K), device=dev)).half()) for _ in range(3)]
        hi, lo = n.to_act_frags(torch.randn((M, K), device=dev))
        y = torch.zeros((max(kslices), M, N), dtype=torch.float32, device=dev)
        KSo = (N // 2 + 31) // 32
        fh = torch.zeros(((M + 15) // 16) * KSo * 512, dtype=torch.float16, device=dev)
        fl = torch.zeros_like(fh)

        def run(i):
            if epi == 2:
                n.gemm_skinny(ws[i % 3], hi, lo, M, N, K, 2, of_hi=fh, of_lo=fl)
            else:
                n.gemm_skinny(ws[i % 3], hi, lo, M, N, K, 0, y=y, ldy=N, kslices=ksl)
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(12):
                run(i)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 12 * 1e3)
        total += best
        line.append(f"{name} {best:6.1f}")
        del ws
    print(f"{model} M={M}: " + "  ".join(line) + f"  | sum {total:6.1f} us  (variant {os.environ.get('PC_ROWS_VARIANT', '-')}, K slices {kslices})", flush=True)
