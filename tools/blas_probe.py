"""Host BLAS probe (what bounds the numpy oracle in the full-depth tests and bench.py's cpu_baseline): sgemm x @ W^T at the row counts
an oracle scaffold pass has, by BLAS thread count.   python tools/blas_probe.py"""
import os
import time

import numpy as np

try:
    from threadpoolctl import threadpool_info, threadpool_limits
except Exception:  # pragma: no cover
    threadpool_info = threadpool_limits = None

print("cpus", os.cpu_count(), threadpool_info() if threadpool_info else None)
rng = np.random.default_rng(0)
K, N = 5120, 13824
w = rng.standard_normal((N, K), dtype=np.float32)
for M in (130, 350, 650, 2000):
    x = rng.standard_normal((M, K), dtype=np.float32)
    for nt in (8, 16, 32, 64, 128, 256):
        ctx = threadpool_limits(limits=nt) if threadpool_limits else None
        try:
            (x @ w.T)
            t0 = time.perf_counter()
            for _ in range(3):
                y = x @ w.T
            dt = (time.perf_counter() - t0) / 3
        finally:
            if ctx is not None:
                ctx.unregister() if hasattr(ctx, "unregister") else None
        print(f"M={M:5d} threads={nt:4d}  {dt * 1e3:8.1f} ms  {2.0 * M * N * K / dt / 1e12:6.2f} TFLOP/s", flush=True)
