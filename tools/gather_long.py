"""pc_kv_gather on ONE long segment at the 13b shape (BASELINE config 4: 8 000 rows, 13.1 GB moved): event-timed sweep of the copy
kernel's tiling (PC_GATHER_TILE / PC_GATHER_PPW / PC_GATHER_UNROLL / PC_GATHER_NT; the library reads them once per process, so
every combination is a child process).  python tools/gather_long.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, torch
sys.path[:0] = [%r, os.path.join(%r, "prompt-cache_amd")]
from promptcache_amd import _native as n
L, Hkv, D, S, cap = 40, 40, 128, 8000, 9186
store = torch.empty((L, 2, Hkv, S, D), dtype=torch.float16, device="cuda").normal_()
arena = torch.empty((L, 2, Hkv, cap, D), dtype=torch.float16, device="cuda")
ts = []
for i in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); n.kv_gather([store.data_ptr()], [S], [0], arena, L, Hkv, D, cap); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = sorted(ts[2:])[len(ts[2:]) // 2]
print(f"{t:.3f} ms  {2 * store.numel() * 2 / t / 1e6:.0f} GB/s")
''' % (ROOT, ROOT)
for tile, ppw, unroll, nt in [(32768, 2, 16, 0), (65536, 1, 16, 0), (131072, 1, 16, 0), (65536, 2, 16, 0), (32768, 1, 16, 0), (262144, 1, 16, 0),
                              (32768, 4, 16, 0), (16384, 2, 16, 0), (65536, 1, 16, 1), (32768, 2, 16, 1)]:
    env = dict(os.environ, PC_GATHER_TILE=str(tile), PC_GATHER_PPW=str(ppw), PC_GATHER_UNROLL=str(unroll), PC_GATHER_NT=str(nt))
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    print(f"tile={tile:7d} planes/wg={ppw} unroll={unroll} nt={nt}: {out[-1] if out else 'failed'}", flush=True)
