"""Ceiling of a weight-streaming launch by geometry (tools/probe/stream_probe.hip): in-graph time of a read-only kernel of
NWG workgroups x NW waves with U 1-KiB loads in flight per wave over buffers of the projection sizes (rotating through > 256 MB).
python tools/stream_probe.py"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = "/tmp/stream_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                       os.path.join(ROOT, "tools/probe/stream_probe.hip"), "-o", so])
lib = C.CDLL(so)
lib.probe_launch.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
DEV = "cuda:0"
sink = torch.zeros(1 << 16, dtype=torch.int32, device=DEV)
pool = torch.randint(0, 2 ** 31 - 1, (1 << 28,), dtype=torch.int32, device=DEV)      # 1 GiB


def timeit(fn, iters=64):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(4):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


for name, mb in (("o_proj", 33.554432), ("down", 90.177536), ("qkv", 100.663296), ("gate_up", 180.355072)):
    nbytes = int(mb * 1e6) // (1 << 20) * (1 << 20)
    ncopy = max(2, (1 << 30) // nbytes)
    rows = []
    for nwg in (256, 512):
        for nw in (4, 8, 16):
            for U in (4, 8, 16, 32):
                for nt in (1, 0):
                    if nt == 0 and not (nw == 8 and nwg == 256):
                        continue
                    i = [0]

                    def fn():
                        i[0] = (i[0] + 1) % ncopy
                        rc = lib.probe_launch(pool.data_ptr() + i[0] * nbytes, nbytes, nwg, nw, U, nt, sink.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream)
                        assert rc == 0
                    t = timeit(fn)
                    rows.append((t, nwg, nw, U, nt))
    rows.sort()
    print(f"{name} {nbytes / 1e6:.1f} MB: best " + " | ".join(f"{t:.2f}us wg{a} w{b} U{c}{'' if d else ' plain'}" for t, a, b, c, d in rows[:6]),
          flush=True)
    ref = [r for r in rows if r[1:] == (256, 8, 4, 1)][0]
    print(f"   wg256 w8 U4 nt: {ref[0]:.2f} us; worst {rows[-1][0]:.2f} us ({rows[-1][1:]})  -> {nbytes / rows[0][0] / 1e6:.2f} TB/s at best", flush=True)
