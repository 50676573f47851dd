"""HBM traffic per launch of the pc_gemm_q8 kernels from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel trace only)
over tools/int8_profile.py; gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> doubled (guide, HBM
section); units KiB.  Prints bytes per launch next to the algorithmic bytes (the int8 weight image once).
    python tools/q8_pmc.py       (on the GPU box)"""
import collections
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "q8_pmc")
os.makedirs(OUT, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp", PC_I8_LOOP="0")     # (the device loop's pinned ring + rocprofv3 --pmc: the profiler crashed)
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(OUT, ctr)
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "p", "-f", "csv", "--", sys.executable,
                    os.path.join(ROOT, "tools", "int8_profile.py")], cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    vals[ctr] = agg
hid, inter = 4096, 11008
alg = {"gemm_q8p_kernel<3, 3, 1": ("q|k|v  1 row", 3 * hid * hid), "gemm_q8p_kernel<3, 3, 3": ("q|k|v 12 rows", 3 * hid * hid),
       "gemm_q8p_kernel<3, 2, 1": ("gate|up  1 row", 2 * inter * hid), "gemm_q8p_kernel<3, 2, 3": ("gate|up 12 rows", 2 * inter * hid),
       "gemm_q8p_kernel<1, 1, 2": ("o_proj  1 row (partials)", hid * hid), "gemm_q8p_kernel<1, 1, 3": ("o_proj 12 rows", hid * hid),
       "gemm_q8c_kernel<1, 11>": ("down  1 row (C form)", hid * inter), "gemm_q8f_kernel<4, 2>": ("down 12 rows (F form)", hid * inter)}
print("kernel | launches | HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) | algorithmic (int8 weights once) | ratio")
for sub, (what, nb) in alg.items():
    f = [v for k in vals["FETCH_SIZE"] if sub in k for v in vals["FETCH_SIZE"][k]]
    w = [v for k in vals["WRITE_SIZE"] if sub in k for v in vals["WRITE_SIZE"][k]]
    if not f:
        continue
    b = (2 * sum(f) / len(f) + sum(w) / max(len(w), 1)) * 1024
    print(f"{sub:28s} {what:26s} {len(f):6d} {b / 1e6:9.2f} MB {nb / 1e6:9.2f} MB  {b / nb:5.2f}")
