"""Collect HBM traffic per launch of the timed step's kernels from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
never combined with trace domains) and write profiles/pmc_traffic.json, which bench.py reads for its `traffic` fields.
gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> doubled (guide, HBM section); units: KiB.

    python tools/pmc_traffic.py            (on the GPU box; runs bench.py --no-context twice under rocprofv3)
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "pmc_traffic")
os.makedirs(OUT, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(OUT, ctr)
    if "--reuse" not in sys.argv:                       # (--reuse: only re-read the CSVs of an earlier run)
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "p", "-f", "csv", "--", sys.executable,
                        os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-context", "--no-cpu-baseline",
                        "--no-library"], cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    vals[ctr] = agg


def pick(sub, largest=False):
    """Bytes per launch of the kernels whose name contains ``sub``: the mean over all their dispatches (several template
    variants of one kernel family are pooled), or -- ``largest`` -- the mean over the dispatches within 10 % of the biggest
    one (kv_copy_kernel also runs the many small slice-store launches of the encode; the gather is the big one)."""
    out = {}
    for ctr in vals:
        xs = [v for k in vals[ctr] if sub in k for v in vals[ctr][k]]
        if not xs:
            return None
        if largest:
            top = max(xs)
            xs = [v for v in xs if v >= 0.9 * top]
        out[ctr] = sum(xs) / len(xs)
    return int((2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024)


src = "tools/pmc_traffic.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --no-context; (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch"
tab = {
    "kv_copy_kernel": {"signature": "S=1725,L=32,Hkv=32,D=128", "hbm_bytes_per_launch": pick("kv_copy_kernel", largest=True), "source": src},
    "gemm_skinny_add": {"signature": "T=12,hid=4096,inter=11008",
                        "hbm_bytes_per_launch": (lambda o, d: (o + d) // 2 if o and d else (o or d))(pick("gemm_skinny_kernel<1, 1, 1"),
                                                                                                    pick("gemm_skinny_ks_kernel")),
                        "source": src + " (average of o_proj = gemm_skinny_kernel<1,1,EPI_ADD> and down_proj = gemm_skinny_ks_kernel)"},
    "gemm_skinny_gate_up": {"signature": "T=12,hid=4096,inter=11008", "hbm_bytes_per_launch": pick("gemm_skinny_kernel<1, 3, 2"), "source": src},
}
c = pick("attn_combine_kernel")
# the timed step's attention since round 4: the staging variant (module K/V read once, staged rows written as they pass)
a = pick("attn_small_kernel<128, false, 0, true, 1>")
tab["attn_staging"] = {"signature": "H=32,Hkv=32,D=128,q=12,S=1725", "hbm_bytes_per_launch": (a + c) if a and c else None,
                       "source": src + " (attn_small_kernel<128, false, 0, true> = pc_attn gather_rows + attn_combine_kernel)"}
a = pick("attn_small_kernel<128, false, 0, false, 1>")          # (only present when the run includes steps that stage by pc_kv_gather)
if a and c:
    tab["attn_cached"] = {"signature": "H=32,Hkv=32,D=128,q=12,S=1725", "hbm_bytes_per_launch": a + c,
                          "source": src + " (attn_small_kernel + attn_combine_kernel)"}
tab["gemm_skinny_qkv"] = {"signature": "T=12,hid=4096,inter=11008", "hbm_bytes_per_launch": pick("gemm_skinny_kernel<1, 3, 3"),
                          "source": src + " (q|k|v + RMSNorm + RoPE + KV append)"}
# --config4: one more pair of passes over bench.py --config 4 for the ring attention of its step (13b, 259 rows over 8 258 keys)
if "--config4" in sys.argv:
    v4 = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(OUT, "c4_" + ctr)
        if "--reuse" not in sys.argv:
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "p", "-f", "csv", "--", sys.executable,
                            os.path.join(ROOT, "bench.py"), "--config", "4", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                           cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        rows_ = list(csv.DictReader(open(f)))
        # round 6: the wide staged-key kernel (staging variant) + the ring kernel over the pass's own rows (split merge not included)
        per = {}
        # (the own-rows launch is told from the schema encode's ring launches of the same run by its grid: 40 heads x 3 blocks of 128 rows x 512 threads)
        for sub, grid in (("attn_wide_kernel<true", None), ("attn_ring_kernel<", str(40 * 3 * 512))):
            xs = [float(r["Counter_Value"]) for r in rows_ if sub in r["Kernel_Name"] and (grid is None or r["Grid_Size"] == grid)]
            per[sub] = sum(xs) / len(xs) if xs else None
        v4[ctr] = per
    if all(v4[c][k] is not None for c in v4 for k in v4[c]):
        tot = {c: sum(v4[c].values()) for c in v4}
        tab["attn_wide_staging"] = {"signature": "H=40,Hkv=40,D=128,q=259,S=8258",
                                    "hbm_bytes_per_launch": int((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024),
                                    "parts_KiB": v4,
                                    "source": src.replace("bench.py --no-context", "bench.py --config 4") +
                                    " (attn_wide_kernel<true, 3> over the staged keys, staging them as they pass, + attn_ring_kernel over the pass's own rows = one pc_attn call; the split merge not included)"}
# keep entries of earlier rounds that this run did not re-measure (e.g. attn_cached: the copy-first step)
try:
    old = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for k_, v_ in old.items():
        if k_ not in tab or not tab[k_].get("hbm_bytes_per_launch"):
            tab[k_] = v_
except (OSError, ValueError):
    pass
json.dump(tab, open(os.path.join(ROOT, "gpurun_out", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(tab, indent=1))
