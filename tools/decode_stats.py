"""Per-kernel averages of the decode steps out of a rocprofv3 kernel trace of tools/decode_profile.py: the LAST `steps` tokens
(every token replays the same captured graph).  python tools/decode_stats.py <kernel_trace.csv> [steps]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the decode steps are periodic: find the period as the distance between the last two launches of the lm_head-sized kernel
names = [r["Kernel_Name"] for r in rows]
last = names[-1]
idx = [i for i, n in enumerate(names) if n == last]
period = idx[-1] - idx[-2]
tail = rows[-period * steps:]
agg = collections.OrderedDict()
for r in tail:
    k = (r["Kernel_Name"][:96], r.get("Grid_Size", "?"))
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e3 / steps
busy = sum(v[1] for v in agg.values()) / steps
print(f"# {period} launches per token; {span:.1f} us per token wall, {busy:.1f} us in kernels, {span - busy:.1f} us between them")
for (name, grid), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:96s} grid={grid:>8s} per_token={n / steps:6.1f} avg_us={t / n:8.2f} us_per_token={t / steps:9.1f}")
