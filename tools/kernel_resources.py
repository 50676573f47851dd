"""Per-kernel register / scratch / LDS table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
python tools/kernel_resources.py report.txt [filter]     (report = stderr of a hipcc -c ... -Rpass-analysis=kernel-resource-usage)"""
import re
import subprocess
import sys

rows, cur = [], None
for line in open(sys.argv[1], errors="replace"):
    m = re.search(r"remark: (?:\s*)(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
names = [r["name"] for r in rows]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r, d in zip(rows, dem):
    d = re.sub(r"\(anonymous namespace\)::|pcg::", "", d).split("(")[0].replace("void ", "")
    if flt and flt not in d:
        continue
    print(f"{d:70s} vgpr {r.get('VGPRs', '?'):>4} sgpr {r.get('TotalSGPRs', '?'):>4} scratch {r.get('ScratchSize', '?'):>5} "
          f"occ {r.get('Occupancy', '?'):>2} lds {r.get('LDS', '?'):>6}")
