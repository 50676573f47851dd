cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/q8prof
rm -rf $OUT; mkdir -p $OUT
python $R/tools/int8_profile.py 2>&1 | grep -v amdgpu.ids | tail -4
for D in 0 1; do
PC_I8_DECODE=$D rocprofv3 --kernel-trace --stats -d $OUT/p$D -o b -f csv -- python $R/tools/int8_profile.py > $OUT/i8_$D.txt 2> $OUT/i8_$D.err
done
python3 - <<PY
import csv,glob
def load(d):
    f=glob.glob("$OUT/p%d/**/*kernel_stats.csv"%d, recursive=True)[0]
    return {r["Name"]:(int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a,b=load(0),load(1)
print("kernel | 12-row avg us (calls) | 1-row avg us (calls)")
for k,(c1,t1) in sorted(b.items(), key=lambda kv:-kv[1][1]):
    c0,t0=a.get(k,(0,0.0))
    if c1-c0 >= 1024 or c0 in (256,288,512,576):
        d = (t1-t0)/(c1-c0)/1e3 if c1>c0 else float('nan')
        print(f'{k[:100]:100s} {t0/max(c0,1)/1e3:8.2f} ({c0:5d}) {d:8.2f} ({c1-c0:5d})')
PY
