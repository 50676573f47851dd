cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r2p
mkdir -p $OUT
cd $R && timeout 600 python -m pytest tests/test_gpu_int8.py -x -q 2>&1 | tail -2
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_i8 -o b -f csv -- python $R/tools/int8_profile.py > $OUT/i8.txt 2> $OUT/i8.err
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/prof_i8/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:40]:
    if "anonymous" in r["Name"] or "_GLOBAL__" in r["Name"]:
        print(f'{r["Name"][:105]:105s} calls={r["Calls"]:>6s} total_ms={float(r["TotalDurationNs"])/1e6:9.2f} avg_us={float(r["AverageNs"])/1e3:9.2f}')
PY
grep ttft $OUT/i8.txt | tail -2
