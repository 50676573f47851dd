# rocprofv3 kernel stats of the timed step only (context legs off): bash tools/prof_bench.sh <outdir-name>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/prof -o b -f csv -- python $R/bench.py --steps 40 --warmup 5 --no-context --no-cpu-baseline --no-library > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
with open("$OUT/kernel_stats.txt","w") as o:
    for r in rows[:16]:
        line=f'{r["Name"][:110]:110s} calls={r["Calls"]:>6s} total_ms={float(r["TotalDurationNs"])/1e6:9.2f} avg_us={float(r["AverageNs"])/1e3:9.2f} pct={r["Percentage"]}'
        print(line); o.write(line+"\n")
PY
