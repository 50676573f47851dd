# rocprofv3 kernel stats of bench.py --config N (cached steps + the no-cache passes): bash tools/prof_config.sh <outdir-name> <N>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/prof_c$2 -o b -f csv -- python $R/bench.py --config $2 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c$2.json 2> $OUT/bench_c$2.err
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/prof_c$2/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
with open("$OUT/config$2_kernel_stats.txt","w") as o:
    for r in rows[:22]:
        line=f'{r["Name"][:100]:100s} calls={r["Calls"]:>6s} total_ms={float(r["TotalDurationNs"])/1e6:9.2f} avg_us={float(r["AverageNs"])/1e3:9.2f} pct={r["Percentage"]}'
        print(line); o.write(line+"\n")
PY
