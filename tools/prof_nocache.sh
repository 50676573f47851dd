# rocprofv3 kernel stats of a no-cache prefill: bash tools/prof_nocache.sh <outdir-name> <T>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/prof$2 -o b -f csv -- python $R/tools/nocache_profile.py $2 > $OUT/nocache_$2.txt 2> $OUT/nocache_$2.err
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/prof$2/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
with open("$OUT/nocache_$2_kernel_stats.txt","w") as o:
    for r in rows[:14]:
        line=f'{r["Name"][:100]:100s} calls={r["Calls"]:>6s} total_ms={float(r["TotalDurationNs"])/1e6:9.2f} avg_us={float(r["AverageNs"])/1e3:9.2f} pct={r["Percentage"]}'
        print(line); o.write(line+"\n")
PY
grep pass $OUT/nocache_$2.txt
