# Round-2 evidence on the GPU box: default bench line, rocprofv3 kernel stats of the timed step and of the encode, PMC traffic.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd $R
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
timeout 600 bash tools/prof_bench.sh r02/step > $OUT/prof_step.log 2>&1; echo "prof rc $?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/enc -o enc -f csv -- python $R/tools/encode_profile.py > $OUT/enc.log 2>&1; echo "enc rc $?"
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/enc/**/*kernel_stats.csv", recursive=True)[0]
with open("$OUT/encode_kernel_stats.txt","w") as o:
    for r in list(csv.DictReader(open(f)))[:18]:
        o.write(f'{r["Name"][:110]:110s} calls={r["Calls"]:>6s} total_ms={float(r["TotalDurationNs"])/1e6:9.2f} avg_us={float(r["AverageNs"])/1e3:9.2f} pct={r["Percentage"]}\n')
PY
cd $R
timeout 900 python tools/pmc_traffic.py > $OUT/pmc_traffic.log 2>&1; echo "pmc rc $?"
cp gpurun_out/pmc_traffic.json $OUT/ 2>/dev/null
tail -3 $OUT/bench_default.err
