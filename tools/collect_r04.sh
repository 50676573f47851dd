# Round-4 evidence in one gpurun call: bash tools/collect_r04.sh   (writes gpurun_out/r04/*)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04; mkdir -p $OUT
python tools/encode_rate_curve.py > $OUT/encode_rate_curve.json 2> $OUT/encode_rate_curve.log
for c in 2 3 4; do python bench.py --config $c --steps 20 --warmup 3 2>/dev/null >> $OUT/configs_2_3_4.jsonl; done
mkdir -p $OUT/enc; TOPN=16 bash tools/prof_encode.sh r04/enc 300 > $OUT/encode_kernel_stats.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_c4 -o b -f csv -- python $R/bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/$OUT/prof_c4/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
with open("$R/$OUT/config4_kernel_stats.txt","w") as o:
    for r in rows[:18]:
        o.write(f'{r["Name"][:100]:100s} calls={r["Calls"]:>6s} total_ms={float(r["TotalDurationNs"])/1e6:9.2f} avg_us={float(r["AverageNs"])/1e3:9.2f} pct={r["Percentage"]}\n')
PY
rm -rf $R/$OUT/prof_c4 $R/$OUT/enc
