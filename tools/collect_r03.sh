# Round-3 evidence on the GPU box: default bench line, rocprofv3 kernel stats of the timed step and of the encode, PMC traffic,
# configs 2-4.   bash tools/collect_r03.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout 900 python tools/pmc_traffic.py > $OUT/pmc_traffic.log 2>&1; echo "pmc rc $?"
cp gpurun_out/pmc_traffic.json $OUT/ 2>/dev/null; cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
timeout 600 bash tools/prof_bench.sh r03/step > $OUT/prof_step.log 2>&1; echo "prof rc $?"
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
for c in 2 3 4; do timeout 900 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline >> $OUT/configs.jsonl 2>> $OUT/configs.err; done; echo "configs rc $?"
tail -2 $OUT/bench_default.err
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().split("\n")[-1])
print("ms_per_step", d["ms_per_step"], "roofline", d["roofline"]["frac"], "step", d["roofline_step"]["frac"], "decode", d.get("decode",{}).get("tokens_per_s"), d.get("decode_device_loop",{}).get("tokens_per_s"), "encode", d["encode"]["tokens_per_s"], d["encode"]["roofline"]["frac"], "parity", d.get("parity"))
for l in open("$OUT/configs.jsonl"):
    c=json.loads(l); print(c["config"]["workload"][:40], c["ms_per_step"], c["roofline"]["frac"])
PY
