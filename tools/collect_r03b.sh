# Round-3 evidence, second half of the round (ring attention, in-place trunk, fused q|k|v epilogue): bash tools/collect_r03b.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03b
mkdir -p $OUT
cd $R
timeout 900 python tools/pmc_traffic.py > $OUT/pmc_traffic.log 2>&1; echo "pmc rc $?"
cp gpurun_out/pmc_traffic.json $OUT/ 2>/dev/null
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
timeout 600 bash tools/prof_bench.sh r03b/step > $OUT/prof_step.log 2>&1; echo "prof rc $?"
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
timeout 600 bash tools/prof_config.sh r03b 4 > $OUT/prof_c4.log 2>&1; echo "prof c4 rc $?"
timeout 600 bash tools/pmc_attn_mid.sh r03b/pmc_ring 40 8258 259 > $OUT/pmc_ring.txt 2>&1; echo "pmc ring rc $?"
timeout 600 bash tools/decode_prof.sh > $OUT/decode.log 2>&1; cp gpurun_out/decode_prof/decode_kernel_stats.txt $OUT/ 2>/dev/null
tail -2 $OUT/bench_default.err
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().split("\n")[-1])
print("ms_per_step", d["ms_per_step"], "roofline", d["roofline"]["frac"], "step", d["roofline_step"]["frac"], "decode", d.get("decode",{}).get("tokens_per_s"), d.get("decode_device_loop",{}).get("tokens_per_s"), "encode", d["encode"]["tokens_per_s"], d["encode"]["roofline"]["frac"], "lib", d["encode_library"]["tokens_per_s"], "parity", d.get("parity"), "nocache", d.get("no_cache",{}).get("ttft_ms"), "int8", d.get("int8_weights",{}).get("ttft_ms"))
for l in open("$OUT/configs.jsonl"):
    c=json.loads(l); print(c["config"]["workload"][:40], c["ms_per_step"], c["roofline"]["frac"], c.get("roofline_attention",{}).get("avg_launch_us"), c.get("roofline_attention",{}).get("frac"))
PY
