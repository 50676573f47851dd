# rocprofv3 kernel stats of the LLM.int8 cached step + 64 decode steps: bash tools/prof_int8.sh <outdir-name>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
python $R/tools/int8_profile.py > $OUT/i8_plain.txt 2> $OUT/i8_plain.err
rocprofv3 --kernel-trace --stats -d $OUT/prof_i8 -o b -f csv -- python $R/tools/int8_profile.py > $OUT/i8.txt 2> $OUT/i8.err
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/prof_i8/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:45]:
    print(f'{r["Name"][:120]:120s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:9.2f} tot_ms={float(r["TotalDurationNs"])/1e6:8.2f}')
PY
grep -h "ttft\|decode" $OUT/i8_plain.txt | tail -4
