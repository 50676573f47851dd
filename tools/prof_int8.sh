# rocprofv3 kernel stats of the LLM.int8 cached step: bash tools/prof_int8.sh <outdir-name>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/prof_i8 -o b -f csv -- python $R/tools/int8_profile.py > $OUT/i8.txt 2> $OUT/i8.err
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/prof_i8/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:40]:
    if int(r["Calls"]) % 32 == 0 and int(r["Calls"]) in (256, 288, 512, 576, 864, 1152):
        print(f'{r["Name"][:105]:105s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:9.2f}')
PY
grep ttft $OUT/i8.txt | tail -1
