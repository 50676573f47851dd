# rocprofv3 kernel stats of tools/step_profile.py: bash tools/prof_step.sh <outdir-name> <question words>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/prof_q$2 -o b -f csv -- python $R/tools/step_profile.py $2 > $OUT/step_q$2.txt 2> $OUT/step_q$2.err
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/prof_q$2/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:30]:
    if int(r["Calls"]) % 32 == 0 and int(r["Calls"]) >= 400 and ("anonymous" in r["Name"] or "_GLOBAL__" in r["Name"] or "pcg::" in r["Name"] or "pca::" in r["Name"]):
        print(f'{r["Name"][:105]:105s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:9.2f}')
PY
grep ttft $OUT/step_q$2.txt | tail -1
