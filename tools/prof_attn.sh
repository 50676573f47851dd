cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
run() {  # name, env..., args
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats -d $OUT/$name -o a -f csv -- python $R/tools/attn_micro.py $ARGS > $OUT/$name.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$OUT/$name/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "attn" in r["Name"]: print("$name", r["Name"][:70], r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,2))
PY
}
ARGS="1725 12 1 300"
run base X=1
run nosmall PC_ATTN_NO_SMALL=1
run wg4 PC_ATTN_SMALL_WG=4
run wg7 PC_ATTN_SMALL_WG=7
run wg15 PC_ATTN_SMALL_WG=15
ARGS="1725 12 0 300"
run notail X=1
run notail_nosmall PC_ATTN_NO_SMALL=1
ARGS="4390 14 1 300"
run game X=1
run game_nosmall PC_ATTN_NO_SMALL=1
