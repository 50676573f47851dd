# A/B of the staging attention's memory policies inside the real step, same box: bash tools/ab_gather.sh <outdir-name>
# variants: default (plain loads, nt stores) | -DPC_GATHER_LOAD_PLAIN | -DPC_GATHER_STORE_PLAIN | both
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
run() {
  for i in 1 2; do
    python bench.py --steps 40 --warmup 8 --no-context --no-library --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$1', r['ms_per_step'], r['breakdown_ms']['prefill_median'], r['staging']['host_glue_ms'])" | tee -a $OUT/ab.txt
  done
}
run default
for flags in "-DPC_GATHER_LOAD_PLAIN" "-DPC_GATHER_STORE_PLAIN" "-DPC_GATHER_LOAD_PLAIN -DPC_GATHER_STORE_PLAIN"; do
  touch prompt-cache_amd/csrc/pc_attn.hip
  PC_BUILD_FLAGS="$flags" python __graft_entry__.py > /dev/null 2>&1
  run "$flags"
done
touch prompt-cache_amd/csrc/pc_attn.hip
python __graft_entry__.py > /dev/null 2>&1
run default-again
PC_DEFER_GATHER=0 run copy-first
