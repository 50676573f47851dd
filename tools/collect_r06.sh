# Round-6 evidence in one gpurun call: bash tools/collect_r06.sh   (writes gpurun_out/r06/*; summaries are copied to profiles/r06_* by hand)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
python bench.py --plan-only > $OUT/plan_only.json 2>/dev/null
PC_ATTN_NO_WIDE=1 python bench.py --config 4 --steps 20 --warmup 3 2>/dev/null > $OUT/config4_ring_only.json
python bench.py --config 4 --steps 20 --warmup 3 2>/dev/null > $OUT/config4_wide.json
bash tools/prof_bench.sh r06/bench > $OUT/bench_kernel_stats.log 2>&1
bash tools/prof_config.sh r06/c4 4 > $OUT/c4_kernel_stats.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/pmc_attn_mid.sh r06/pmc 40 8258 259 > $OUT/pmc_attn_wide.txt 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p $OUT/enc; TOPN=16 bash tools/prof_encode.sh r06/enc 300 > $OUT/encode_kernel_stats.txt 2>&1
rm -rf $OUT/bench/prof $OUT/c4/prof_c4 $OUT/enc $OUT/pmc/pmc1 $OUT/pmc/pmc2 $OUT/pmc/pmc3
ls -la $OUT $OUT/bench $OUT/c4
