# Round-5 evidence in one gpurun call: bash tools/collect_r05.sh   (writes gpurun_out/r05/*; summaries are copied to profiles/r05_* by hand)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
for c in 2 3 4; do python bench.py --config $c --steps 20 --warmup 3 2>/dev/null >> $OUT/configs_2_3_4.jsonl; done
python bench.py --plan-only > $OUT/plan_only.json 2>/dev/null
bash tools/prof_bench.sh r05/bench > $OUT/bench_kernel_stats.log 2>&1
bash tools/prof_config.sh r05/c4 4 > $OUT/c4_kernel_stats.log 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p $OUT/enc; TOPN=16 bash tools/prof_encode.sh r05/enc 300 > $OUT/encode_kernel_stats.txt 2>&1
rm -rf $OUT/bench/prof $OUT/c4/prof_c4 $OUT/enc
ls -la $OUT $OUT/bench $OUT/c4
