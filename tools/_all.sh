R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02
cd $R
bash tools/collect_r02.sh > gpurun_out/r02/collect.log 2>&1
for c in 2 3 4; do timeout 900 python bench.py --config $c > gpurun_out/r02/bench_config$c.json 2> gpurun_out/r02/bench_config$c.err; echo "config $c rc $?"; done
tail -2 gpurun_out/r02/collect.log
