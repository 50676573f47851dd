cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "shared_prefix or ragged_past" > gpurun_out/r3l/pytest_k.log 2>&1; echo rc_k=$?
tail -3 gpurun_out/r3l/pytest_k.log
for rep in 1 2; do for ip in 1 0; do
PC_PREFIX_IN_PLACE=$ip timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-context --no-library > gpurun_out/r3l/bench_ip$ip.json 2> gpurun_out/r3l/bench_ip$ip.err; echo rc=$?
python3 - <<PY
import json
d=json.loads(open("gpurun_out/r3l/bench_ip$ip.json").read().strip().split("\n")[-1])
print("in_place=$ip", d["ms_per_step"], d["encode"]["tokens_per_s"], d["encode"]["seconds"])
PY
done; done
bash tools/prof_config.sh r3l 4 > gpurun_out/r3l/prof4.log 2>&1; tail -24 gpurun_out/r3l/prof4.log
