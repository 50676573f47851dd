cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "shared_prefix or ragged_past" > gpurun_out/r3k/pytest_k.log 2>&1; echo rc_k=$?
tail -5 gpurun_out/r3k/pytest_k.log
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_sharded_encode.py -x -q -m gpu -s -k "suffix or trunk or ragged or encode or shard" > gpurun_out/r3k/pytest_e.log 2>&1; echo rc_e=$?
grep -i "in-place\|passed\|failed\|error" gpurun_out/r3k/pytest_e.log | tail -8
for ip in 1 0; do
PC_PREFIX_IN_PLACE=$ip timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-context > gpurun_out/r3k/bench_ip$ip.json 2> gpurun_out/r3k/bench_ip$ip.err; echo rc=$?
python3 - <<PY
import json
d=json.loads(open("gpurun_out/r3k/bench_ip$ip.json").read().strip().split("\n")[-1])
print("in_place=$ip", d["ms_per_step"], d["encode"]["tokens_per_s"], d["encode"]["seconds"], d["encode_library"]["tokens_per_s"], d["encode_library"].get("seconds"))
PY
done
