cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3q
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "dense" > gpurun_out/r3q/pytest_k.log 2>&1; echo rc_k=$?
tail -15 gpurun_out/r3q/pytest_k.log
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_sharded_encode.py -x -q -m gpu > gpurun_out/r3q/pytest_e.log 2>&1; echo rc_e=$?
tail -5 gpurun_out/r3q/pytest_e.log
for f in 1 0; do
PC_FUSED_DENSE_QKV=$f timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-library > gpurun_out/r3q/b_f$f.json 2> gpurun_out/r3q/b_f$f.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/r3q/b_f$f.json").read().strip().split("\n")[-1])
print("fused=$f", d["ms_per_step"], "encode", d["encode"]["tokens_per_s"], d["encode"]["seconds"], "nocache", d.get("no_cache",{}).get("ttft_ms"))
PY
done
