cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
timeout 1700 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_int8.py tests/test_gpu_chain.py -x -q > $O/pytest_k.log 2>&1; echo "pytest rc=$?" >> $O/pytest_k.log; tail -3 $O/pytest_k.log
timeout 1700 python -m pytest tests/test_gpu_engine.py -x -q -k "not full_depth" > $O/pytest_e.log 2>&1; echo "pytest rc=$?" >> $O/pytest_e.log; tail -3 $O/pytest_e.log
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-library --no-int8"
run() { name=$1; shift; env "$@" timeout 900 $B > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().split("\n")[-1]); print("$name", round(d["ms_per_step"],4), "cold", d.get("cold_shape_ttft"))
except Exception as e:
    print("$name FAILED", e)
PY
}
run bucket16 PC_GRAPH_BUCKET=16
run bucket0 PC_GRAPH_BUCKET=0
run bucket16b PC_GRAPH_BUCKET=16
