cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wide
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -x -q -m gpu > gpurun_out/wide/pytest.log 2>&1; echo rc=$?
tail -5 gpurun_out/wide/pytest.log
for cfg in "PC_ROWS_QKV_KS=2" "PC_ROWS_QKV_KS=1" "PC_ROWS_QKV_KS=2 PC_ROWS_KQ=5" "PC_ROWS_QKV_KS=2 PC_ROWS_KQ=4" "PC_ROWS_WIDE=0"; do
env $cfg timeout 900 python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/wide/c4.json 2> gpurun_out/wide/c4.err
python3 - <<PY
import json
c=json.loads(open("gpurun_out/wide/c4.json").read().strip().split("\n")[-1])
print("$cfg", "config4", c["ms_per_step"])
PY
done
