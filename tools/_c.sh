mkdir -p gpurun_out/r2p
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_chain.py -x -q > gpurun_out/r2p/t_eng.txt 2>&1; tail -5 gpurun_out/r2p/t_eng.txt
for c in 0 1; do PC_CHAIN=$c timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-library > gpurun_out/r2p/bench_chain$c.json 2> gpurun_out/r2p/bench_chain$c.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r2p/bench_chain$c.json").read().strip().splitlines()[-1])
print("PC_CHAIN=$c", "value", d["value"], "ms", d["ms_per_step"], "parity", d.get("parity"), "decode", d.get("decode"), d.get("decode_device_loop"))
PY
done
