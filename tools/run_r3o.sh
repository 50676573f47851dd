cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3o
python -c "import __graft_entry__ as g; g.build(force=True)" > gpurun_out/r3o/build.log 2>&1; echo build=$?
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3o/pytest.log 2>&1; echo rc=$?
tail -4 gpurun_out/r3o/pytest.log
for nr in 0 1; do
PC_ATTN_NO_RING=$nr timeout 900 python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3o/c4_nr$nr.json 2> gpurun_out/r3o/c4_nr$nr.err
PC_ATTN_NO_RING=$nr timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-library > gpurun_out/r3o/b_nr$nr.json 2> gpurun_out/r3o/b_nr$nr.err
python3 - <<PY
import json
c=json.loads(open("gpurun_out/r3o/c4_nr$nr.json").read().strip().split("\n")[-1])
d=json.loads(open("gpurun_out/r3o/b_nr$nr.json").read().strip().split("\n")[-1])
print("no_ring=$nr config4", c["ms_per_step"], "nocache", c.get("no_cache",{}).get("ttft_ms"), "| default", d["ms_per_step"], "encode", d["encode"]["tokens_per_s"], "nocache", d.get("no_cache",{}).get("ttft_ms"))
PY
done
