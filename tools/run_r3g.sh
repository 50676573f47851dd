cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
timeout 1700 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py tests/test_gpu_sharded_encode.py -x -q -s -k "row_bucketed or int8_weight_mode_matches or full_depth_13b or full_depth_codellama or outlier_feature or sharded or add_schemas or library" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -n "full depth\|logits beyond\|\[int8\|passed\|failed\|Error\|rc=" $O/pytest.log | tail -40
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-library --no-int8"
run() { name=$1; shift; env "$@" timeout 900 $B > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().split("\n")[-1]); print("$name", round(d["ms_per_step"],4), "cold", [(c.get("new_tokens"), round(c.get("first_call_ms",0),2), round(c.get("warm_ms",0),2)) for c in d.get("cold_shape_ttft",[])] if isinstance(d.get("cold_shape_ttft"), list) else d.get("cold_shape_ttft"))
except Exception as e:
    print("$name FAILED", e)
PY
}
run bucket4 PC_GRAPH_BUCKET=4
run bucket16 PC_GRAPH_BUCKET=16
run bucket0 PC_GRAPH_BUCKET=0
