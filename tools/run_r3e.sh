cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "single_launch or ks_in_launch or attn or gemm_skinny" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-library --no-int8"
run() { name=$1; shift; env "$@" timeout 900 $B > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().split("\n")[-1]); print("$name", round(d["ms_per_step"],4), "decode", {k: round(v,1) for k,v in d.get("decode",{}).items() if k in ("tokens_per_s",)}, "loop", round(d.get("decode_device_loop",{}).get("tokens_per_s",0),1))
except Exception as e:
    print("$name FAILED", e)
PY
}
run base PC_ATTN_FUSED=0
run ksd22 PC_ATTN_FUSED=0 PC_KS_DOWN=2,2
run ksd22_o22 PC_ATTN_FUSED=0 PC_KS_DOWN=2,2 PC_KS_O=2,2
run fused PC_ATTN_FUSED=1 PC_KS_DOWN=2,2
timeout 300 python tools/ks_micro.py 1 2>&1 | grep -v amdgpu
