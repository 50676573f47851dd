cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "single_launch or ks_in_launch" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
B="python bench.py --steps 100 --warmup 10 --no-context --no-cpu-baseline --no-library"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/$name.json 2> $O/$name.err; python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().split("\n")[-1]); print("$name", round(d["ms_per_step"],4))
PY
}
run fused_ns8 PC_ATTN_FUSED=1
run unfused_ns8 PC_ATTN_FUSED=0
run fused_ns7 PC_ATTN_FUSED=1 PC_ATTN_SMALL_WG=7
run unfused_ns7 PC_ATTN_FUSED=0 PC_ATTN_SMALL_WG=7
run fused_ns8_b PC_ATTN_FUSED=1
run unfused_ns8_b PC_ATTN_FUSED=0
run fused_ksd22 PC_ATTN_FUSED=1 PC_KS_DOWN=2,2
run fused_ksd24 PC_ATTN_FUSED=1 PC_KS_DOWN=2,4
run fused_ksd88 PC_ATTN_FUSED=1 PC_KS_DOWN=8,8
run fused_ksd22_o22 PC_ATTN_FUSED=1 PC_KS_DOWN=2,2 PC_KS_O=2,2
run unfused_ksd22 PC_ATTN_FUSED=0 PC_KS_DOWN=2,2
