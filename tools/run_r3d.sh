cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
B="python bench.py --steps 100 --warmup 10 --no-context --no-cpu-baseline --no-library"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().split("\n")[-1]); print("$name", round(d["ms_per_step"],4))
except Exception as e:
    print("$name FAILED", e)
PY
}
run base PC_ATTN_FUSED=0
run devkernarg PC_ATTN_FUSED=0 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 PC_ATTN_FUSED=0 HIP_FORCE_DEV_KERNARG=0
run hwq1 PC_ATTN_FUSED=0 GPU_MAX_HW_QUEUES=1
run hwq8 PC_ATTN_FUSED=0 GPU_MAX_HW_QUEUES=8
run base2 PC_ATTN_FUSED=0
env | grep -i "HIP_\|HSA_\|ROC_\|GPU_" | head -20
