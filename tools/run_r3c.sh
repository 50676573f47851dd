cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
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
for sk in 8 16 24 32 48; do run skew$sk PC_ATTN_FUSED=0 PC_GEMM_KSKEW=$sk; done
run prioalt PC_ATTN_FUSED=0 PC_GEMM_PRIO_ALT=1
run base2 PC_ATTN_FUSED=0
for sk in 16 32; do PC_GEMM_KSKEW=$sk python tools/gemm_trace.py 12 2>&1 | grep "workgroups,\|by wave\|done (stores"; done
PC_GEMM_PRIO_ALT=1 python tools/gemm_trace.py 12 2>&1 | grep "workgroups,\|by wave\|done (stores"
