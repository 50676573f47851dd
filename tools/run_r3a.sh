cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "single_launch or ks_in_launch or attn_matches or attn_fragment or gemm_skinny_norm or qkv_rope or attn_softmax" > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
tail -5 gpurun_out/r3a/pytest.log
for f in 0 1; do timeout 120 python tools/attn_micro.py 1725 12 1 300 $f; done > gpurun_out/r3a/attn_micro.log 2>&1
PC_ATTN_SMALL_WG=7 timeout 120 python tools/attn_micro.py 1725 12 1 300 1 >> gpurun_out/r3a/attn_micro.log 2>&1
PC_ATTN_SMALL_WG=7 timeout 120 python tools/attn_micro.py 1725 12 1 300 0 >> gpurun_out/r3a/attn_micro.log 2>&1
timeout 120 python tools/attn_micro.py 1725 1 0 300 1 >> gpurun_out/r3a/attn_micro.log 2>&1
timeout 120 python tools/attn_micro.py 1725 1 0 300 0 >> gpurun_out/r3a/attn_micro.log 2>&1
cat gpurun_out/r3a/attn_micro.log
timeout 300 python tools/ks_micro.py 12 > gpurun_out/r3a/ks_micro.log 2>&1; cat gpurun_out/r3a/ks_micro.log
timeout 600 bash tools/prof_bench.sh r3a/fused
PC_ATTN_FUSED=0 timeout 600 bash tools/prof_bench.sh r3a/unfused
cat gpurun_out/r3a/fused/bench.json | head -c 600
