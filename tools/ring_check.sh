cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ring_check
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ring or shared_prefix or ragged_past or attn" > gpurun_out/ring_check/pytest_k.log 2>&1; echo rc_k=$?
tail -30 gpurun_out/ring_check/pytest_k.log
for nr in 0 1; do PC_ATTN_NO_RING=$nr timeout 300 python tools/attn_mid.py; PC_ATTN_NO_RING=$nr timeout 300 python tools/attn_mid.py 40 8258 256;  PC_ATTN_NO_RING=$nr timeout 300 python tools/attn_mid.py 32 1727 258; done
