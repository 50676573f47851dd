cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_q8.py -x -q 2>&1 | tail -8
python tools/int8_profile.py 2>&1 | grep -v amdgpu.ids | tail -5
echo "--- PC_Q8_DEFER_MERGE=0"
PC_Q8_DEFER_MERGE=0 python tools/int8_profile.py 2>&1 | grep -v amdgpu.ids | tail -5
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -k int8 2>&1 | tail -4
