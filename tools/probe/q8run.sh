cd $GRAFT_REPO_ROOT
PC_I8_DECODE=1 python tools/int8_profile.py 2>&1 | grep -v amdgpu.ids | tail -6
echo "--- PC_INT8_INLAUNCH=0"
PC_INT8_INLAUNCH=0 PC_I8_DECODE=1 python tools/int8_profile.py 2>&1 | grep -v amdgpu.ids | tail -4
timeout 1500 python -m pytest tests/test_gpu_engine.py -x -q -k "int8" -s 2>&1 | grep -v amdgpu.ids | tail -30
