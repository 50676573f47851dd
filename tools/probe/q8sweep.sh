cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_q8.py -x -q 2>&1 | tail -5
python tools/q8_micro.py --rows 1 --which down 2>&1 | grep -v amdgpu.ids | tail -3
python tools/int8_profile.py 2>&1 | grep -v amdgpu.ids | tail -5
