cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06d
python tools/wide_debug.py 40 8258 259 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee gpurun_out/r06d/dbg.txt
NAN_TAIL=1 python tools/wide_debug.py 40 8258 259 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee -a gpurun_out/r06d/dbg.txt
PC_ATTN_NO_WIDE=1 NAN_TAIL=1 python tools/wide_debug.py 40 8258 259 2>&1 | grep -v amdgpu.ids | cut -c1-400| tee -a gpurun_out/r06d/dbg.txt
