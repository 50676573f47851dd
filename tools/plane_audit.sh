# Round 6 plane audit in one gpurun call: bash tools/plane_audit.sh <outdir-name>
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT
python tools/plane_audit.py mid 2>&1 | grep -v amdgpu.ids | tee $OUT/audit.txt
for V in "-DPC_WIDE_NOQLO=1" "-DPC_WIDE_NOPLO=1" "-DPC_WIDE_NOQLO=1 -DPC_WIDE_NOPLO=1" ""; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form $V -c prompt-cache_amd/csrc/pc_attn_wide.hip -o prompt-cache_amd/csrc/_build/pc_attn_wide.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC prompt-cache_amd/csrc/_build/*.o -o prompt-cache_amd/promptcache_amd/libpromptcache_hip.so
  python tools/plane_audit.py attn "${V:-product}" 2>&1 | grep -v amdgpu.ids | grep 'attention over' | tee -a $OUT/audit.txt
  echo "   $(python tools/attn_mid.py 40 8258 259 2>&1 | grep -v amdgpu.ids)" | tee -a $OUT/audit.txt
done
