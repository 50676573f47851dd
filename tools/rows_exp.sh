# dev: rebuild pc_gemm_rows.hip with extra hipcc flags on the GPU box and time tools/rows_bench.py (timing attribution probes):
#   bash tools/rows_exp.sh <outdir-name> "-DPC_ROWS_EXP=1" ... ""      ("" = the product build; run it LAST so the .so on the box is the product's)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for V in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form -Iinclude $V -c prompt-cache_amd/csrc/pc_gemm_rows.hip -o prompt-cache_amd/csrc/_build/pc_gemm_rows.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC prompt-cache_amd/csrc/_build/*.o -o prompt-cache_amd/promptcache_amd/libpromptcache_hip.so
  echo "== flags: $V" | tee -a $OUT/exp.txt
  timeout 300 python tools/rows_bench.py 13b ${ROWS:-256 259} 2>&1 | grep -v amdgpu.ids | tee -a $OUT/exp.txt
done
