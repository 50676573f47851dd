# dev: the wide-panel candidates of gemm_rows_kernel (-DPC_DEV_ROWS_VARIANTS) against the product dispatch: bash tools/rows_var.sh
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form -DPC_DEV_ROWS_VARIANTS -c prompt-cache_amd/csrc/pc_gemm_rows.hip -o prompt-cache_amd/csrc/_build/pc_gemm_rows.hip.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC prompt-cache_amd/csrc/_build/*.o -o prompt-cache_amd/promptcache_amd/libpromptcache_hip.so
for M in 256 259; do
python tools/rows_bench.py 13b $M 2>&1 | grep -v amdgpu
PC_ROWS_VARIANT=A PC_ROWS_KSL=2,6,1,6 python tools/rows_bench.py 13b $M 2>&1 | grep -v amdgpu
PC_ROWS_VARIANT=B PC_ROWS_KSL=1,4,1,4 python tools/rows_bench.py 13b $M 2>&1 | grep -v amdgpu
PC_ROWS_VARIANT=B PC_ROWS_KSL=2,5,1,5 python tools/rows_bench.py 13b $M 2>&1 | grep -v amdgpu
PC_ROWS_VARIANT=C PC_ROWS_KSL=2,6,1,6 python tools/rows_bench.py 13b $M 2>&1 | grep -v amdgpu
done
