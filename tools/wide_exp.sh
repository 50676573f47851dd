# dev: rebuild pc_attn_wide.hip with extra hipcc flags on the GPU box and time tools/attn_mid.py (timing attribution probes):
#   bash tools/wide_exp.sh <outdir-name> "" "-DPC_WIDE_EXP=1" ...      ("" = the product build; run it LAST so the .so on the box is the product's)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for V in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form $V -c prompt-cache_amd/csrc/pc_attn_wide.hip -o prompt-cache_amd/csrc/_build/pc_attn_wide.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC prompt-cache_amd/csrc/_build/*.o -o prompt-cache_amd/promptcache_amd/libpromptcache_hip.so
  echo "== flags: $V" | tee -a $OUT/exp.txt
  for shape in "40 8258 259" "40 8258 256" "40 8258 130" "32 1727 100"; do
    timeout 300 python tools/attn_mid.py $shape 2>&1 | grep -v amdgpu.ids | tee -a $OUT/exp.txt
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o w -- python $GRAFT_REPO_ROOT/tools/attn_mid.py 40 8258 259 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) 2>&1 | head -6 | tee $OUT/kernel_stats.txt
rm -rf $OUT/prof
