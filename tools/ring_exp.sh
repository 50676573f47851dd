# dev: rebuild pc_attn_ring.hip with extra hipcc flags on the GPU box and time tools/attn_mid.py (timing attribution probes):
#   bash tools/ring_exp.sh "-DPC_RING_EXP=1" "-fno-slp-vectorize" ...      ("" = the product build)
cd $GRAFT_REPO_ROOT
for V in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form $V -c prompt-cache_amd/csrc/pc_attn_ring.hip -o prompt-cache_amd/csrc/_build/pc_attn_ring.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC prompt-cache_amd/csrc/_build/*.o -o prompt-cache_amd/promptcache_amd/libpromptcache_hip.so
  echo "== flags: $V"
  timeout 300 python tools/attn_mid.py 40 8258 256
  timeout 300 python tools/attn_mid.py 40 8258 259
  timeout 300 python tools/attn_mid.py 32 1727 258
done
