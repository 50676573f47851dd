# dev: rebuild pc_attn_ring.hip with -DPC_RING_EXP=<n> on the GPU box and time tools/attn_mid.py (timing attribution probes)
cd $GRAFT_REPO_ROOT
for V in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form -DPC_RING_EXP=$V -c prompt-cache_amd/csrc/pc_attn_ring.hip -o prompt-cache_amd/csrc/_build/pc_attn_ring.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC prompt-cache_amd/csrc/_build/*.o -o prompt-cache_amd/promptcache_amd/libpromptcache_hip.so
  echo "== PC_RING_EXP=$V"
  timeout 300 python tools/attn_mid.py 40 8258 256
  timeout 300 python tools/attn_mid.py 40 8258 259
done
