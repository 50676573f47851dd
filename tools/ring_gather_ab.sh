# where the staging ring launch loses time against the plain one: dev builds of pc_attn_ring.hip on the GPU box, timing only
# (bash tools/ring_gather_ab.sh): default | plain addressing, entries still fetched | plain addressing, no entry DMA
cd $GRAFT_REPO_ROOT
for flags in "" "-DPC_RING_G_PLAINADDR" "-DPC_RING_G_PLAINADDR -DPC_RING_G_NOFETCH"; do
  touch prompt-cache_amd/csrc/pc_attn_ring.hip
  PC_BUILD_FLAGS="$flags" python __graft_entry__.py > /dev/null 2>&1
  echo "== flags: $flags"
  python tools/ring_gather_micro.py 2>&1 | grep -E "plain|already|stages"
done
touch prompt-cache_amd/csrc/pc_attn_ring.hip
python __graft_entry__.py > /dev/null 2>&1
