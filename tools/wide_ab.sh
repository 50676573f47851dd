# Round 6: the wide staged-key attention against the ring kernel on config 4's launch shape (same box), parity first.
#   bash tools/wide_ab.sh <outdir-name>
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -s 2>&1 | tail -15 > $OUT/parity.txt
for shape in "40 8258 259" "40 8258 256" "32 1727 100" "32 4390 66" "40 8258 320" "40 8258 130"; do
  for nw in 1 0; do
    echo "== PC_ATTN_NO_WIDE=$nw shape $shape" >> $OUT/ab.txt
    PC_ATTN_NO_WIDE=$nw timeout 300 python tools/attn_mid.py $shape >> $OUT/ab.txt 2>&1
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o w -- python $GRAFT_REPO_ROOT/tools/attn_mid.py 40 8258 259 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) > $OUT/kernel_stats.txt 2>&1
rm -rf $OUT/prof
cat $OUT/parity.txt $OUT/ab.txt; head -12 $OUT/kernel_stats.txt
