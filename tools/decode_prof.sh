cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/decode_prof
mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/dec -o d -f csv -- python $R/tools/decode_profile.py 96 > $OUT/dec.log 2>&1
tail -3 $OUT/dec.log
python $R/tools/decode_stats.py $(ls $OUT/dec/*kernel_trace.csv $OUT/dec/*/*kernel_trace.csv 2>/dev/null | head -1) 32 | tee $OUT/decode_kernel_stats.txt
