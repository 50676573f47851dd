#!/bin/bash
# rocprofv3 kernel trace of the schema encode (tools/encode_profile.py); per-kernel summary of the last N ms.
# usage: tools/prof_encode.sh <outdir-under-gpurun_out> [tail_ms]
out=gpurun_out/$1; tail_ms=${2:-250}
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out -o enc -- python tools/encode_profile.py > $out.log 2>&1
grep share_trunk $out.log
python tools/rocpd_stats.py $out/enc_results.db --tail-ms $tail_ms | cut -c1-200 | head -${TOPN:-22}
rm -f $out/enc_results.db
