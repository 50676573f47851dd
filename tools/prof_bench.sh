#!/bin/bash
# rocprofv3 kernel trace of the bench's timed steps; prints the per-kernel summary of the last N ms.
# usage: tools/prof_bench.sh <outdir-under-gpurun_out> [tail_ms] [extra bench args]
out=gpurun_out/$1; tail_ms=${2:-30}; shift; shift
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out -o bench -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-context --no-library --no-int8 "$@" > $out.log 2>&1
grep metric $out.log | cut -c1-160
python tools/rocpd_stats.py $out/bench_results.db --tail-ms $tail_ms | cut -c1-200 | head -${TOPN:-16}
rm -f $out/bench_results.db   # keep gpurun_out small; the summary is what gets committed
