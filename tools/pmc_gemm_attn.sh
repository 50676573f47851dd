export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc/g_$c -o g -- python tools/microbench.py --gemm-only > /dev/null 2>&1
  python tools/rocpd_pmc.py gpurun_out/pmc/g_$c/g_results.db gemm_skinny | grep -v columns
  rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc/a_$c -o a -- python tools/microbench.py --attn-cached-only > /dev/null 2>&1
  python tools/rocpd_pmc.py gpurun_out/pmc/a_$c/a_results.db attn_ | grep -v columns
done
rm -rf gpurun_out/pmc
