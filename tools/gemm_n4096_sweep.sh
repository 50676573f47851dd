cd $GRAFT_REPO_ROOT
# needs a build with the sweep instantiations: PC_BUILD_FLAGS=-DPC_DEV_SWEEPS python __graft_entry__.py --force
for M in 1 12; do
  timeout 120 python tools/gemm_n4096_sweep.py $M 1 1
  PC_GEMM_U=16 timeout 120 python tools/gemm_n4096_sweep.py $M 1 1
  PC_GEMM_U=4 timeout 120 python tools/gemm_n4096_sweep.py $M 1 1
  PC_GEMM_T=2 timeout 120 python tools/gemm_n4096_sweep.py $M 1 1
  timeout 120 python tools/gemm_n4096_sweep.py $M 0 1
  timeout 120 python tools/gemm_n4096_sweep.py $M 0 2
  timeout 120 python tools/gemm_n4096_sweep.py $M 0 4
  PC_GEMM_T=2 timeout 120 python tools/gemm_n4096_sweep.py $M 0 2
  PC_GEMM_T=4 timeout 120 python tools/gemm_n4096_sweep.py $M 0 4
  PC_GEMM_T=8 timeout 120 python tools/gemm_n4096_sweep.py $M 0 8
done
