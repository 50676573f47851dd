# full -m gpu suite on the box: bash tools/run_full_gpu.sh <outdir-name>
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/$1/pytest.log 2>&1; echo rc=$?
tail -5 gpurun_out/$1/pytest.log
