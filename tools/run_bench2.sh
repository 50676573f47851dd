cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
PC_BENCH_SAME_DEVICE=1 PC_BENCH_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --no-context > gpurun_out/r3j/bench2.json 2> gpurun_out/r3j/bench2.err; echo rc=$?
tail -3 gpurun_out/r3j/bench2.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/r3j/bench2.json").read().strip().split("\n")[-1])
print(d["n_gpus"], d["ms_per_step"], d["value"], d["encode"]["per_rank_computed_tokens"], d["encode"]["library_identical_on_all_ranks"], d["encode"]["tokens_per_s"], d["encode_library"]["per_rank_computed_tokens"], d["encode_library"]["tokens_per_s"])
PY
