# Same-box A/B of the timed step: the round-2 tree (git worktree _r02_tree, commit 722bb40) against this tree, interleaved.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ab; mkdir -p $O
B="bench.py --steps 200 --warmup 20 --no-context --no-cpu-baseline --no-library"
for i in 1 2 3; do
  (cd _r02_tree && timeout 600 python $B > ../$O/r02_$i.json 2> ../$O/r02_$i.err)
  timeout 600 python $B > $O/r03_$i.json 2> $O/r03_$i.err
done
python3 - <<PY
import json
for t in ("r02","r03"):
    v=[json.loads(open(f"$O/{t}_{i}.json").read().strip().split("\n")[-1])["ms_per_step"] for i in (1,2,3)]
    print(t, [round(x,4) for x in v], "median", sorted(v)[1])
PY
