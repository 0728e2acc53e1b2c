# r08v: same-box comparison of the headline (c3, c2) between the sources of round 4's final set (worktree _prev_r07j, commit bd20e8a) and the tree
cd $GRAFT_REPO_ROOT
O=gpurun_out
F="--no-cpu-baseline --no-batched-step --sustained-steps 0 --min-seconds 3"
for rep in 1 2; do
  for cfg in c3 c2; do
    (cd _prev_r07j && timeout -k 10 200 python bench.py --config $cfg $F) > $O/r08v_${cfg}_r07j_$rep.json 2>> $O/r08v.err || exit 1
    timeout -k 10 200 python bench.py --config $cfg $F --full-run-steps 0 > $O/r08v_${cfg}_tree_$rep.json 2>> $O/r08v.err || exit 1
  done
done
python - <<'PY' > gpurun_out/r08v_same_box_r07j_vs_tree.txt
import json
print("same box, alternating runs: bench.py --config <cfg> --no-cpu-baseline --no-batched-step --sustained-steps 0 --min-seconds 3")
for cfg in ("c3", "c2"):
    for who in ("r07j", "tree"):
        vals = []
        for rep in (1, 2):
            j = json.loads(open(f"gpurun_out/r08v_{cfg}_{who}_{rep}.json").read().strip().splitlines()[-1])
            vals.append((j["value"], j["ms_per_step"]))
        print(f"{cfg} {who:5s} " + "  ".join(f"{v:9.2f} it/s ({ms:.4f} ms)" for v, ms in vals))
PY
cat gpurun_out/r08v_same_box_r07j_vs_tree.txt
