# r09k: c2 on one box: round 4's sources, the tree, the tree with the long-list variants off
cd $GRAFT_REPO_ROOT
O=gpurun_out
F="--config c2 --no-cpu-baseline --no-batched-step --sustained-steps 0 --min-seconds 3"
for rep in 1 2; do
  (cd _prev_r07j && timeout -k 10 200 python bench.py $F) > $O/r09k_r07j_$rep.json 2>> $O/r09k.err || exit 1
  timeout -k 10 200 python bench.py $F --full-run-steps 0 > $O/r09k_tree_$rep.json 2>> $O/r09k.err || exit 1
  WDGS_LONG_LISTS=0 timeout -k 10 200 python bench.py $F --full-run-steps 0 > $O/r09k_nolong_$rep.json 2>> $O/r09k.err || exit 1
done
python - <<'PY' > gpurun_out/r09k_c2_same_box.txt
import json
print("c2, same box, alternating runs; kernels: eager per-kernel pass of the same run, us per view")
for who in ("r07j", "tree", "nolong"):
    for rep in (1, 2):
        j = json.loads(open(f"gpurun_out/r09k_{who}_{rep}.json").read().strip().splitlines()[-1])
        print(f"{who:7s} {j['value']:8.1f} it/s ({j['ms_per_step']:.4f} ms)  " + "  ".join(f"{k} {v * 1e3:.1f}" for k, v in sorted(j["kernel_ms_per_view"].items())))
PY
cat gpurun_out/r09k_c2_same_box.txt
