# r08z: where the 2 % went: eager per-kernel times of c3, same box: r07j sources, the tree, the tree with the long-list variants off
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_nan.py -q -m gpu -x --timeout 300 -k "long or lists" > $O/r08z_pytest.txt 2>&1 || { tail -5 $O/r08z_pytest.txt; exit 1; }
F="--no-cpu-baseline --no-batched-step --sustained-steps 0 --min-seconds 2"
for rep in 1 2; do
  (cd _prev_r07j && timeout -k 10 200 python bench.py --config c3 $F) > $O/r08z_r07j_$rep.json 2>> $O/r08z.err || exit 1
  timeout -k 10 200 python bench.py --config c3 $F --full-run-steps 0 > $O/r08z_tree_$rep.json 2>> $O/r08z.err || exit 1
  WDGS_LONG_LISTS=0 timeout -k 10 200 python bench.py --config c3 $F --full-run-steps 0 > $O/r08z_nolong_$rep.json 2>> $O/r08z.err || exit 1
done
python - <<'PY' > gpurun_out/r08z_kernels_same_box.txt
import json
for who in ("r07j", "tree", "nolong"):
    for rep in (1, 2):
        j = json.loads(open(f"gpurun_out/r08z_{who}_{rep}.json").read().strip().splitlines()[-1])
        print(f"{who:7s} {j['value']:8.1f} it/s  " + "  ".join(f"{k} {v * 1e3:.1f}" for k, v in sorted(j["kernel_ms_per_view"].items())))
PY
cat gpurun_out/r08z_kernels_same_box.txt
