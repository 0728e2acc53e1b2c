# r09b: parity of the long-list paths, then the headline (c3, c2) on one box: round 4's final sources (worktree _prev_r07j, commit bd20e8a) against the tree
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_nan.py tests/test_gpu_parity.py -q -m gpu -x --timeout 300 > $O/r09b_pytest.txt 2>&1 || { tail -5 $O/r09b_pytest.txt; exit 1; }
tail -2 $O/r09b_pytest.txt
F="--no-cpu-baseline --no-batched-step --sustained-steps 0 --min-seconds 3"
for rep in 1 2; do
  for cfg in c3 c2; do
    (cd _prev_r07j && timeout -k 10 200 python bench.py --config $cfg $F) > $O/r09b_${cfg}_r07j_$rep.json 2>> $O/r09b.err || exit 1
    timeout -k 10 200 python bench.py --config $cfg $F --full-run-steps 0 > $O/r09b_${cfg}_tree_$rep.json 2>> $O/r09b.err || exit 1
  done
done
python - <<'PY' > gpurun_out/r09b_same_box_r07j_vs_tree.txt
import json
print("same box, alternating runs: bench.py --config <cfg> --no-cpu-baseline --no-batched-step --sustained-steps 0 --min-seconds 3")
print("r07j = the sources of round 4's final profile set (commit bd20e8a), tree = this round's; kernels: eager per-kernel pass of the same run, us per view")
for cfg in ("c3", "c2"):
    for who in ("r07j", "tree"):
        for rep in (1, 2):
            j = json.loads(open(f"gpurun_out/r09b_{cfg}_{who}_{rep}.json").read().strip().splitlines()[-1])
            print(f"{cfg} {who:5s} {j['value']:9.2f} it/s ({j['ms_per_step']:.4f} ms)  " + "  ".join(f"{k} {v * 1e3:.1f}" for k, v in sorted(j["kernel_ms_per_view"].items())))
PY
cat gpurun_out/r09b_same_box_r07j_vs_tree.txt
