# r09e: the key skim of dead blocks + the pipelined oversized segment sort: NaN tests, late-regime profile, c3 same-box check
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_nan.py tests/test_gpu_edges.py -q -m gpu -x --timeout 300 > $O/r09e_pytest.txt 2>&1 || { tail -15 $O/r09e_pytest.txt; exit 1; }
tail -2 $O/r09e_pytest.txt
timeout -k 10 400 python scripts/late_regime_profile.py > $O/r09e_late_regime.txt 2>&1 || { tail -5 $O/r09e_late_regime.txt; exit 1; }
tail -16 $O/r09e_late_regime.txt
F="--no-cpu-baseline --no-batched-step --sustained-steps 0 --min-seconds 3"
for rep in 1 2; do
  (cd _prev_r07j && timeout -k 10 200 python bench.py --config c3 $F) > $O/r09e_c3_r07j_$rep.json 2>> $O/r09e.err || exit 1
  timeout -k 10 200 python bench.py --config c3 $F --full-run-steps 0 > $O/r09e_c3_tree_$rep.json 2>> $O/r09e.err || exit 1
done
python - <<'PY'
import json
for who in ("r07j", "tree"):
    for rep in (1, 2):
        j = json.loads(open(f"gpurun_out/r09e_c3_{who}_{rep}.json").read().strip().splitlines()[-1])
        print(f"c3 {who:5s} {j['value']:9.2f} it/s ({j['ms_per_step']:.4f} ms)  " + "  ".join(f"{k} {v * 1e3:.1f}" for k, v in sorted(j["kernel_ms_per_view"].items())))
PY
