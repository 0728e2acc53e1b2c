# r10c: oversized segments, packed-prefix ranking: sort + NaN + edge tests, late regime, scenes, c3 per-kernel against the previous library
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_nan.py tests/test_gpu_edges.py tests/test_gpu_ops.py tests/test_gpu_parity.py -q -m gpu -x --timeout 300 > $O/r10c_pytest.txt 2>&1 || { tail -15 $O/r10c_pytest.txt; exit 1; }
tail -2 $O/r10c_pytest.txt
timeout -k 10 400 python scripts/late_regime_profile.py > $O/r10c_late_regime.txt 2>&1 || { tail -5 $O/r10c_late_regime.txt; exit 1; }
tail -14 $O/r10c_late_regime.txt | head -8
timeout -k 10 300 python scripts/long_list_scenes.py > $O/r10c_long_list_scenes.txt 2>&1; tail -6 $O/r10c_long_list_scenes.txt
F="--no-cpu-baseline --no-batched-step --sustained-steps 0 --min-seconds 2 --full-run-steps 0"
for rep in 1 2; do
  WDGS_LIB_PATH=$PWD/webdgs_amd/lib/libwebdgs_hip_prev.so timeout -k 10 200 python bench.py $F > $O/r10c_prev_$rep.json 2>> $O/r10c.err || exit 1
  timeout -k 10 200 python bench.py $F > $O/r10c_new_$rep.json 2>> $O/r10c.err || exit 1
done
python - <<'PY'
import json
for who in ("prev", "new"):
    for rep in (1, 2):
        j = json.loads(open(f"gpurun_out/r10c_{who}_{rep}.json").read().strip().splitlines()[-1])
        print(f"c3 {who:5s} {j['value']:9.2f} it/s ({j['ms_per_step']:.4f} ms)  " + "  ".join(f"{k} {v * 1e3:.1f}" for k, v in sorted(j["kernel_ms_per_view"].items())))
PY
