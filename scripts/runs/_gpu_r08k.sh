# r08k: collective-safe densify event, allocation cache, full_run bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_lifecycle.py -q -m gpu -x --timeout 600 > $O/r08k_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/r08k_pytest.txt
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/r08k_bench_c3.json 2> $O/r08k_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r08k_bench_c3.json'))
print(d['value'], d['ms_per_step'], d['c3_as_written_iters_per_s'], d['c3_full_run_iters_per_s'])
print(json.dumps(d['full_run'], indent=None)[:1500])
PY
