cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > gpurun_out/r02i_gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02i_gputest.log; tail -4 gpurun_out/r02i_gputest.log
python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --sustained-steps 0 > gpurun_out/r02i_bench_c3.json 2> gpurun_out/r02i_bench_c3.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02i_bench_c3.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["kernel_ms_per_view"])
PY
python scripts/profile_step.py c3 10 2>&1 | grep "sort_\|tile_ranges\|kernel sum"
