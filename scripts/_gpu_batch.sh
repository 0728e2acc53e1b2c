cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > gpurun_out/r02j_gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02j_gputest.log; tail -4 gpurun_out/r02j_gputest.log
for i in 1 2; do python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --sustained-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['kernel_ms_per_view'])"; done
