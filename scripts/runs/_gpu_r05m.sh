cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r05k
O=$GRAFT_REPO_ROOT/gpurun_out
S=$(date +%s.%N)
python3 bench.py > $O/${TAG}_bench_c3.json 2> $O/${TAG}_bench_c3.err || { tail -5 $O/${TAG}_bench_c3.err; exit 1; }
E=$(date +%s.%N)
echo "default bench.py wall seconds: $(python3 -c "print(round($E - $S, 1))")"
python3 -c "
import json
d=json.load(open('$O/${TAG}_bench_c3.json')); print(d['value'], d['ms_per_step'], d['ms_per_step_awaiting_every_step'], d['c3_as_written_iters_per_s']); print(d['roofline']); print(d['batched_step']); print(d['cpu_baseline']); print(d['sustained']); print(d['kernel_ms_per_view'], d['kernel_ms_per_step'])"
