# the 8-views-per-step bench at several lane counts:  bash scripts/_gpu_lanes_sweep.sh <tag> <lanes...>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
for lanes in "$@"; do
timeout -k 10 500 python bench.py --views-per-rank 8 --lanes $lanes --steps 10 --warmup 2 --sustained-steps 0 --no-cpu-baseline > gpurun_out/${TAG}_vpr8_lanes${lanes}.json 2> gpurun_out/${TAG}_vpr8.err || { echo "bench failed"; tail -8 gpurun_out/${TAG}_vpr8.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/${TAG}_vpr8_lanes${lanes}.json'));print('lanes=$lanes', d['value'], d['ms_per_step'], d['timed_blocks']['ms_per_step_min'])"
done
