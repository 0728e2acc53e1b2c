cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03g}
for lanes in 3 1; do
timeout -k 10 500 python bench.py --views-per-rank 8 --lanes $lanes --steps 10 --warmup 2 --sustained-steps 0 --no-cpu-baseline > gpurun_out/${TAG}_vpr8_lanes${lanes}.json 2> gpurun_out/${TAG}_vpr8.err || { echo "bench failed"; tail -8 gpurun_out/${TAG}_vpr8.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/${TAG}_vpr8_lanes${lanes}.json'));k=d['kernel_ms_per_view'];print('lanes=$lanes', d['value'], d['ms_per_step'], d['timed_blocks'], {a:round(b,4) for a,b in k.items()})"
done
timeout -k 10 300 python scripts/bwr_lane_utilisation.py c3 300 > gpurun_out/${TAG}_bwr_lane_utilisation.txt 2>&1; tail -12 gpurun_out/${TAG}_bwr_lane_utilisation.txt
