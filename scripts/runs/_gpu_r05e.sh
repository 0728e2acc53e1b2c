# c2: backward_rasterize with fewer resident waves per CU (dynamic dispatch balances unequal tiles); c3 batched step with view groups
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05e}
for rep in 1 2; do
for pad in 0 1700 3000 5000 8300 11000; do
  WDGS_BWR_PAD_LDS=$pad timeout -k 10 300 python bench.py --config c2 --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.5 > gpurun_out/${TAG}_c2_pad${pad}_${rep}.json 2> gpurun_out/${TAG}_c2.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_c2.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/${TAG}_c2_pad${pad}_${rep}.json'));k=d['kernel_ms_per_view'];print('c2 pad=$pad rep=$rep', d['value'], d['ms_per_step'], {a:k[a] for a in ('backward_rasterize','rasterize') if a in k})"
done
done
for g in 0 4 2; do
  WDGS_VIEW_GROUP=$g timeout -k 10 300 python bench.py --views-per-rank 8 --lanes 3 --steps 10 --warmup 2 --sustained-steps 0 --no-cpu-baseline --no-profile --min-seconds 1.5 > gpurun_out/${TAG}_vpr8_group${g}.json 2> gpurun_out/${TAG}_vpr8.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_vpr8.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/${TAG}_vpr8_group${g}.json'));print('vpr8 view_group=$g', d['value'], d['ms_per_step'], d['timed_blocks']['ms_per_step_min'], d['timed_blocks']['ms_per_step_max'])"
done
