# backward_rasterize with two splats in flight per wave (PAIRS) on grids that fit the machine at once: parity, c2 A/B, and c3 unchanged (prev = HEAD's build)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05q}
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${TAG}_tests.log
for rep in 1 2; do
for pz in 0 1; do
  WDGS_BWR_PAIRS=$pz timeout -k 10 300 python bench.py --config c2 --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.5 > gpurun_out/${TAG}_c2_pairs${pz}_${rep}.json 2> gpurun_out/${TAG}_c2.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_c2.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/${TAG}_c2_pairs${pz}_${rep}.json'));k=d['kernel_ms_per_view'];print('c2 pairs=$pz rep=$rep', d['value'], d['ms_per_step'], {a:k[a] for a in ('backward_rasterize','rasterize') if a in k})"
done
for v in prev new; do
  L=$GRAFT_REPO_ROOT/webdgs_amd/lib/libwebdgs_hip.so; [ $v = prev ] && L=$GRAFT_REPO_ROOT/webdgs_amd/lib/libwebdgs_hip_prev.so
  WDGS_LIB_PATH=$L timeout -k 10 300 python bench.py --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.5 > gpurun_out/${TAG}_c3_${v}_${rep}.json 2> gpurun_out/${TAG}_c3.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_c3.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/${TAG}_c3_${v}_${rep}.json'));k=d['kernel_ms_per_view'];print('c3 $v rep=$rep', d['value'], d['ms_per_step'], {a:k[a] for a in ('backward_rasterize','rasterize') if a in k})"
done
done
WDGS_BWR_PAIRS=1 timeout -k 10 300 python bench.py --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.5 > gpurun_out/${TAG}_c3_pairs1.json 2> gpurun_out/${TAG}_c3.err && python -c "
import json;d=json.load(open('gpurun_out/${TAG}_c3_pairs1.json'));k=d['kernel_ms_per_view'];print('c3 pairs forced on', d['value'], d['ms_per_step'], {a:k[a] for a in ('backward_rasterize','rasterize') if a in k})"
