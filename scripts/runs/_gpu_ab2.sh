cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03l}; VAR=${2:-WDGS_BWR_WPW}
env $VAR=${4:-1} timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_viewer.py -x -q -m gpu > gpurun_out/${TAG}_test.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/${TAG}_test.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do
for v in ${3:-0} ${4:-1}; do
  env $VAR=$v timeout -k 10 300 python bench.py --sustained-steps 0 --no-cpu-baseline > gpurun_out/${TAG}_ab_${v}_${rep}.json 2> gpurun_out/${TAG}_ab.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_ab.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/${TAG}_ab_${v}_${rep}.json'));k=d['kernel_ms_per_view'];print('$VAR=$v rep=$rep', d['value'], d['ms_per_step'], d['timed_blocks']['ms_per_step_min'], {a:k[a] for a in ('backward_rasterize','rasterize','loss_grad') if a in k})"
done
done
