cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03f}
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_test.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/${TAG}_test.log
[ $rc -eq 0 ] || exit 1
for v in 0 1; do
  WDGS_DEFERRED_SH=$v timeout -k 10 300 python bench.py --sustained-steps 0 --no-cpu-baseline > gpurun_out/${TAG}_ab_dsh${v}.json 2> gpurun_out/${TAG}_ab.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_ab.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/${TAG}_ab_dsh${v}.json'));k=d['kernel_ms_per_view'];print('deferred_sh=$v', d['value'], d['ms_per_step'], d['timed_blocks']['ms_per_step_min'], {a:k[a] for a in ('geometry_backward_adam','project_count','sort','emit') if a in k})"
done
