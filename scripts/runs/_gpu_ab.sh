cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03e}
for rep in 1 2; do
for v in 0 1; do
  WDGS_FORWARD_COLUMNS=$v timeout -k 10 300 python bench.py --sustained-steps 0 --no-cpu-baseline > gpurun_out/${TAG}_ab_cols${v}_${rep}.json 2> gpurun_out/${TAG}_ab.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_ab.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/${TAG}_ab_cols${v}_${rep}.json'));k=d['kernel_ms_per_view'];print('columns=$v rep=$rep', d['value'], d['ms_per_step'], d['timed_blocks']['ms_per_step_min'], {a:k[a] for a in ('sort','emit','project_count','scan') if a in k})"
done
done
