# same-box A/B of one environment toggle:  bash scripts/_gpu_ab_env.sh <tag> <ENV_NAME> <value A> <value B> [kernel names...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; VAR=$2; A=$3; B=$4; shift 4
KERNELS="${@:-backward_rasterize rasterize}"
for rep in 1 2; do
for v in "$A" "$B"; do
  env $VAR=$v timeout -k 10 300 python bench.py --sustained-steps 0 --no-cpu-baseline > gpurun_out/${TAG}_ab_${v}_${rep}.json 2> gpurun_out/${TAG}_ab.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_ab.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/${TAG}_ab_${v}_${rep}.json'));k=d['kernel_ms_per_view'];print('$VAR=$v rep=$rep', d['value'], d['ms_per_step'], d['timed_blocks']['ms_per_step_min'], {a:k[a] for a in '$KERNELS'.split() if a in k})"
done
done
