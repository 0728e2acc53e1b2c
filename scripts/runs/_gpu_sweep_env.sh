# one bench per value of an environment variable:  bash scripts/_gpu_sweep_env.sh <tag> <ENV_NAME> <v1> <v2> ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; VAR=$2; shift 2
for v in "$@"; do
  env $VAR=$v timeout -k 10 300 python bench.py --sustained-steps 0 --no-cpu-baseline --min-seconds 0.5 > gpurun_out/${TAG}_${v}.json 2> gpurun_out/${TAG}.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/${TAG}_${v}.json'));k=d['kernel_ms_per_view'];print('$VAR=$v', d['value'], d['ms_per_step'], {a:k[a] for a in ('backward_rasterize','rasterize') if a in k})"
done
