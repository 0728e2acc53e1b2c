# same-box A/B of two builds of the library:  bash scripts/_gpu_ab_lib.sh <tag> [kernel names...]   (prev = webdgs_amd/lib/libwebdgs_hip_prev.so)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
KERNELS="${@:-backward_rasterize rasterize}"
for rep in 1 2; do
for v in prev new; do
  L=$GRAFT_REPO_ROOT/webdgs_amd/lib/libwebdgs_hip.so; [ $v = prev ] && L=$GRAFT_REPO_ROOT/webdgs_amd/lib/libwebdgs_hip_prev.so
  WDGS_LIB_PATH=$L timeout -k 10 300 python bench.py --sustained-steps 0 --no-cpu-baseline --min-seconds 0.5 > gpurun_out/${TAG}_${v}_${rep}.json 2> gpurun_out/${TAG}.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/${TAG}_${v}_${rep}.json'));k=d['kernel_ms_per_view'];print('$v rep=$rep', d['value'], d['ms_per_step'], {a:k[a] for a in '$KERNELS'.split() if a in k})"
done
done
