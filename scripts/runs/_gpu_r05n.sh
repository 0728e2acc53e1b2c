# view-batched K1 at 4 (122 VGPRs), 5 (96 + 56 B scratch) and 6 (80 + 120 B scratch) waves per SIMD: the c3 batched step with per-kernel events
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05n}
for rep in 1 2; do
for v in "" _k1v5 _k1v6; do
  WDGS_LIB_PATH=$GRAFT_REPO_ROOT/webdgs_amd/lib/libwebdgs_hip${v}.so timeout -k 10 300 python bench.py --views-per-rank 8 --lanes 3 --steps 10 --warmup 2 --sustained-steps 0 --no-cpu-baseline --min-seconds 1.0 > gpurun_out/${TAG}_vpr8${v}_${rep}.json 2> gpurun_out/${TAG}.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/${TAG}_vpr8${v}_${rep}.json'));print('lib${v} rep=$rep', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done
done
