cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03h}
timeout -k 10 900 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_dp.py tests/test_gpu_trainer_oracle.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/${TAG}_test.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/${TAG}_test.log
[ $rc -eq 0 ] || { grep -n "Error\|error\|assert" gpurun_out/${TAG}_test.log | head -20; exit 1; }
for v in 0 1; do
  WDGS_BATCH_VIEWS=$v timeout -k 10 500 python bench.py --views-per-rank 8 --lanes 3 --steps 10 --warmup 2 --sustained-steps 0 --no-cpu-baseline > gpurun_out/${TAG}_vpr8_bv${v}.json 2> gpurun_out/${TAG}_vpr8.err || { echo "bench failed"; tail -8 gpurun_out/${TAG}_vpr8.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/${TAG}_vpr8_bv${v}.json'));k=d['kernel_ms_per_view'];print('batch_views=$v', d['value'], d['ms_per_step'], d['timed_blocks']['ms_per_step_min'], {a:round(b,4) for a,b in k.items()})"
done
