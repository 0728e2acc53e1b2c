# c2: backward_rasterize as persistent waves pulling 8x8 blocks from per-XCD work queues (WDGS_BWR_QUEUE_WAVES = number of waves; 0 = hardware dispatch)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05h}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py tests/test_gpu_edges.py -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${TAG}_tests.log
for rep in 1 2; do
for q in 0 4800 4096 3584 3072 2048; do
  WDGS_BWR_QUEUE_WAVES=$q timeout -k 10 300 python bench.py --config c2 --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.5 > gpurun_out/${TAG}_c2_q${q}_${rep}.json 2> gpurun_out/${TAG}_c2.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_c2.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/${TAG}_c2_q${q}_${rep}.json'));k=d['kernel_ms_per_view'];print('c2 queue_waves=$q rep=$rep', d['value'], d['ms_per_step'], {a:k[a] for a in ('backward_rasterize','rasterize') if a in k})"
done
done
