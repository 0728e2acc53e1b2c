cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_dp.py -q -x > gpurun_out/r02_dp.log 2>&1; echo "dp rc=$?"; tail -5 gpurun_out/r02_dp.log
WDGS_BENCH_WATCHDOG=100 WDGS_DIST_BACKEND=gloo WDGS_FORCE_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config c2 --steps 10 --warmup 3 --views-per-rank 4 --views 8 > gpurun_out/r02_bench_dp2_gloo_c2.json 2> gpurun_out/r02_bench_dp2_gloo_c2.err; echo "rc=$?"; tail -c 1500 gpurun_out/r02_bench_dp2_gloo_c2.json; grep -v "^\[W\|amdgpu.ids\|^\*\*\*\|OMP_NUM" gpurun_out/r02_bench_dp2_gloo_c2.err | tail -20
