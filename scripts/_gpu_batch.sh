cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02q
python bench.py --gpus 1 --steps 30 --warmup 5 > gpurun_out/r02q/bench_c3.json 2> gpurun_out/r02q/bench_c3.err; echo "c3 rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 --views-per-rank 8 --no-cpu-baseline --sustained-steps 0 > gpurun_out/r02q/bench_c3_vpr8.json 2> gpurun_out/r02q/bench_c3_vpr8.err; echo "vpr8 rc=$?"
python bench.py --gpus 1 --config c2 --steps 100 --warmup 10 > gpurun_out/r02q/bench_c2.json 2> gpurun_out/r02q/bench_c2.err; echo "c2 rc=$?"
python bench.py --gpus 1 --config c5 --steps 10 --warmup 3 --no-cpu-baseline --sustained-steps 0 > gpurun_out/r02q/bench_c5.json 2> gpurun_out/r02q/bench_c5.err; echo "c5 rc=$?"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02q/prof -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --sustained-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/r02q/bench_c3_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02q/bench_rocprof.err); echo "rocprof rc=$?"
bash scripts/pmc.sh r02c c3 3 > gpurun_out/r02q/pmc.log 2>&1; echo "pmc rc=$?"
python scripts/profile_step.py c5 5 > gpurun_out/r02q/c5_kernel_times.txt 2>&1
