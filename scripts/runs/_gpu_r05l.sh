# the kernel trace of the c3 bench again, without the batched-step leg (its co-running kernels stretch the per-kernel averages), the table, and the default bench line timed
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=r05k
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/prof_${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG} -- python3 bench.py --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.3 > $O/${TAG}_bench_c3_under_rocprof.json 2> $O/${TAG}_rocprof.err || { tail -5 $O/${TAG}_rocprof.err; exit 1; }
cp $(find $O/prof_${TAG} -name "*kernel_stats.csv" | head -1) $O/${TAG}_c3_kernel_stats.csv
python3 scripts/hbm_table.py profiles/${TAG}_pmc.json $O/${TAG}_c3_kernel_stats.csv ${TAG} > $O/${TAG}_hbm_by_kernel.md
cat $O/${TAG}_hbm_by_kernel.md
/usr/bin/time -v python3 bench.py > $O/${TAG}_bench_c3.json 2> $O/${TAG}_bench_c3.err; grep -E "Elapsed|Maximum resident" $O/${TAG}_bench_c3.err
python3 -c "
import json
d=json.load(open('$O/${TAG}_bench_c3.json')); print(d['value'], d['ms_per_step'], d['ms_per_step_awaiting_every_step'], d['c3_as_written_iters_per_s']); print(d['roofline']); print(d['batched_step']); print(d['cpu_baseline']); print(d['sustained'])"
