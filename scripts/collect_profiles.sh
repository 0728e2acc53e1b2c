#!/bin/bash
# Everything the round's profiles/ set is made of, on the GPU box:   bash scripts/collect_profiles.sh <tag>
#   bench lines (c3 full with sustained + cpu_baseline legs, c2, c5, c3 with 8 views per step on 3 lanes and on 1),
#   rocprofv3 --kernel-trace --stats of the c3 bench (kernel_stats.csv), the PMC passes (scripts/pmc.sh) and the per-kernel HBM / VALU table.
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
# (counters first: bench.py takes its VALU counts and HBM traffic from the newest profiles/*_pmc.json whose recorded source hashes are the working tree's,
# so the set collected HERE is put where the bench lines that follow find it -- VERDICT r4 "weak" 12: the committed bench line carried a null frac)
echo "== kernel trace of the c3 bench"
rm -rf $O/prof_${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG} -- python3 bench.py --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.3 > $O/${TAG}_bench_c3_under_rocprof.json 2> $O/${TAG}_rocprof.err
cp $(find $O/prof_${TAG} -name "*kernel_stats.csv" | head -1) $O/${TAG}_c3_kernel_stats.csv
echo "== PMC passes"
timeout -k 10 900 bash scripts/pmc.sh ${TAG} c3 3 > $O/${TAG}_pmc.log 2>&1 || { tail -5 $O/${TAG}_pmc.log; exit 1; }
python3 scripts/pmc_to_json.py $O/pmc_${TAG} $O/${TAG}_pmc.json c3
cp $O/${TAG}_pmc.json $R/profiles/${TAG}_pmc.json
python3 scripts/hbm_table.py $O/${TAG}_pmc.json $O/${TAG}_c3_kernel_stats.csv ${TAG} > $O/${TAG}_hbm_by_kernel.md
cat $O/${TAG}_hbm_by_kernel.md
echo "== bench c3 (headline, full: the driver's command)"; T0=$(date +%s.%N); timeout -k 10 600 python3 bench.py > $O/${TAG}_bench_c3.json 2> $O/${TAG}_bench_c3.err; echo "default bench.py wall: $(python3 -c "import time,sys; print(round(time.time() - float(sys.argv[1]), 1))" $T0) s" | tee $O/${TAG}_bench_c3_wall.txt
echo "== bench c2"; timeout -k 10 300 python3 bench.py --config c2 > $O/${TAG}_bench_c2.json 2> $O/${TAG}_bench_c2.err
echo "== bench c5"; timeout -k 10 400 python3 bench.py --config c5 --steps 10 --warmup 2 --sustained-steps 0 --no-cpu-baseline > $O/${TAG}_bench_c5.json 2> $O/${TAG}_bench_c5.err
echo "== bench c3, 8 views per step"; for L in 3 1; do timeout -k 10 500 python3 bench.py --views-per-rank 8 --lanes $L --steps 10 --warmup 2 --sustained-steps 0 --no-cpu-baseline > $O/${TAG}_bench_c3_vpr8_lanes${L}.json 2> $O/${TAG}_bench_vpr8.err; done
echo "== the same measurement through the TypeScript-side host (node + N-API addon)"
timeout -k 10 400 node bindings/napi/bench.js --config c3 --sustained-steps 608 > $O/${TAG}_benchjs_c3.json 2> $O/${TAG}_benchjs_c3.err
timeout -k 10 500 node bindings/napi/bench.js --config c3 --views-per-step 8 --steps 10 --warmup 2 > $O/${TAG}_benchjs_c3_vpr8.json 2> $O/${TAG}_benchjs_c3_vpr8.err
echo "== c2 kernel trace + counters"
rm -rf $O/prof_${TAG}c2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}c2 -- python3 bench.py --config c2 --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.3 > $O/${TAG}_bench_c2_under_rocprof.json 2>> $O/${TAG}_rocprof.err || true
cp $(find $O/prof_${TAG}c2 -name "*kernel_stats.csv" | head -1) $O/${TAG}_c2_kernel_stats.csv || true
timeout -k 10 600 bash scripts/pmc.sh ${TAG}c2 c2 3 > $O/${TAG}_c2_pmc.log 2>&1 || tail -5 $O/${TAG}_c2_pmc.log
python3 scripts/pmc_to_json.py $O/pmc_${TAG}c2 $O/${TAG}_c2_pmc.json c2 || true
python3 scripts/hbm_table.py $O/${TAG}_c2_pmc.json $O/${TAG}_c2_kernel_stats.csv ${TAG}c2 > $O/${TAG}_c2_hbm_by_kernel.md || true
python3 -c "
import json
for f in ('bench_c3','bench_c2','bench_c5','bench_c3_vpr8_lanes3','bench_c3_vpr8_lanes1'):
    d=json.load(open('$O/${TAG}_'+f+'.json')); print(f, d['value'], d['ms_per_step'], d.get('c3_as_written_iters_per_s'), d['roofline']['kernel'], d['roofline'].get('frac'), d['roofline'].get('hbm_frac'), d.get('batched_step'))
for f in ('benchjs_c3','benchjs_c3_vpr8'):
    d=json.load(open('$O/${TAG}_'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['ms_per_step_awaiting_every_step'])
"
