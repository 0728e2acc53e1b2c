#!/bin/bash
# c2: kernel trace of the bench (durations inside the replayed graph) beside the step time
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
rm -rf $O/prof_r06m
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06m -- python3 bench.py --config c2 --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.3 > $O/r06m_bench_c2_under_rocprof.json 2> $O/r06m_rocprof.err
cp $(find $O/prof_r06m -name "*kernel_stats.csv" | head -1) $O/r06m_c2_kernel_stats.csv
python3 - <<'PY'
import csv,json
rows=list(csv.DictReader(open('gpurun_out/r06m_c2_kernel_stats.csv')))
tot=0
for r in rows[:16]:
    name=r['Name'].split('(')[0][-40:]; avg=float(r['AverageNs'])/1e3; calls=int(r['Calls'])
    print(f"{name:42s} calls={calls:6d} avg_us={avg:8.2f}")
d=json.load(open('gpurun_out/r06m_bench_c2_under_rocprof.json')); print('step ms under rocprof', d['ms_per_step'])
PY
