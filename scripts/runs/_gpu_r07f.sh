#!/bin/bash
# c2: counter passes and the per-kernel table (the c3 set is r06n), then the c2 bench line with its roofline fraction from them
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
rm -rf $O/prof_r07f
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r07f -- python3 bench.py --config c2 --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.3 > $O/r07f_bench_c2_under_rocprof.json 2> $O/r07f_rocprof.err
cp $(find $O/prof_r07f -name "*kernel_stats.csv" | head -1) $O/r07f_c2_kernel_stats.csv
timeout -k 10 900 bash scripts/pmc.sh r07f c2 3 > $O/r07f_pmc.log 2>&1 || { tail -5 $O/r07f_pmc.log; exit 1; }
python3 scripts/pmc_to_json.py $O/pmc_r07f $O/r07f_c2_pmc.json c2
python3 scripts/hbm_table.py $O/r07f_c2_pmc.json $O/r07f_c2_kernel_stats.csv r07f_c2 > $O/r07f_c2_hbm_by_kernel.md
tail -14 $O/r07f_c2_hbm_by_kernel.md
mkdir -p profiles && cp $O/r07f_c2_pmc.json profiles/r07f_c2_pmc.json
timeout -k 10 300 python3 bench.py --config c2 > $O/r07f_bench_c2.json 2>/dev/null
python3 -c "
import json;d=json.load(open('$O/r07f_bench_c2.json'));print(d['value'],d['ms_per_step'],d['roofline'])"
