#!/bin/bash
# final sources: the c3 profile set (r07j), c2's counter passes (r07k), the full GPU suite
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
bash scripts/collect_profiles.sh r07j
rm -rf $O/prof_r07k
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r07k -- python3 bench.py --config c2 --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.3 > $O/r07k_bench_c2_under_rocprof.json 2> $O/r07k_rocprof.err
cp $(find $O/prof_r07k -name "*kernel_stats.csv" | head -1) $O/r07k_c2_kernel_stats.csv
timeout -k 10 900 bash scripts/pmc.sh r07k c2 3 > $O/r07k_pmc.log 2>&1 || { tail -5 $O/r07k_pmc.log; exit 1; }
python3 scripts/pmc_to_json.py $O/pmc_r07k $O/r07k_c2_pmc.json c2
python3 scripts/hbm_table.py $O/r07k_c2_pmc.json $O/r07k_c2_kernel_stats.csv r07k_c2 > $O/r07k_c2_hbm_by_kernel.md
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/r07j_pytest.txt 2>&1 || { tail -30 $O/r07j_pytest.txt; exit 1; }
tail -3 $O/r07j_pytest.txt
