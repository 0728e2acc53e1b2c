#!/bin/bash
# issue priorities in backward_rasterize (default on) and rasterize: parity, then A/B at c2, c3, c5
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer_oracle.py tests/test_gpu_edges.py tests/test_gpu_viewer.py -x -q -m gpu > $O/r05y_pytest.txt 2>&1 || { tail -30 $O/r05y_pytest.txt; exit 1; }
tail -2 $O/r05y_pytest.txt
run() {  # name, config, env...
  local name=$1; local cfg=$2; shift; shift
  env "$@" timeout -k 10 300 python3 bench.py --config $cfg --sustained-steps 0 --no-cpu-baseline --no-batched-step > $O/r05y_$name.json 2> $O/r05y_bench.err
  python3 -c "
import json;d=json.load(open('$O/r05y_$name.json'));print('$name',d['value'],d['ms_per_step'])"
}
for rep in 1 2; do
run c2_none_$rep c2 WDGS_BWR_PRIO=0 WDGS_FWR_PRIO=0
run c2_bwr_$rep c2 WDGS_FWR_PRIO=0
run c2_both_$rep c2 A=1
run c3_none_$rep c3 WDGS_BWR_PRIO=0 WDGS_FWR_PRIO=0
run c3_bwr_$rep c3 WDGS_FWR_PRIO=0
run c3_both_$rep c3 A=1
done
