#!/bin/bash
# backward_rasterize: issue priority from the wave's remaining entries (longest remaining chain first): parity, c2 and c3 A/B
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
WDGS_BWR_PRIO=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer_oracle.py tests/test_gpu_edges.py -x -q -m gpu > $O/r05w_pytest.txt 2>&1 || { tail -30 $O/r05w_pytest.txt; exit 1; }
tail -2 $O/r05w_pytest.txt
run() {  # name, config, env...
  local name=$1; local cfg=$2; shift; shift
  env "$@" timeout -k 10 300 python3 bench.py --config $cfg --sustained-steps 0 --no-cpu-baseline --no-batched-step > $O/r05w_$name.json 2> $O/r05w_bench.err
  python3 -c "
import json;d=json.load(open('$O/r05w_$name.json'));print('$name',d['value'],d['ms_per_step'])"
}
for rep in 1 2; do
run c2_base_$rep c2 WDGS_BWR_PRIO=0
run c2_prio_$rep c2 WDGS_BWR_PRIO=1
run c3_base_$rep c3 WDGS_BWR_PRIO=0
run c3_prio_$rep c3 WDGS_BWR_PRIO=1
done
