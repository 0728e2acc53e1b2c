#!/bin/bash
# backward_rasterize with the next iteration's geometry / conic records read one iteration ahead: parity, then c2 and c3 A/B
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
WDGS_BWR_ROLES=1 WDGS_BWR_PREFETCH=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer_oracle.py tests/test_gpu_edges.py -x -q -m gpu > $O/r05u_pytest.txt 2>&1 || { tail -30 $O/r05u_pytest.txt; exit 1; }
tail -2 $O/r05u_pytest.txt
run() {  # name, config, env...
  local name=$1; local cfg=$2; shift; shift
  env "$@" WDGS_BWR_ROLES=1 timeout -k 10 300 python3 bench.py --config $cfg --sustained-steps 0 --no-cpu-baseline --no-batched-step > $O/r05u_$name.json 2> $O/r05u_bench.err
  python3 -c "
import json;d=json.load(open('$O/r05u_$name.json'));print('$name',d['value'],d['ms_per_step'])"
}
for rep in 1 2; do
run c2_base_$rep c2 A=1
run c2_prefetch_$rep c2 WDGS_BWR_PREFETCH=1
run c3_base_$rep c3 A=1
run c3_prefetch_$rep c3 WDGS_BWR_PREFETCH=1
done
