#!/bin/bash
# loss_grad with 1 / 2 / 4 pixels per thread: parity under each, then same-box A/B at c2 and c3
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
for P in 1 2; do
WDGS_LOSS_PPT=$P timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q -m gpu > $O/r06j_pytest_$P.txt 2>&1 || { tail -30 $O/r06j_pytest_$P.txt; exit 1; }
tail -1 $O/r06j_pytest_$P.txt
done
run() {  # name, config, env...
  local name=$1; local cfg=$2; shift; shift
  env "$@" timeout -k 10 300 python3 bench.py --config $cfg --sustained-steps 0 --no-cpu-baseline --no-batched-step > $O/r06j_$name.json 2> $O/r06j_bench.err
  python3 -c "
import json;d=json.load(open('$O/r06j_$name.json'));print('$name',d['value'],d['ms_per_step'],d['kernel_ms_per_view'].get('loss_grad'))"
}
for rep in 1 2; do
for P in 4 2 1; do run c2_ppt${P}_$rep c2 WDGS_LOSS_PPT=$P; done
for P in 4 2 1; do run c3_ppt${P}_$rep c3 WDGS_LOSS_PPT=$P; done
done
