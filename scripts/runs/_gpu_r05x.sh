#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
run() {  # name, config, env...
  local name=$1; local cfg=$2; shift; shift
  env "$@" timeout -k 10 300 python3 bench.py --config $cfg --sustained-steps 0 --no-cpu-baseline --no-batched-step > $O/r05x_$name.json 2> $O/r05x_bench.err
  python3 -c "
import json;d=json.load(open('$O/r05x_$name.json'));print('$name',d['value'],d['ms_per_step'])"
}
run c2_base c2 WDGS_BWR_PRIO=0
for S in 020100 010100 010000 1100804 1180c06 1201008 10c0603 1080402 1180c04 1100402 1302010; do
run c2_prio_$S c2 WDGS_BWR_PRIO=1 WDGS_BWR_PRIO_STEPS=$S
done
run c2_base2 c2 WDGS_BWR_PRIO=0
run c3_base c3 WDGS_BWR_PRIO=0
run c3_prio_020100 c3 WDGS_BWR_PRIO=1 WDGS_BWR_PRIO_STEPS=020100
run c3_prio_1100804 c3 WDGS_BWR_PRIO=1 WDGS_BWR_PRIO_STEPS=1100804
