#!/bin/bash
# c2 backward_rasterize ablations (timing only; the ablated builds compute wrong sums): where does an iteration's time go?
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
run() {  # name, env...
  local name=$1; shift
  env "$@" WDGS_BWR_ROLES=1 timeout -k 10 300 python3 bench.py --config c2 --sustained-steps 0 --no-cpu-baseline > $O/r05t_$name.json 2> $O/r05t_bench.err
  python3 -c "
import json;d=json.load(open('$O/r05t_$name.json'));print('$name',d['value'],d['ms_per_step'],d.get('kernel_ms_per_step') or d.get('kernel_ms'))"
}
run base A=1
run no_atomics WDGS_LIB_PATH=$PWD/webdgs_amd/lib/libwebdgs_hip_noat.so
run no_lds_sums WDGS_LIB_PATH=$PWD/webdgs_amd/lib/libwebdgs_hip_nosum.so
run state_only WDGS_BWR_FF_ALL=1
run base2 A=1
