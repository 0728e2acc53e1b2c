#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
run() {  # name, config, env...
  local name=$1; local cfg=$2; shift; shift
  env "$@" timeout -k 10 300 python3 bench.py --config $cfg --sustained-steps 0 --no-cpu-baseline --no-batched-step > $O/r05z_$name.json 2> $O/r05z_bench.err
  python3 -c "
import json;d=json.load(open('$O/r05z_$name.json'));print('$name',d['value'],d['ms_per_step'])"
}
for rep in 1 2; do
run c2_none_$rep c2 WDGS_BWR_PRIO=0 WDGS_FWR_PRIO=0
run c2_chunk_$rep c2 A=1
for F in 8 16 24; do run c2_fine${F}_$rep c2 WDGS_LIB_PATH=$PWD/webdgs_amd/lib/libwebdgs_hip_fine$F.so; done
done
run c3_default c3 A=1
run c1_none c1 WDGS_BWR_PRIO=0 WDGS_FWR_PRIO=0
run c1_default c1 A=1
