#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
run() {  # name, config, env...
  local name=$1; local cfg=$2; shift; shift
  env "$@" timeout -k 10 300 python3 bench.py --config $cfg --sustained-steps 0 --no-cpu-baseline --no-batched-step > $O/r06i_$name.json 2> $O/r06i_bench.err
  python3 -c "
import json;d=json.load(open('$O/r06i_$name.json'));print('$name',d['value'],d['ms_per_step'],d['kernel_ms_per_view'].get('rasterize'))"
}
for rep in 1 2; do
run c2_wpw4_$rep c2 A=1
run c2_wpw1_$rep c2 WDGS_RASTER_WPW=1
run c3_wpw4_$rep c3 A=1
run c3_wpw1_$rep c3 WDGS_RASTER_WPW=1
done
