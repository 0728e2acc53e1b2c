#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
timeout -k 10 500 python3 scripts/soak.py c3 2000 8 none > $O/r06v_soak_c3_vpr8.txt 2>&1 || { tail -20 $O/r06v_soak_c3_vpr8.txt; exit 1; }
tail -12 $O/r06v_soak_c3_vpr8.txt
timeout -k 10 500 python3 scripts/soak.py c3 1500 4 capi > $O/r06v_soak_c3_vpr4_capi.txt 2>&1 || { tail -20 $O/r06v_soak_c3_vpr4_capi.txt; exit 1; }
tail -9 $O/r06v_soak_c3_vpr4_capi.txt
timeout -k 10 600 node bindings/napi/bench.js --config c3 --sustained-steps 4000 --no-profile --min-seconds 0.2 > $O/r06v_benchjs_c3_4000.json 2> $O/r06v_benchjs_c3_4000.err || { tail -20 $O/r06v_benchjs_c3_4000.err; exit 1; }
tail -2 $O/r06v_benchjs_c3_4000.err; python3 -c "
import json;d=json.load(open('$O/r06v_benchjs_c3_4000.json'));s=d['sustained'];print(s['steps'],s['iters_per_s_overall'],s['densify_events'],s['ms_per_densify_event'],s['points'][-3:])"
