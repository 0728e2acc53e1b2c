#!/bin/bash
# node: c3 as written (608 steps) and the reference's default 10 000 iterations through the JS Trainer; Python: c2 for 10 000
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
timeout -k 10 400 node bindings/napi/bench.js --config c3 --sustained-steps 608 > $O/r06r_benchjs_c3.json 2> $O/r06r_benchjs_c3.err || { tail -20 $O/r06r_benchjs_c3.err; exit 1; }
python3 -c "
import json;d=json.load(open('$O/r06r_benchjs_c3.json'));print(d['value'],d['ms_per_step'],d['c3_as_written_iters_per_s'],d['sustained'])"
timeout -k 10 600 node bindings/napi/bench.js --config c3 --sustained-steps 10000 --no-profile --min-seconds 0.2 > $O/r06r_benchjs_c3_10000.json 2> $O/r06r_benchjs_c3_10000.err || { tail -20 $O/r06r_benchjs_c3_10000.err; exit 1; }
cat $O/r06r_benchjs_c3_10000.err | tail -3
python3 -c "
import json;d=json.load(open('$O/r06r_benchjs_c3_10000.json'));s=d['sustained'];print(s['steps'],s['iters_per_s_overall'],s['densify_events'],s['points'][:4],s['points'][-3:])"
timeout -k 10 600 python3 -W always scripts/train_demo.py c2 10000 8 1.0 1000 > $O/r06r_train_demo_c2_10000.txt 2>&1 || { tail -20 $O/r06r_train_demo_c2_10000.txt; exit 1; }
tail -12 $O/r06r_train_demo_c2_10000.txt
