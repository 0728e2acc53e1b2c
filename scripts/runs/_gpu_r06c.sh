#!/bin/bash
# closing check: full GPU suite, smoke(), the default bench line (as the driver runs it) and c2
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/r06c_pytest.txt 2>&1 || { tail -40 $O/r06c_pytest.txt; exit 1; }
tail -3 $O/r06c_pytest.txt
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
S=$(date +%s.%N); timeout -k 10 600 python3 bench.py > $O/r06c_bench_c3.json 2> $O/r06c_bench_c3.err; E=$(date +%s.%N); echo "default bench wall: $(echo "$E - $S" | bc) s"
timeout -k 10 300 python3 bench.py --config c2 > $O/r06c_bench_c2.json 2> $O/r06c_bench_c2.err
python3 -c "
import json
for c in ('c3','c2'):
    d=json.load(open('$O/r06c_bench_'+c+'.json')); print(c, d['value'], d['ms_per_step'], d.get('c3_as_written_iters_per_s'), d['roofline'].get('frac'), d['roofline'].get('achieved'), d['roofline'].get('traffic'), (d.get('batched_step') or {}).get('views_per_s'), d['cpu_baseline'])
"
