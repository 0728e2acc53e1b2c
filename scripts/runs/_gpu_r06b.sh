#!/bin/bash
# segment_sort with 8- / 9-bit one-pass routes: sort parity tests, then same-box A/B at c3 and c2 (prev = HEAD's build)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_ops.py -x -q -m gpu > $O/r06b_pytest.txt 2>&1 || { tail -30 $O/r06b_pytest.txt; exit 1; }
tail -2 $O/r06b_pytest.txt
for rep in 1 2; do for v in prev new; do for C in c3 c2; do
  L=$PWD/webdgs_amd/lib/libwebdgs_hip.so; [ $v = prev ] && L=$PWD/webdgs_amd/lib/libwebdgs_hip_prev.so
  WDGS_LIB_PATH=$L timeout -k 10 300 python3 bench.py --config $C --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.5 > $O/r06b_${C}_${v}_${rep}.json 2> $O/r06b.err
  python3 -c "
import json;d=json.load(open('$O/r06b_${C}_${v}_${rep}.json'));k=d['kernel_ms_per_view'];print('$C $v rep=$rep', d['value'], d['ms_per_step'], k.get('sort'))"
done; done; done
