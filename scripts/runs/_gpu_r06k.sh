#!/bin/bash
# row-pass partitions of 1024 / 2048 keys instead of 4096 (variant builds): sort tests under the smallest, then same-box A/B at c2 and c3
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
WDGS_LIB_PATH=$PWD/webdgs_amd/lib/libwebdgs_hip_si4.so timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_ops.py -x -q -m gpu > $O/r06k_pytest.txt 2>&1 || { tail -30 $O/r06k_pytest.txt; exit 1; }
tail -1 $O/r06k_pytest.txt
for rep in 1 2; do for v in 16 8 4; do for C in c2 c3; do
  L=$PWD/webdgs_amd/lib/libwebdgs_hip.so; [ $v != 16 ] && L=$PWD/webdgs_amd/lib/libwebdgs_hip_si$v.so
  WDGS_LIB_PATH=$L timeout -k 10 300 python3 bench.py --config $C --sustained-steps 0 --no-cpu-baseline --no-batched-step > $O/r06k_${C}_items${v}_${rep}.json 2> $O/r06k.err
  python3 -c "
import json;d=json.load(open('$O/r06k_${C}_items${v}_${rep}.json'));k=d['kernel_ms_per_view'];print('$C items=$v rep=$rep', d['value'], d['ms_per_step'], k.get('sort'))"
done; done; done
