#!/bin/bash
# rasterize without its exec-mask region (inactive pixels composite with weight 0): parity under the variant build, then same-box A/B
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
WDGS_LIB_PATH=$PWD/webdgs_amd/lib/libwebdgs_hip_bl.so timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_viewer.py -x -q -m gpu > $O/r06e_pytest.txt 2>&1 || { tail -30 $O/r06e_pytest.txt; exit 1; }
tail -2 $O/r06e_pytest.txt
for rep in 1 2; do for v in tree branchless; do for C in c3 c2; do
  L=$PWD/webdgs_amd/lib/libwebdgs_hip.so; [ $v = branchless ] && L=$PWD/webdgs_amd/lib/libwebdgs_hip_bl.so
  WDGS_LIB_PATH=$L timeout -k 10 300 python3 bench.py --config $C --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.5 > $O/r06e_${C}_${v}_${rep}.json 2> $O/r06e.err
  python3 -c "
import json;d=json.load(open('$O/r06e_${C}_${v}_${rep}.json'));k=d['kernel_ms_per_view'];print('$C $v rep=$rep', d['value'], d['ms_per_step'], k.get('rasterize'))"
done; done; done
