#!/bin/bash
# rasterize leaving a chunk's list as soon as the whole wave is saturated (tested every 4 / 8 / 16 iterations): parity under one variant, then same-box A/B
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
WDGS_LIB_PATH=$PWD/webdgs_amd/lib/libwebdgs_hip_ee4.so timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_viewer.py -x -q -m gpu > $O/r06f_pytest.txt 2>&1 || { tail -30 $O/r06f_pytest.txt; exit 1; }
tail -2 $O/r06f_pytest.txt
for rep in 1 2; do for v in tree ee4 ee8 ee16; do
  L=$PWD/webdgs_amd/lib/libwebdgs_hip.so; [ $v != tree ] && L=$PWD/webdgs_amd/lib/libwebdgs_hip_$v.so
  WDGS_LIB_PATH=$L timeout -k 10 300 python3 bench.py --config c3 --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.5 > $O/r06f_c3_${v}_${rep}.json 2> $O/r06f.err
  python3 -c "
import json;d=json.load(open('$O/r06f_c3_${v}_${rep}.json'));k=d['kernel_ms_per_view'];print('c3 $v rep=$rep', d['value'], d['ms_per_step'], k.get('rasterize'))"
done; done
