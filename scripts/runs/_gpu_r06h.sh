#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_js_host.py tests/test_gpu_napi.py -x -q -m gpu -k "typescript_side_host" > $O/r06h_pytest.txt 2>&1 || { tail -40 $O/r06h_pytest.txt; exit 1; }
tail -4 $O/r06h_pytest.txt
