#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_js_host.py tests/test_gpu_napi.py tests/test_gpu_dp.py tests/test_gpu_viewer.py tests/test_gpu_deferred_sh.py -x -q -m gpu > $O/r07c_pytest.txt 2>&1 || { tail -50 $O/r07c_pytest.txt; exit 1; }
tail -3 $O/r07c_pytest.txt
