#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_js_host.py tests/test_gpu_viewer.py tests/test_gpu_pipeline.py -x -q -m gpu > $O/r06x_pytest.txt 2>&1 || { tail -60 $O/r06x_pytest.txt; exit 1; }
tail -3 $O/r06x_pytest.txt
