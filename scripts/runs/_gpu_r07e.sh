#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests/test_gpu_napi.py tests/test_gpu_js_host.py tests/test_gpu_lifecycle.py tests/test_gpu_fullsize.py -x -q -m gpu -k "node or js or typescript or napi or lifecycle or come_and_go or hosts_alike or bench_js" > $O/r07e_pytest.txt 2>&1 || { tail -40 $O/r07e_pytest.txt; exit 1; }
tail -3 $O/r07e_pytest.txt
