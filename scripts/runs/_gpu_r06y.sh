#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_dp.py -x -q -m gpu -k "overflows" > $O/r06y_pytest.txt 2>&1 || { tail -60 $O/r06y_pytest.txt; exit 1; }
tail -3 $O/r06y_pytest.txt
