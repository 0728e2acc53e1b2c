#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > $O/r06p_pytest.txt 2>&1 || { tail -40 $O/r06p_pytest.txt; exit 1; }
tail -3 $O/r06p_pytest.txt
timeout -k 10 600 python3 -W always scripts/train_demo.py c3 10000 8 1.0 1000 > $O/r06o_train_demo_c3_10000.txt 2>&1 || { tail -20 $O/r06o_train_demo_c3_10000.txt; exit 1; }
tail -14 $O/r06o_train_demo_c3_10000.txt
