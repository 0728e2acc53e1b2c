#!/bin/bash
# soak: the reference's default training length (10 000 iterations, densify schedule on) at c3 and c2 through the Python host; c2 through node's bench sustained leg is in the bench lines
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python3 scripts/train_demo.py c3 10000 8 1.0 1000 > $O/r06o_train_demo_c3_10000.txt 2>&1 || { tail -20 $O/r06o_train_demo_c3_10000.txt; exit 1; }
tail -12 $O/r06o_train_demo_c3_10000.txt
timeout -k 10 600 python3 scripts/train_demo.py c2 10000 8 1.0 1000 > $O/r06o_train_demo_c2_10000.txt 2>&1 || { tail -20 $O/r06o_train_demo_c2_10000.txt; exit 1; }
tail -12 $O/r06o_train_demo_c2_10000.txt
