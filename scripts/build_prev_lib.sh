#!/bin/bash
# Builds the library from a committed state of webdgs_amd/csrc (default HEAD) as webdgs_amd/lib/libwebdgs_hip_prev.so, for same-box A/B runs
# against the working tree's build (scripts/runs/_gpu_ab_lib.sh; WDGS_LIB_PATH).   bash scripts/build_prev_lib.sh [commit]
set -e
REV=${1:-HEAD}
R=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
mkdir -p $T/webdgs_amd $T/include
git -C $R archive $REV webdgs_amd/csrc include/webdgs.h | tar -x -C $T
make -C $T/webdgs_amd/csrc > $T/build.log 2>&1 || { tail -20 $T/build.log; exit 1; }
cp $T/webdgs_amd/lib/libwebdgs_hip.so $R/webdgs_amd/lib/libwebdgs_hip_prev.so
rm -rf $T
echo "built $REV -> webdgs_amd/lib/libwebdgs_hip_prev.so"
