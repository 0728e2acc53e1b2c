#!/bin/bash
# lone-wave iteration time of backward_rasterize and its growth with the waves per SIMD; the same for the ablated builds (timing only)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
export WDGS_BWR_TIMELINE=/tmp/tl.bin WDGS_BWR_ROLES=1
echo "== product kernel" > $O/r05v_wave_rate.txt
timeout -k 10 300 python3 scripts/bwr_wave_rate.py >> $O/r05v_wave_rate.txt 2>$O/r05v.err
echo "== built without the global atomics" >> $O/r05v_wave_rate.txt
WDGS_LIB_PATH=$PWD/webdgs_amd/lib/libwebdgs_hip_noat.so timeout -k 10 300 python3 scripts/bwr_wave_rate.py >> $O/r05v_wave_rate.txt 2>>$O/r05v.err
echo "== built without the LDS transposition and DPP steps of the sums" >> $O/r05v_wave_rate.txt
WDGS_LIB_PATH=$PWD/webdgs_amd/lib/libwebdgs_hip_nosum.so timeout -k 10 300 python3 scripts/bwr_wave_rate.py >> $O/r05v_wave_rate.txt 2>>$O/r05v.err
cat $O/r05v_wave_rate.txt
