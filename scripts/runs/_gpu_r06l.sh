#!/bin/bash
# row-pass partitions of 2048 keys (1024 for small sorters) against 4096 (prev = HEAD's build): tests, then same-box A/B at c2, c3, c5 and the 8-view step
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu -k "not typescript_side" > $O/r06l_pytest.txt 2>&1 || { tail -30 $O/r06l_pytest.txt; exit 1; }
tail -1 $O/r06l_pytest.txt
for rep in 1 2; do for v in prev new; do
  L=$PWD/webdgs_amd/lib/libwebdgs_hip.so; [ $v = prev ] && L=$PWD/webdgs_amd/lib/libwebdgs_hip_prev.so
  for C in c2 c3 c5; do
    X=""; [ $C = c5 ] && X="--steps 10 --warmup 2"
    WDGS_LIB_PATH=$L timeout -k 10 400 python3 bench.py --config $C --sustained-steps 0 --no-cpu-baseline --no-batched-step $X > $O/r06l_${C}_${v}_${rep}.json 2> $O/r06l.err
    python3 -c "
import json;d=json.load(open('$O/r06l_${C}_${v}_${rep}.json'));k=d['kernel_ms_per_view'];print('$C $v rep=$rep', d['value'], d['ms_per_step'], k.get('sort'))"
  done
  WDGS_LIB_PATH=$L timeout -k 10 400 python3 bench.py --views-per-rank 8 --lanes 3 --steps 10 --warmup 2 --sustained-steps 0 --no-cpu-baseline > $O/r06l_vpr8_${v}_${rep}.json 2> $O/r06l.err
  python3 -c "
import json;d=json.load(open('$O/r06l_vpr8_${v}_${rep}.json'));print('vpr8 $v rep=$rep', d['value'], d['ms_per_step'])"
done; done
