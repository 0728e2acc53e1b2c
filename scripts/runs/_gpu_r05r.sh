#!/bin/bash
# per-wave timeline of backward_rasterize at c2 (and c3 beside it)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
for C in c2 c3; do
  rm -f /tmp/tl_$C.bin
  WDGS_BWR_TIMELINE=/tmp/tl_$C.bin WDGS_PROFILE_FROZEN=1 timeout -k 10 300 python3 scripts/profile_step.py $C 3 > $O/r05r_profile_$C.txt 2>&1
  python3 scripts/bwr_timeline.py /tmp/tl_$C.bin > $O/r05r_bwr_timeline_$C.txt 2>&1
  cat $O/r05r_bwr_timeline_$C.txt
  python3 -c "import sys;sys.path.insert(0,'scripts');import bwr_timeline as b,struct;L=list(b.launches('/tmp/tl_$C.bin'))[-1];open('$O/r05r_tl_$C.bin','wb').write(struct.pack('<II',L[0],L[1])+L[2].tobytes())"
done
