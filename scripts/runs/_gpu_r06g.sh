#!/bin/bash
# forward rasterize per-wave timeline at c3 and c2, then the profile set of the final sources (tag r06g)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
for C in c3 c2; do
  rm -f /tmp/tlf.bin
  WDGS_FWR_TIMELINE=/tmp/tlf.bin WDGS_PROFILE_FROZEN=1 timeout -k 10 300 python3 scripts/profile_step.py $C 3 > $O/r06g_profile_$C.txt 2>&1
  python3 scripts/bwr_timeline.py /tmp/tlf.bin > $O/r06g_fwr_timeline_$C.txt 2>&1
  grep -v "^  xcc" $O/r06g_fwr_timeline_$C.txt | head -24
done
bash scripts/collect_profiles.sh r06g
