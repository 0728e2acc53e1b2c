#!/bin/bash
# c3 / c5-like: backward_rasterize's tiles in raster order against descending work (host-computed order, timeline form): how much of the span is dispatch order?
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
for ORDER in 0 1; do
  rm -f /tmp/tl.bin
  if [ $ORDER = 1 ]; then export WDGS_BWR_ORDER=1; fi
  WDGS_BWR_TIMELINE=/tmp/tl.bin WDGS_PROFILE_FROZEN=1 timeout -k 10 400 python3 scripts/profile_step.py c3 3 > $O/r05ab_profile_c3_order$ORDER.txt 2>&1
  python3 scripts/bwr_timeline.py /tmp/tl.bin > $O/r05ab_bwr_timeline_c3_order$ORDER.txt 2>&1
  grep -v "^  xcc" $O/r05ab_bwr_timeline_c3_order$ORDER.txt | head -12
done
