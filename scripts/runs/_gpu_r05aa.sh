#!/bin/bash
# round-4 closing set after the issue-priority change: full GPU suite, c2 timeline, profile collection (tag r05ab)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/r05aa_pytest.txt 2>&1 || { tail -40 $O/r05aa_pytest.txt; exit 1; }
tail -3 $O/r05aa_pytest.txt
rm -f /tmp/tl.bin
WDGS_BWR_TIMELINE=/tmp/tl.bin WDGS_PROFILE_FROZEN=1 timeout -k 10 300 python3 scripts/profile_step.py c2 3 > $O/r05aa_profile_c2.txt 2>&1
python3 scripts/bwr_timeline.py /tmp/tl.bin > $O/r05aa_bwr_timeline_c2_prio.txt 2>&1
grep -v "^  xcc" $O/r05aa_bwr_timeline_c2_prio.txt | head -30
