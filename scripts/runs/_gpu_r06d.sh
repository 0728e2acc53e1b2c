#!/bin/bash
# the awaited step (the reference's own form): hipEventSynchronize against polling hipEventQuery
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
for rep in 1 2; do for S in 0 1; do for C in c3 c2; do
  WDGS_WAIT_SPIN=$S timeout -k 10 300 python3 bench.py --config $C --sustained-steps 0 --no-cpu-baseline --no-batched-step > $O/r06d_${C}_spin${S}_${rep}.json 2> $O/r06d.err
  python3 -c "
import json;d=json.load(open('$O/r06d_${C}_spin${S}_${rep}.json'));print('$C spin=$S rep=$rep', d['value'], d['ms_per_step'], 'awaiting every step:', d['ms_per_step_awaiting_every_step'])"
done; done; done
