#!/bin/bash
# depth-split backward_rasterize: parity suites, then c2 A/B over the number of roles, timeline at 2 roles
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer_oracle.py tests/test_gpu_edges.py tests/test_gpu_ops.py -x -q -m gpu > $O/r05s_pytest.txt 2>&1 || { tail -30 $O/r05s_pytest.txt; exit 1; }
tail -3 $O/r05s_pytest.txt
for R in 1 2 3 4 2 1; do
  WDGS_BWR_ROLES=$R timeout -k 10 300 python3 bench.py --config c2 --sustained-steps 0 --no-cpu-baseline > $O/r05s_bench_c2_roles$R.json 2> $O/r05s_bench.err
  python3 -c "
import json;d=json.load(open('$O/r05s_bench_c2_roles$R.json'));print('roles',$R,d['value'],d['ms_per_step'],d['roofline']['kernel'],d['roofline'].get('avg_us'), {k:v for k,v in d.get('kernels',{}).items()} if 'kernels' in d else '')"
done
rm -f /tmp/tl.bin
WDGS_BWR_ROLES=2 WDGS_BWR_TIMELINE=/tmp/tl.bin WDGS_PROFILE_FROZEN=1 timeout -k 10 300 python3 scripts/profile_step.py c2 3 > $O/r05s_profile_c2.txt 2>&1
python3 scripts/bwr_timeline.py /tmp/tl.bin > $O/r05s_bwr_timeline_c2_roles2.txt 2>&1
grep -v "^  xcc" $O/r05s_bwr_timeline_c2_roles2.txt
