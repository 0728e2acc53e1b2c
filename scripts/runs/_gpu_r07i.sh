#!/bin/bash
# opt-in long-list passes of rasterize (and backward_rasterize): parity with every list above 64 entries on those paths, then the late-regime step
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
WDGS_FWR_LONG=64 WDGS_BWR_LONG=64 timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer_oracle.py tests/test_gpu_edges.py tests/test_gpu_viewer.py -x -q -m gpu > $O/r07i_pytest.txt 2>&1 || { tail -40 $O/r07i_pytest.txt; exit 1; }
tail -2 $O/r07i_pytest.txt
for L in 0 4096; do
  echo "== late regime, WDGS_FWR_LONG=$L WDGS_BWR_LONG=$L"
  WDGS_FWR_LONG=$L WDGS_BWR_LONG=$L timeout -k 10 400 python3 scripts/late_regime_profile.py c3 6000 2>/dev/null | grep -v amdgpu.ids | head -16
done > $O/r07i_late_regime_long_lists.txt 2>&1
cat $O/r07i_late_regime_long_lists.txt
