#!/bin/bash
# allocation cache: free / event costs in both hosts, then the full GPU suite
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
{ echo "# buffer create / destroy (50 x 1 MB), ms per call: the node host (ROCm 7.2 runtime of /opt/rocm) and the Python host (the 7.0 runtime PyTorch loads); cache on / off";
  for C in 1 0; do echo "WDGS_ALLOC_CACHE=$C"; WDGS_ALLOC_CACHE=$C node scripts/probes/free_cost.js; WDGS_ALLOC_CACHE=$C python3 scripts/probes/free_cost.py 2>/dev/null; done; } > $O/r06s_free_cost.txt 2>&1
cat $O/r06s_free_cost.txt
for C in 1 0; do echo "== densify event, node, WDGS_ALLOC_CACHE=$C"; WDGS_ALLOC_CACHE=$C timeout -k 10 300 node bindings/napi/densify_timing.js c3 2>&1 | grep -E "^applyPointCloudSwap|^runDensify|steps around|Optimizer.destroy|ensureSize|HipBuffer.destroy"; done > $O/r06s_densify_event_node.txt 2>&1
cat $O/r06s_densify_event_node.txt
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/r06s_pytest.txt 2>&1 || { tail -40 $O/r06s_pytest.txt; exit 1; }
tail -3 $O/r06s_pytest.txt
