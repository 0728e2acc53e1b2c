#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out; mkdir -p $O
for L in 1 0; do echo "== node densify event, WDGS_LAZY_GRAPH_DESTROY=$L"; WDGS_LAZY_GRAPH_DESTROY=$L timeout -k 10 300 node bindings/napi/densify_timing.js c3 2>&1 | grep -E "^applyPointCloudSwap|^invalidate|steps around|steps 36|total ms"; done > $O/r06t_lazy_graph_destroy_node.txt 2>&1
cat $O/r06t_lazy_graph_destroy_node.txt
