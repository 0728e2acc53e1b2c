#!/bin/bash
# densify event cost after the single-allocation long-list scratch; same-box A/B of c5 / vpr8 with the long lists off
set -e
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --no-batched-step --full-run-steps 0 --min-seconds 2 > gpurun_out/r08u_c3.json 2> gpurun_out/r08u_c3.err
python bench.py --config c5 --no-cpu-baseline --no-batched-step --full-run-steps 0 --sustained-steps 0 --min-seconds 2 > gpurun_out/r08u_c5_on.json 2>> gpurun_out/r08u_c3.err
WDGS_LONG_LISTS=0 python bench.py --config c5 --no-cpu-baseline --no-batched-step --full-run-steps 0 --sustained-steps 0 --min-seconds 2 > gpurun_out/r08u_c5_off.json 2>> gpurun_out/r08u_c3.err
python bench.py --views-per-rank 8 --no-cpu-baseline --no-batched-step --full-run-steps 0 --sustained-steps 0 --min-seconds 2 > gpurun_out/r08u_vpr8_on.json 2>> gpurun_out/r08u_c3.err
WDGS_LONG_LISTS=0 python bench.py --views-per-rank 8 --no-cpu-baseline --no-batched-step --full-run-steps 0 --sustained-steps 0 --min-seconds 2 > gpurun_out/r08u_vpr8_off.json 2>> gpurun_out/r08u_c3.err
python bench.py --config c5 --no-cpu-baseline --no-batched-step --full-run-steps 0 --sustained-steps 0 --min-seconds 2 > gpurun_out/r08u_c5_on2.json 2>> gpurun_out/r08u_c3.err
echo done
