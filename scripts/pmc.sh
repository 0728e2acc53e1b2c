#!/bin/bash
# PMC passes for the training step kernels (run on the GPU box through gpurun):  bash scripts/pmc.sh <tag> [config] [steps]
# Separate --pmc passes, kernel-trace only (no sys/hip traces), as the MI355X guide prescribes.
set -e
TAG=${1:-r01}; CFG=${2:-c3}; STEPS=${3:-3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${TAG}/$name -- python3 $R/scripts/profile_step.py $CFG $STEPS > $R/gpurun_out/pmc_${TAG}/$name.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${TAG}/$name.log; return 1; }
}
mkdir -p $R/gpurun_out/pmc_${TAG}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE TCC_ATOMIC_sum
# request-size split of the memory-side traffic (FETCH_SIZE x2 is calibrated for wide streaming reads only); the counter
# names differ between ROCm releases, so these passes may fail without failing the script
rocprofv3 -L > $R/gpurun_out/pmc_${TAG}/counters.txt 2>&1 || true
run ea_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum || true
run ea_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum || true
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || true
python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_${TAG} > $R/gpurun_out/pmc_${TAG}/summary.txt
cat $R/gpurun_out/pmc_${TAG}/summary.txt
