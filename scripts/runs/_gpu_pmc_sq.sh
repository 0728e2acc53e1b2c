# the two SQ counter passes only (instruction mix, waits, LDS):  bash scripts/_gpu_pmc_sq.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_${TAG}
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${TAG}/$name -- python3 $R/scripts/profile_step.py c3 3 > $R/gpurun_out/pmc_${TAG}/$name.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${TAG}/$name.log; return 1; }
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY && \
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 && \
run sq3 SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_INSTS_SMEM
python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_${TAG} > $R/gpurun_out/pmc_${TAG}_summary.txt
grep -A30 "== backward_rasterize" $R/gpurun_out/pmc_${TAG}_summary.txt | head -40
