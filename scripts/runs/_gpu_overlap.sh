# same-box comparison of ways to make the batched (8 views per step) step overlap:  bash scripts/_gpu_overlap.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05c}
P=$GRAFT_REPO_ROOT/webdgs_amd/lib/libwebdgs_hip_prio3.so
run() {  # name, lanes, env...
  local name=$1 lanes=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --views-per-rank 8 --lanes $lanes --steps 10 --warmup 2 --sustained-steps 0 --no-cpu-baseline --no-profile --min-seconds 1.5 \
      > gpurun_out/${TAG}_${name}.json 2> gpurun_out/${TAG}_${name}.err || { echo "$name failed"; tail -5 gpurun_out/${TAG}_${name}.err; return 1; }
  python -c "
import json;d=json.load(open('gpurun_out/${TAG}_${name}.json'));print('%-22s lanes=%s  %8.1f views/s  %.4f ms/step  [%.4f .. %.4f] blocks=%d' % ('$name', '$lanes', d['value'], d['ms_per_step'], d['timed_blocks']['ms_per_step_min'], d['timed_blocks']['ms_per_step_max'], d['timed_blocks']['blocks']))"
}
for rep in 1 2; do
run base_$rep 3 WDGS_X=0 &&
run bwr4_$rep 3 WDGS_BWR_WPW=4 &&
run prio3_$rep 3 WDGS_LIB_PATH=$P &&
run bwr4_prio3_$rep 3 WDGS_BWR_WPW=4 WDGS_LIB_PATH=$P &&
run pad1700_$rep 3 WDGS_BWR_PAD_LDS=1700 &&
run rast1_prio3_$rep 3 WDGS_RASTER_WPW=1 WDGS_LIB_PATH=$P || exit 1
done
run bwr4_prio3_L4 4 WDGS_BWR_WPW=4 WDGS_LIB_PATH=$P
run bwr4_prio3_L2 2 WDGS_BWR_WPW=4 WDGS_LIB_PATH=$P
run base_L1 1 WDGS_X=0
