cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python scripts/profile_step.py c3 10 > gpurun_out/r02f_profile_noorder.txt 2>&1; head -8 gpurun_out/r02f_profile_noorder.txt
WDGS_TILE_ORDER=1 python scripts/profile_step.py c3 10 > gpurun_out/r02f_profile_order.txt 2>&1; head -8 gpurun_out/r02f_profile_order.txt; grep "tile_order\|scan_block\|kernel sum" gpurun_out/r02f_profile_order.txt
WDGS_TILE_ORDER=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3
