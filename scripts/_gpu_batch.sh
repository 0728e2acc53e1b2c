cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k c5_from > gpurun_out/r02_c5ply.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r02_c5ply.log
