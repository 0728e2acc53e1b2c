cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_napi.py -q -x > gpurun_out/r02g_napi.log 2>&1; echo "napi pytest rc=$?"; tail -30 gpurun_out/r02g_napi.log
timeout -k 10 120 node bindings/napi/smoke.js gpu; echo "smoke rc=$?"
