# r08m: first run of the long-list tasks (default threshold 2048): edge / NaN / parity tests, then the synthetic long-list scenes timed
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_nan.py tests/test_gpu_edges.py tests/test_gpu_parity.py -q -m gpu -x --timeout 120 > $O/r08m_pytest.txt 2>&1; echo "pytest rc=$?"; tail -12 $O/r08m_pytest.txt
timeout -k 10 120 python scripts/long_list_scenes.py 10400 40000 > $O/r08m_long_list_scenes.txt 2>&1; tail -7 $O/r08m_long_list_scenes.txt
