# r08p: long-list tasks with relaxed agent-scope accesses instead of fences: tests, synthetic scenes, late regime
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_nan.py tests/test_gpu_edges.py tests/test_gpu_parity.py -q -m gpu -x --timeout 120 > $O/r08p_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r08p_pytest.txt
timeout -k 10 120 python scripts/long_list_scenes.py 10400 40000 > $O/r08p_long_list_scenes.txt 2>&1; tail -6 $O/r08p_long_list_scenes.txt
timeout -k 10 300 python3 scripts/late_regime_profile.py c3 6000 > $O/r08p_late_regime.txt 2>&1; grep -E "after|rasterize|sort_segments|kernel sum" $O/r08p_late_regime.txt
