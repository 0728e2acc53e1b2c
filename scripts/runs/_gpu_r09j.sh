# r09j: oversized segments with LDS-only barriers: tests, late-regime profile
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_nan.py tests/test_gpu_edges.py tests/test_gpu_ops.py -q -m gpu -x --timeout 300 > $O/r09j_pytest.txt 2>&1 || { tail -15 $O/r09j_pytest.txt; exit 1; }
tail -2 $O/r09j_pytest.txt
timeout -k 10 400 python scripts/late_regime_profile.py > $O/r09j_late_regime.txt 2>&1 || { tail -5 $O/r09j_late_regime.txt; exit 1; }
tail -15 $O/r09j_late_regime.txt
timeout -k 10 300 python scripts/long_list_scenes.py > $O/r09j_long_list_scenes.txt 2>&1; tail -6 $O/r09j_long_list_scenes.txt
