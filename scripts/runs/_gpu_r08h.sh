# r08h: EXACT forward with the dead-state filter: non-finite parity cases, late-regime kernel times, the 10 000-iteration run
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_nan.py tests/test_gpu_parity.py -q -m gpu -x --timeout 300 > $O/r08h_pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/r08h_pytest.txt
timeout -k 10 300 python3 scripts/late_regime_profile.py c3 6000 > $O/r08h_late_regime_profile.txt 2>&1; tail -14 $O/r08h_late_regime_profile.txt
timeout -k 10 600 python3 scripts/train_demo.py c3 10000 8 1.0 1000 > $O/r08h_train_demo_c3_10000.txt 2>&1; tail -11 $O/r08h_train_demo_c3_10000.txt
