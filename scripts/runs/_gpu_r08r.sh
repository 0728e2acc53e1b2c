# r08r: trainer-level long-list tests (oracle trajectory, three lanes), then the whole GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_trainer_oracle.py tests/test_gpu_lanes.py -q -m gpu -x --timeout 300 > $O/r08r_pytest1.txt 2>&1; echo "pytest1 rc=$?"; tail -4 $O/r08r_pytest1.txt
timeout -k 10 1000 python -m pytest tests -q -m gpu -x --timeout 600 > $O/r08r_pytest_all.txt 2>&1; echo "pytest all rc=$?"; tail -4 $O/r08r_pytest_all.txt
