# r09c: the whole GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -x --timeout 600 > $O/r09c_pytest_all.txt 2>&1; echo "pytest all rc=$?"; tail -6 $O/r09c_pytest_all.txt
