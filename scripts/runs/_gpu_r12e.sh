# r12e: the whole GPU suite and smoke() on the final sources
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu --timeout 600 > $O/r12e_pytest_all.txt 2>&1; rc=$?; echo "pytest all rc=$rc"; tail -4 $O/r12e_pytest_all.txt
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
exit $rc
