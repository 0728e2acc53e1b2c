# r10: the whole GPU suite, then the round's final profile set
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x --timeout 600 > $O/r10_pytest_all.txt 2>&1; rc=$?; echo "pytest all rc=$rc"; tail -5 $O/r10_pytest_all.txt
[ $rc -eq 0 ] || exit 1
bash scripts/collect_profiles.sh r10 > $O/r10_collect.log 2>&1; rc=$?; tail -12 $O/r10_collect.log; exit $rc
