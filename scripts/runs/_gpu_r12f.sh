# r12f: soak of the long-list path: the batched step (8 views on 3 lanes) with the default densify schedule, threshold lowered so that it engages early
cd $GRAFT_REPO_ROOT
O=gpurun_out
WDGS_LONG_LISTS=1024 timeout -k 10 300 python scripts/soak.py c3 1500 8 none > $O/r12f_soak_vpr8_ll1024.txt 2>&1; echo "soak vpr8 ll1024 rc=$?"; tail -4 $O/r12f_soak_vpr8_ll1024.txt
WDGS_LONG_LISTS=1024 timeout -k 10 300 python scripts/soak.py c3 4000 1 none > $O/r12f_soak_vpr1_ll1024.txt 2>&1; echo "soak vpr1 ll1024 rc=$?"; tail -4 $O/r12f_soak_vpr1_ll1024.txt
grep -c "gave up waiting" $O/r12f_soak_vpr8_ll1024.txt $O/r12f_soak_vpr1_ll1024.txt
