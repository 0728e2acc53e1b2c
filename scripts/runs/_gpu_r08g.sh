# r08g: the late regime under the pinned non-finite semantics: list statistics and kernel times after 6 000 iterations, and the 10 000-iteration run
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python3 scripts/long_list_stats.py c3 6000 3 > $O/r08g_long_list_stats.txt 2>&1; tail -30 $O/r08g_long_list_stats.txt
timeout -k 10 300 python3 scripts/late_regime_profile.py c3 6000 > $O/r08g_late_regime_profile.txt 2>&1; tail -16 $O/r08g_late_regime_profile.txt
timeout -k 10 600 python3 scripts/train_demo.py c3 10000 8 1.0 1000 > $O/r08g_train_demo_c3_10000.txt 2>&1; tail -12 $O/r08g_train_demo_c3_10000.txt
