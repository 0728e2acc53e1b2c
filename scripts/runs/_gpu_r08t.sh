# r08t: all-or-nothing long lists + batched tickets: parity tests of the path, event timing, scenes, c3 sustained
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_nan.py tests/test_gpu_trainer_oracle.py tests/test_gpu_lanes.py -q -m gpu -x --timeout 300 > $O/r08t_pytest1.txt 2>&1; rc=$?; echo "pytest1 rc=$rc"; tail -4 $O/r08t_pytest1.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python scripts/event_timing.py > $O/r08t_event_timing.txt 2>&1 || exit 1
timeout -k 10 300 python scripts/long_list_scenes.py > $O/r08t_long_list_scenes.txt 2>&1 || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-batched-step --full-run-steps 0 --min-seconds 2 > $O/r08t_c3.json 2> $O/r08t_c3.err || exit 1
echo done
