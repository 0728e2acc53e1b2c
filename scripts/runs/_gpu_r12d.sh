# r12d: the 10 000-iteration c3 run with the long-list path on and off, same box, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out
F="--no-cpu-baseline --no-batched-step --no-profile --min-seconds 0.5"
for rep in 1 2; do
  timeout -k 10 200 python bench.py $F > $O/r12d_on_$rep.json 2>> $O/r12d.err || exit 1
  WDGS_LONG_LISTS=0 timeout -k 10 200 python bench.py $F > $O/r12d_off_$rep.json 2>> $O/r12d.err || exit 1
done
python - <<'PY' > gpurun_out/r12d_full_run_long_lists_on_off.txt
import json
print("c3, the reference's schedule, 10 000 iterations (bench.py full_run), same box, alternating: long tile lists on (default) / off (WDGS_LONG_LISTS=0); it/s per window of 1 000 iterations")
for who in ("on", "off"):
    for rep in (1, 2):
        j = json.loads(open(f"gpurun_out/r12d_{who}_{rep}.json").read().strip().splitlines()[-1])
        f = j["full_run"]
        print(f"{who:3s} overall {f['iters_per_s_overall']:7.1f}  as written (620) {j['c3_as_written_iters_per_s']:7.1f}  windows " + " ".join(f"{w['iters_per_s']:.0f}" for w in f["windows"]))
PY
cat gpurun_out/r12d_full_run_long_lists_on_off.txt
