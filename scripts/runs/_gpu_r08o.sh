# r08o: the late regime (c3 after 6 000 iterations) and the headline with the long-list tasks on (default) and off (WDGS_LONG_LISTS=0)
cd $GRAFT_REPO_ROOT
O=gpurun_out
for LL in 2048 0; do
  echo "== WDGS_LONG_LISTS=$LL"
  WDGS_LONG_LISTS=$LL timeout -k 10 300 python3 scripts/late_regime_profile.py c3 6000 > $O/r08o_late_regime_LL$LL.txt 2>&1; grep -E "after|rasterize|sort_segments|kernel sum" $O/r08o_late_regime_LL$LL.txt
  WDGS_LONG_LISTS=$LL timeout -k 10 300 python bench.py --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.5 > $O/r08o_bench_c3_LL$LL.json 2> $O/r08o.err || tail -3 $O/r08o.err
  python -c "
import json;d=json.load(open('gpurun_out/r08o_bench_c3_LL$LL.json'));k=d['kernel_ms_per_view'];print('c3', d['value'], d['ms_per_step'], {a:k[a] for a in 'backward_rasterize rasterize sort'.split() if a in k})"
  WDGS_LONG_LISTS=$LL timeout -k 10 300 python bench.py --config c2 --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.5 > $O/r08o_bench_c2_LL$LL.json 2> $O/r08o.err || tail -3 $O/r08o.err
  python -c "
import json;d=json.load(open('gpurun_out/r08o_bench_c2_LL$LL.json'));k=d['kernel_ms_per_view'];print('c2', d['value'], d['ms_per_step'], {a:k[a] for a in 'backward_rasterize rasterize sort'.split() if a in k})"
done
