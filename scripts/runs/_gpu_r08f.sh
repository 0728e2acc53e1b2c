# r08f: full GPU suite with the non-finite changes, then same-box A/B (c3, c2) of the new build against round 4's (libwebdgs_hip_prev.so = 3cc455c)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -q -m gpu -x --timeout 600 > gpurun_out/r08f_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r08f_pytest.txt
bash scripts/runs/_gpu_ab_lib.sh r08f_c3 backward_rasterize rasterize project_count scan_forward
for rep in 1 2; do for v in prev new; do
  L=$GRAFT_REPO_ROOT/webdgs_amd/lib/libwebdgs_hip.so; [ $v = prev ] && L=$GRAFT_REPO_ROOT/webdgs_amd/lib/libwebdgs_hip_prev.so
  WDGS_LIB_PATH=$L timeout -k 10 300 python bench.py --config c2 --sustained-steps 0 --no-cpu-baseline --min-seconds 0.5 > gpurun_out/r08f_c2_${v}_${rep}.json 2> gpurun_out/r08f_c2.err || { echo "bench failed"; tail -5 gpurun_out/r08f_c2.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/r08f_c2_${v}_${rep}.json'));k=d['kernel_ms_per_view'];print('c2 $v rep=$rep', d['value'], d['ms_per_step'], {a:k[a] for a in 'backward_rasterize rasterize project_count'.split() if a in k})"
done; done
