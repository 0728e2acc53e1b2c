# c2: where backward_rasterize's 82 us go -- CU-level busy cycles, wave cycles, active VALU cycles; and the workgroup = tile form
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05f}; export TAG_=$TAG
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/c2 -- python3 scripts/profile_step.py c2 3 > $O/c2.log 2>&1 || { tail -5 $O/c2.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/c3 -- python3 scripts/profile_step.py c3 2 > $O/c3.log 2>&1 || { tail -5 $O/c3.log; exit 1; }
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_" + os.environ.get("TAG_", "r05f")
for leg in ("c2", "c3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"{root}/{leg}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].strip()
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
    print("==", leg, "(per launch)")
    for k in ("backward_rasterize_kernel", "rasterize_kernel", "segment_sort_kernel", "loss_grad_kernel"):
        v = acc[k]
        if n[k]: print(f"  {k[:30]:30s}", "  ".join(f"{c}={v[c]/n[k]/1e6:.3f}M" for c in sorted(v)))
PY
for w in 1 4; do
  WDGS_BWR_WPW=$w timeout -k 10 300 python bench.py --config c2 --sustained-steps 0 --no-cpu-baseline --no-batched-step --min-seconds 0.5 > gpurun_out/${TAG}_c2_wpw${w}.json 2> gpurun_out/${TAG}_c2.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_c2.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/${TAG}_c2_wpw${w}.json'));k=d['kernel_ms_per_view'];print('c2 WDGS_BWR_WPW=$w', d['value'], d['ms_per_step'], {a:k[a] for a in ('backward_rasterize','rasterize') if a in k})"
done
