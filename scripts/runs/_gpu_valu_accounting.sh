# VALU instruction counts (SQ_INSTS_VALU, one --pmc pass each) of the c2 step and of the batched c3 step:  bash scripts/_gpu_valu_accounting.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05d}; export TAG_=$TAG
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/c2 -- python3 scripts/profile_step.py c2 3 > $O/c2.log 2>&1 || { tail -5 $O/c2.log; exit 1; }
WDGS_PROFILE_VPR=8 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/c3_vpr8 -- python3 scripts/profile_step.py c3 2 > $O/c3_vpr8.log 2>&1 || { tail -5 $O/c3_vpr8.log; exit 1; }
python3 - <<'PY'
import csv, glob, os, re, collections
root = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_%s" % os.environ.get("TAG_", "r05d")
for leg in ("c2", "c3_vpr8"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"{root}/{leg}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0].strip()
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
    print("==", leg, "(per launch)")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
        if n[k]: print(f"  {k[:34]:34s} launches {n[k]:4d}  VALU {v['SQ_INSTS_VALU']/n[k]/1e6:9.3f} M  waves {v['SQ_WAVES']/n[k]:9.0f}  busy_cycles {v['SQ_BUSY_CYCLES']/n[k]/1e6:7.3f} M  wait_inst_any {v['SQ_WAIT_INST_ANY']/n[k]/1e6:8.2f} M")
PY
