#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/ (scripts/pmc.sh) -> profiles/<tag>_pmc.json: per-kernel mean PMC values per dispatch, plus HBM traffic.

HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes).  The factor 2 on the read side is the MI355X guide's gfx950 correction (FETCH_SIZE
= TCC_EA0_RDREQ x 64 B while the requests are 128-byte line fills).  The guide calibrates it for 16-byte-per-lane streaming reads only; round 2
re-calibrated it on this repo's own kernels whose read volume is known exactly (`calibration` in the output: known bytes / FETCH_SIZE):
dword and 8-byte coalesced loads (sort_hist, emit, geometry_backward, segment_sort) give 1.94-2.05 as well, and TCC_EA0_RDREQ_32B is 0 for
every kernel -- every read request is a full line, whatever the width of the access that missed -- so the factor is applied to the gather
kernels (rasterize, backward_rasterize) too.  `head` records the commit the counters were taken at."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        full = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        name = full.split("<")[0]
        if name == "geometry_backward_kernel" and "<" in full:  # the template argument selects what K17 goes on to do with its gradient
            name = {"<1>": "geometry_backward_accumulate_kernel", "<2>": "geometry_backward_adam_kernel"}.get(full[full.index("<"):], name)
        if name.endswith("_kernel"):
            name = name[: -len("_kernel")]
        vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in vals.items():
    if k.startswith("at::") or "rocclr" in k:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        m["hbm_bytes"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
        m["hbm_bytes_corrected"] = m["hbm_bytes"]  # (name used by round-1 tooling)
        m["hbm_bytes_raw"] = (m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
    res[k] = m

# calibration of the read-side factor on kernels whose read volume is known (c3: the step's E, N, V are in the run's log)
calib = {}
try:
    log = open(os.path.join(root, "sq1.log")).read()
    import re
    mm = re.search(r"N=(\d+) E=(\d+) V=(\d+)", log)
    if mm:
        n, e, v = (int(x) for x in mm.groups())
        known = {"sort_hist": 4 * e, "segment_sort": 8 * e, "geometry_backward": 72 * n, "emit": 8 * n + 28 * v}
        for k, b in known.items():
            if k in res and res[k].get("FETCH_SIZE"):
                calib[k] = dict(known_read_bytes=b, fetch_size_bytes=res[k]["FETCH_SIZE"] * 1024.0, ratio=round(b / (res[k]["FETCH_SIZE"] * 1024.0), 3),
                                rdreq_32B=res[k].get("TCC_EA0_RDREQ_32B_sum"))
except OSError:
    pass
try:
    head = subprocess.check_output(["git", "-C", os.path.dirname(os.path.abspath(__file__)), "rev-parse", "--short=12", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
except Exception:
    head = ""
# hash of every kernel's source file at the time the counters were taken: bench.py refuses a profile whose hash for the dominant
# kernel differs from the working tree's
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_sha  # noqa: E402
shas = {k: source_sha(k) for k in list(res) + ["sort", "scan"] if source_sha(k)}
json.dump(dict(source=os.path.basename(root.rstrip("/")), head=head, workload=(sys.argv[3] if len(sys.argv) > 3 else "c3"), source_sha=shas,
               note="mean per dispatch; FETCH_SIZE/WRITE_SIZE in KiB; hbm_bytes = 2*FETCH + WRITE (gfx950 line fills are 128 B, counted as 64)",
               calibration=calib, kernels=res), open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, "kernels:", len(res), "calibration:", {k: c["ratio"] for k, c in calib.items()})
