#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc CSVs (one directory per pass) into a per-kernel table: mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for k in vals.values() for c in k})
keep = [k for k in vals if any(s in k for s in ("rasterize", "project_count", "adam_repack", "sort_", "emit", "loss_grad", "geometry", "scan_", "tile_ranges"))]
print("mean per dispatch (dispatch counts may differ per pass)")
for k in sorted(keep, key=lambda k: -sum(vals[k].get("SQ_BUSY_CYCLES", [0]))):
    print(f"== {k}  (n={len(next(iter(vals[k].values())))})")
    for c in counters:
        if c in vals[k]:
            v = vals[k][c]
            print(f"   {c:28s} {sum(v) / len(v):16.1f}")
