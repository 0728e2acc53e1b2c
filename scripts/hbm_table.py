#!/usr/bin/env python3
"""profiles/<tag>_pmc.json + a rocprofv3 kernel_stats.csv -> the per-kernel HBM / VALU table (markdown) kept under profiles/.

    python scripts/hbm_table.py profiles/r02f_pmc.json gpurun_out/r02f/prof/runc/*_kernel_stats.csv r02f > profiles/r02f_hbm_by_kernel.md"""
import csv
import json
import re
import sys

pmc = json.load(open(sys.argv[1]))
stats, tag = sys.argv[2], sys.argv[3]
dur = {}
for r in csv.DictReader(open(stats)):
    m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(<[^()]*>)?\s*\((?!anonymous)", r["Name"])
    name = (m.group(1) if m else r["Name"])
    if name == "geometry_backward_kernel":  # the template argument selects what K17 goes on to do with its gradient
        name = {"<1>": "geometry_backward_accumulate_kernel", "<2>": "geometry_backward_adam_kernel"}.get((m.group(2) or "") if m else "", name)
    name = name[:-len("_kernel")] if name.endswith("_kernel") else name
    if name not in dur:
        dur[name] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
cal = pmc.get("calibration", {})
print(f"# Per-kernel HBM traffic and VALU issue at c3-perf ({tag}; counters at {pmc.get('head', '?')})\n")
print(f"Average duration: `rocprofv3 --kernel-trace --stats` of `bench.py` (`{tag}_c3_kernel_stats.csv`). Counters: separate `--pmc` passes (`scripts/pmc.sh`,")
print(f"`{tag}_pmc.json`). HBM bytes per launch = `2*FETCH_SIZE + WRITE_SIZE` KiB: the read-side factor 2 is re-calibrated on this repo's own kernels whose read volume is")
print("known exactly -- " + ", ".join(f"{k} {v['ratio']:.2f}" for k, v in cal.items()) + " -- and `TCC_EA0_RDREQ_32B` is 0 for every kernel: every read request is a full")
print("128-byte line fill counted as 64 bytes, whatever the width of the access that missed, so the factor applies to the gather kernels (rasterize,")
print("backward_rasterize) as well. VALU issue share = `SQ_INSTS_VALU` x 2 cycles / 1024 SIMDs / 2.4 GHz over the duration.\n")
print("| kernel | launches in trace | avg us | HBM MB / launch | GB/s | of 8.0 TB/s | VALU wave-instr / launch | VALU issue share | memory-side atomics |")
print("|---|---|---|---|---|---|---|---|---|")
rows = []
for k, m in pmc["kernels"].items():
    if k not in dur or "hbm_bytes" not in m:
        continue
    calls, us = dur[k]
    gbs = m["hbm_bytes"] / (us * 1e-6) / 1e9
    valu = m.get("SQ_INSTS_VALU", 0.0)
    share = valu * 2.0 / 1024 / 2.4e9 / (us * 1e-6)
    rows.append((us * calls, f"| {k} | {calls} | {us:.1f} | {m['hbm_bytes'] / 1e6:.1f} | {gbs:.0f} | {gbs / 8000:.2f} | {valu / 1e6:.1f} M | {share:.2f} | {m.get('TCC_ATOMIC_sum', 0):.0f} |"))
for _, line in sorted(rows, reverse=True):
    print(line)
