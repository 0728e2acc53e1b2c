#!/usr/bin/env python3
"""How much of a batched step's device time ran on more than one lane at once, from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --views-per-rank 8 --no-profile ...
    python3 scripts/lane_overlap.py <dir> [last_n_kernels]

Reads every *kernel_trace.csv under <dir> (columns Kernel_Name, Start_Timestamp, End_Timestamp, Queue_Id / Stream_Id), keeps the
LAST n kernel records (default: all; the timed region of bench.py is the tail of the trace) and prints: the wall span they cover,
the sum of their durations, the time during which 1, 2, 3.. kernels were in flight, and the per-kernel average durations (a kernel
that shares the machine runs longer than alone: compare with the one-lane profile)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short_name(full: str) -> str:
    """`(anonymous namespace)::backward_rasterize_kernel(RenderSettings, ...)` -> `backward_rasterize_kernel`"""
    m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(<[^()]*>)?\s*\((?!anonymous)", full)
    return m.group(1) if m else full[:60]


def main():
    root = sys.argv[1]
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = []
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short_name(r["Kernel_Name"]), r.get("Queue_Id", r.get("Stream_Id", "?"))))
    rows.sort()
    if last:
        rows = rows[-last:]
    if not rows:
        print("no kernel records")
        return
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    events = sorted([(s, 1) for s, _, _, _ in rows] + [(e, -1) for _, e, _, _ in rows])
    depth, prev, at_depth = 0, t0, defaultdict(int)
    for t, d in events:
        at_depth[depth] += t - prev
        depth, prev = depth + d, t
    busy = sum(e - s for s, e, _, _ in rows)
    span = t1 - t0
    print(f"kernels {len(rows)}  queues {len(set(r[3] for r in rows))}  span {span / 1e6:.3f} ms  sum of durations {busy / 1e6:.3f} ms  "
          f"mean kernels in flight {busy / span:.2f}")
    for k in sorted(at_depth):
        print(f"  {k} in flight: {at_depth[k] / 1e6:8.3f} ms  {100.0 * at_depth[k] / span:5.1f} %")
    per = defaultdict(lambda: [0, 0])
    for s, e, name, _ in rows:
        per[name][0] += 1
        per[name][1] += e - s
    print("kernel, launches, avg us")
    for name, (n, total) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"  {name[:60]:60s} {n:6d} {total / n / 1e3:9.1f}")


if __name__ == "__main__":
    main()
