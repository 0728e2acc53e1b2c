#!/usr/bin/env python3
"""The longest waves of the LAST rasterize launch in a WDGS_FWR_TIMELINE file (dev tool): record index = tile * 4 + block.
    WDGS_FWR_TIMELINE=/tmp/tl.bin python scripts/late_regime_profile.py c3 6000 && python scripts/fwr_timeline_top.py /tmp/tl.bin"""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from bwr_timeline import launches  # noqa: E402

slots, tiles, rec = list(launches(sys.argv[1]))[-1]
ran = np.flatnonzero(rec[:, 1] != 0)
t0, t1, it = rec[ran, 0].astype(np.int64), rec[ran, 1].astype(np.int64), rec[ran, 3].astype(np.int64)
base = t0.min()
dur = (t1 - t0) * 0.01
print(f"launch span {(t1.max() - base) * 0.01:.1f} us, {len(ran)} waves with work; wave life mean {dur.mean():.1f} p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f} us; records composited: total {it.sum()}")
order = np.argsort(-dur)[:12]
for o in order:
    print(f"  tile {ran[o] // 4:5d} block {ran[o] % 4}: start {(t0[o] - base) * 0.01:7.1f} us, life {dur[o]:7.1f} us, records composited {it[o]}")
for b in range(4):
    w = np.flatnonzero(ran == b)
    if len(w):
        print(f"  tile 0 block {b}: start {(t0[w[0]] - base) * 0.01:7.1f} us, life {dur[w[0]]:7.1f} us, records composited {it[w[0]]}")
