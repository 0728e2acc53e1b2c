#!/usr/bin/env python3
"""Dev probe: what a dependent kernel boundary costs inside a replayed command buffer (HIP graph).   python scripts/graph_gap_probe.py

Records n back-to-back launches of a one-wave kernel (guard_accumulate) and times the replays; the slope is the floor a chain of
small kernels pays per launch (MI355X, ROCm 7.2: 11 us for a graph of one, 1.6 us per further kernel -- DESIGN.md section 6)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webdgs_amd import ops  # noqa: E402

dev = ops.HipDevice(0)
flag, stats = dev.createBuffer(16), dev.createBuffer(64)
for n in (1, 16, 64, 256):
    with dev.createCommandEncoder("gap", record=True) as enc:
        for _ in range(n):
            ops.guardAccumulate(dev, flag, stats, 8, overwrite=False)
        cmd = enc.finish()
    for _ in range(3):
        dev.queue.submit([cmd])
    dev.synchronize()
    reps, t0 = 50, time.perf_counter()
    for _ in range(reps):
        dev.queue.submit([cmd])
    dev.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"graph of {n:4d} dependent 1-wave kernels: {dt * 1e6:9.1f} us per replay, {dt * 1e6 / n:7.2f} us per kernel", flush=True)
    cmd.destroy()
