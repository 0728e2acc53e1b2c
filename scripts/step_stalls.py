#!/usr/bin/env python3
"""Dev probe: every step of a c3 run at the reference's default schedule that takes longer than 3 ms, with what it spent its time in.
    python scripts/step_stalls.py [iterations]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from webdgs_amd import ops, synth  # noqa: E402
from webdgs_amd.trainer import Trainer  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
cfg = synth.CONFIGS["c3"]
dev = ops.HipDevice(0)
g, sh = synth.make_gaussians(cfg)
tg, tsh = synth.make_target_scene(g, sh)
cameras, images = bench.make_dataset(dev, cfg, tg, tsh, synth.circle_cameras(cfg, 8))
t = Trainer(dev, seed=99, pipeline_depth=2)
t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg)); t.setDataset(cameras, images); t.setMaxIterations(10 ** 9); t.start()
for _ in range(3):
    t.step()
t.warmupCommandBuffers()
dev.synchronize()
spent = {}


def wrap(obj, attr, key=None):
    fn = getattr(obj, attr)

    def timed(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        spent[key or attr] = spent.get(key or attr, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, attr, timed)


def wrap_class(cls, attr):
    fn = getattr(cls, attr)

    def timed(self, *a, **k):
        t0 = time.perf_counter()
        r = fn(self, *a, **k)
        key = cls.__name__ + "." + attr
        spent[key] = spent.get(key, 0.0) + time.perf_counter() - t0
        return r
    setattr(cls, attr, timed)


for a in ("_grow_long_lists", "runDensifyPruneMultiView", "applyPointCloudSwap", "drain", "_invalidate_command_buffers", "_wait", "_synchronize", "_run"):
    wrap(t, a)
for a in ("allocatePointCloudLike", "allocateOptimizerStateBuffers", "downsampleRGBA8"):
    wrap(ops, a)
for cls, names in ((ops.DensifyPrunePass, ("ensureSize", "encodePrepare", "readTotal", "encodeScatter")), (ops.TiledForwardPass, ("encode", "__init__", "longListStats", "setLongLists")),
                   (ops.TiledRasterizer, ("encode",)), (ops.TiledBackwardPass, ("computeMetricMap", "computeMetricCounts", "__init__")), (ops.HipDevice, ("createBuffer", "laneOrder")),
                   (ops.HipBuffer, ("write",))):
    for n in names:
        wrap_class(cls, n)
events, stalls, t_all = 0, 0, time.perf_counter()
while t.getIteration() < iters:
    before = t.getLastDensifyPruneIteration()
    spent.clear()
    t0 = time.perf_counter()
    t.step()
    dt = time.perf_counter() - t0
    ev = t.getLastDensifyPruneIteration() != before
    events += ev
    if dt > (8e-3 if ev else 3e-3):
        stalls += 1
        top = sorted(spent.items(), key=lambda kv: -kv[1])[:6]
        print(f"iteration {t.getIteration()} ({'event' if ev else 'step'}, {t.getPointCount()} points): {dt * 1e3:.1f} ms  " + "  ".join(f"{k} {v * 1e3:.1f}" for k, v in top), flush=True)
t.drain(); dev.synchronize()
print(f"{iters} iterations in {time.perf_counter() - t_all:.2f} s, {events} events, {stalls} long steps", flush=True)
