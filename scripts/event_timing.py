#!/usr/bin/env python3
"""Dev probe: wall time of the pieces of Trainer.step() at densify events, as bench.py's sustained leg sees them.   python scripts/event_timing.py [config]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from webdgs_amd import ops, synth  # noqa: E402
from webdgs_amd.trainer import Trainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
cfg = synth.CONFIGS[name]
dev = ops.HipDevice(0)
g, sh = synth.make_gaussians(cfg)
tg, tsh = synth.make_target_scene(g, sh)
cameras, images = bench.make_dataset(dev, cfg, tg, tsh, synth.circle_cameras(cfg, 8))
t = Trainer(dev, seed=99, pipeline_depth=2)
t.setDensifyPruneConfig(dict(schedule=dict(enabled=True, warmupIterations=30, interval=20, stopIterations=10 ** 6)))
t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg)); t.setDataset(cameras, images); t.setMaxIterations(10 ** 9); t.start()
for _ in range(3):
    t.step()
t.warmupCommandBuffers()
dev.synchronize()
spent = {}


def wrap(obj, attr):
    fn = getattr(obj, attr)

    def timed(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        spent[attr] = spent.get(attr, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, attr, timed)


for a in ("_grow_long_lists", "runDensifyPruneMultiView", "applyPointCloudSwap", "drain", "_invalidate_command_buffers", "ensureMetricsPipelines", "_metric_overflow", "_agree"):
    if hasattr(t, a):
        wrap(t, a)
wrap(t, "_metric_set"); wrap(t, "_synchronize"); wrap(t, "syncOptimizerState")
for a in ("allocatePointCloudLike", "allocateOptimizerStateBuffers", "downsampleRGBA8"):
    wrap(ops, a)
def wrap_class(cls, attr):
    fn = getattr(cls, attr)

    def timed(self, *a, **k):
        t0 = time.perf_counter()
        r = fn(self, *a, **k)
        key = cls.__name__ + "." + attr
        spent[key] = spent.get(key, 0.0) + time.perf_counter() - t0
        return r
    setattr(cls, attr, timed)


for cls, names in ((ops.DensifyPrunePass, ("ensureSize", "encodePrepare", "readTotal", "encodeScatter")), (ops.TiledForwardPass, ("encode", "__init__")),
                   (ops.TiledRasterizer, ("encode", "__init__")), (ops.TiledBackwardPass, ("computeMetricMap", "computeMetricCounts", "normalizeMetricCounts", "__init__"))):
    for n in names:
        wrap_class(cls, n)
for cls, names in ((ops.HipBuffer, ("write",)), (ops.HipEncoder, ("clearBuffer",)), (ops.HipDevice, ("createCommandEncoder", "selectLane", "laneOrder", "createBuffer")),
                   (ops.TiledBackwardPass, ("setMetricCountsTarget",))):
    for n in names:
        if hasattr(cls, n):
            wrap_class(cls, n)
while t.getIteration() < 100:
    before = t.getLastDensifyPruneIteration()
    spent.clear()
    t0 = time.perf_counter()
    t.step()
    dt = time.perf_counter() - t0
    if t.getLastDensifyPruneIteration() != before or dt > 2e-3:
        print(f"iteration {t.getIteration()}: step {dt * 1e3:.2f} ms  " + "  ".join(f"{k} {v * 1e3:.2f}" for k, v in spent.items()), flush=True)
t.drain(); dev.synchronize()
