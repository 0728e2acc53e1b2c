import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from webdgs_amd import ops
from webdgs_amd.trainer import Trainer
import harness
from test_gpu_lifecycle import _dataset
dev = ops.HipDevice(0)
cfg = harness.small_config("c2", num_points=20000, width=320, height=240, s0=0.006)
g, sh, _ = harness.scene(cfg)
cameras, images = _dataset(dev, cfg, g, sh, 4)
dens = dict(schedule=dict(enabled=True, warmupIterations=10, interval=10, stopIterations=100), metricViews=3, cloneThresholdCount=5, splitScaleThreshold=0.03, pruneOpacity=0.2, maxNewPointsPerStep=500)
def mem(tag):
    dev.synchronize(); i = dev.memoryInfo(); print(f"{tag:40s} free {i['free'] / 2**20:10.1f} MiB cached {i['cached'] / 2**20:8.1f} MiB", flush=True)
mem("start")
for cycle in range(3):
    t = Trainer(dev, seed=cycle, views_per_rank=1, pipeline_depth=2)
    if cycle == 2:
        t.longLists = dict(threshold=40, maxItems=8192, maxRows=65536)
    t.setDensifyPruneConfig(dens)
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg)); t.setDataset(cameras, images); t.start()
    mem(f"cycle {cycle}: trainer built")
    for i in range(25):
        t.step()
        if i in (0, 9, 10, 24):
            t.drain(); print("   step", i + 1, t.forwardPass.longListStats(), flush=True)
    t.drain(); mem(f"cycle {cycle}: trained")
    cloud = t.pointCloud
    t.destroy(); cloud.gaussian_3d_buffer.destroy(); cloud.sh_buffer.destroy()
    mem(f"cycle {cycle}: destroyed")
