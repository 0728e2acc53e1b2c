#!/usr/bin/env python3
"""Dev probe: does the step time drift over a long run, and is it the scene (E grows as training moves the Gaussians) or the clock?
    python scripts/step_time_drift.py [config] [steps] [window]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webdgs_amd import ops, synth  # noqa: E402
from webdgs_amd.trainer import Trainer  # noqa: E402
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
window = int(sys.argv[3]) if len(sys.argv) > 3 else 100
cfg = synth.CONFIGS[name]
dev = ops.HipDevice(0)
g, sh = synth.make_gaussians(cfg)
tg, tsh = synth.make_target_scene(g, sh)
cams = synth.circle_cameras(cfg, 8)
cameras, images = bench.make_dataset(dev, cfg, tg, tsh, cams)
for frozen in (False, True):
    t = Trainer(dev, seed=99, pipeline_depth=2)
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.setMaxIterations(10 ** 9)
    if frozen:  # all learning rates zero: the same kernels on a scene that never changes
        t.setOptimizerHyperparameters({k: 0.0 for k in t.getOptimizerHyperparameters() if k.startswith("lr_")})
    t.start()
    for _ in range(3):
        t.step()
    t.warmupCommandBuffers()
    print(f"{name}, {'frozen scene (lr = 0)' if frozen else 'training'}:", flush=True)
    for w in range(steps // window):
        t.drain(); dev.synchronize()
        t0 = time.perf_counter()
        for _ in range(window):
            t.step()
        t.drain(); dev.synchronize()
        dt = (time.perf_counter() - t0) / window
        st = t.forwardPass.check()
        pairs = int(t.rasterizer.getNContribTextureView().read(np.uint32).astype(np.uint64).sum())
        print(f"  steps {w * window:5d}..{(w + 1) * window:5d}: {dt * 1e3:7.4f} ms/step   E {int(st[0]):9d}  V {int(st[1]):8d}  sum n_contrib {pairs:11d}", flush=True)
    t.destroy()
