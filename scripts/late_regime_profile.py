#!/usr/bin/env python3
"""Where a step's time goes late in a run of the default schedule (dev tool): trains [config] for [iterations] (default c3, 6000), then times ten eager steps
kernel by kernel.   python scripts/late_regime_profile.py [config] [iterations]"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webdgs_amd import ops, synth  # noqa: E402
from webdgs_amd.trainer import Trainer  # noqa: E402

warnings.simplefilter("always")
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
cfg = synth.CONFIGS[name]
dev = ops.HipDevice(0)
g, sh = synth.make_gaussians(cfg)
tg, tsh = synth.make_target_scene(g, sh)
cams = synth.circle_cameras(cfg, 8)
tpc = ops.createPointCloud(dev, tg, tsh, cfg.sh_deg)
tcam = dev.createBuffer(272)
tfw = ops.TiledForwardPass(dev, tpc, tcam, dict(viewportWidth=cfg.width, viewportHeight=cfg.height))
trs = ops.TiledRasterizer(dict(device=dev, forwardPass=tfw))
images, cameras = [], []
for c in cams:
    tcam.write(c); tfw.encode(None); trs.encode(None, cfg.width, cfg.height); dev.synchronize()
    images.append(dict(texture=dev.bufferFrom(trs.getOutputTextureView().read(np.uint8)), width=cfg.width, height=cfg.height))
    cameras.append(dict(camera=c, width=cfg.width, height=cfg.height))
trs.destroy(); tfw.destroy()
t = Trainer(dev, seed=3, pipeline_depth=2)
t.setDensifyPruneConfig(dict(schedule=dict(enabled=True, warmupIterations=500, interval=100, stopIterations=15_000), maxBufferBytes=0))
t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg)); t.setDataset(cameras, images); t.setMaxIterations(10 ** 9); t.start()
while t.getIteration() < iters:
    t.step()
t.drain(); dev.synchronize()
t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
t.use_command_buffers = False
t._invalidate_command_buffers()
t.step(); dev.synchronize()
dev.setProfiling(True); dev.kernelTimes(reset=True)
steps = 10
for _ in range(steps):
    t.step()
t.drain(); dev.synchronize()
dev.setProfiling(False)
st = t.forwardPass.check()
print(f"{name} after {t.getIteration()} iterations: N={t.getPointCount()} E={int(st[0])} V={int(st[1])} tiles={cfg.tiles_x * cfg.tiles_y} E/tile={int(st[0]) / (cfg.tiles_x * cfg.tiles_y):.0f} E/V={int(st[0]) / max(1, int(st[1])):.1f}")
tot = 0.0
for k, (n, ms) in sorted(dev.kernelTimes().items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:24s} launches/step={n / steps:5.1f}  ms/step={ms / steps:8.4f}")
    tot += ms / steps
print(f"  kernel sum ms/step = {tot:.4f}")
print("  long-list work of the last frame:", t.forwardPass.longListStats())
