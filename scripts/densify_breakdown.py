#!/usr/bin/env python3
"""Dev probe: where a densify event's wall time goes at a config (default c3).   python scripts/densify_breakdown.py [config] [events]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webdgs_amd import ops, synth  # noqa: E402
from webdgs_amd.trainer import Trainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
events = int(sys.argv[2]) if len(sys.argv) > 2 else 3
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
for i in range(8):
    tcam.write(cams[i]); tfw.encode(None); trs.encode(None, cfg.width, cfg.height); dev.synchronize()
    images.append(dict(texture=dev.bufferFrom(trs.getOutputTextureView().read(np.uint8)), width=cfg.width, height=cfg.height))
    cameras.append(dict(camera=cams[i], width=cfg.width, height=cfg.height))
trs.destroy(); tfw.destroy()
t = Trainer(dev, seed=1)
t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg)); t.setDataset(cameras, images); t.start()


def timed(label, fn):
    dev.synchronize()
    t0 = time.perf_counter()
    r = fn()
    dev.synchronize()
    print(f"    {label:46s} {(time.perf_counter() - t0) * 1e3:8.2f} ms", flush=True)
    return r


for _ in range(20):
    t.step()
for e in range(events):
    print(f"event {e}: {t.getPointCount()} points", flush=True)
    orig_sync = t.syncOptimizerState
    if e == events - 1:  # per-kernel view of the last event (eager launches with events around them)
        dev.setProfiling(True)
        dev.kernelTimes(reset=True)
    timed("metric views + prepare + scatter (runDensifyPruneMultiView)", t.runDensifyPruneMultiView)
    if e == events - 1:
        dev.setProfiling(False)
        tot = 0.0
        for k, (n, ms) in sorted(dev.kernelTimes().items(), key=lambda kv: -kv[1][1]):
            print(f"        {k:28s} launches {n:4d}  total {ms:8.3f} ms", flush=True)
            tot += ms
        print(f"        kernel sum {tot:.3f} ms", flush=True)
    req = t.consumePointCloudSwapRequest()
    if req is not None:
        timed("applyPointCloudSwap (destroy ops, new optimizer, new ops)", lambda: t.applyPointCloudSwap(req))
    timed("first step after (eager, first-use allocations)", t.step)
    timed("next 8 steps (recording the views' command buffers)", lambda: [t.step([v]) for v in range(8)])
    timed("8 replayed steps", lambda: [t.step([v]) for v in range(8)])
    for _ in range(30):
        t.step()
