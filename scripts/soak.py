#!/usr/bin/env python3
"""Soak run of the Trainer (dev tool): python scripts/soak.py [config] [steps] [views_per_step] [exchange: none|capi]
Default densify schedule, pipeline depth 2, 3 lanes; prints the point count, step rate and free device memory every 250 steps; any warning is shown."""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webdgs_amd import ops, parallel, synth  # noqa: E402
from webdgs_amd.trainer import Trainer  # noqa: E402

warnings.simplefilter("always")
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
vpr = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ex = sys.argv[4] if len(sys.argv) > 4 else "none"
cfg = synth.CONFIGS[name]
dev = ops.HipDevice(0)
g, sh = synth.make_gaussians(cfg)
tg, tsh = synth.make_target_scene(g, sh)
cams = synth.circle_cameras(cfg, 16)
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
exchange = parallel.CapiExchange(dev) if ex == "capi" else None
t = Trainer(dev, seed=5, views_per_rank=vpr, overlap_views=3 if vpr > 1 else None, pipeline_depth=2, exchange=exchange)
t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg)); t.setDataset(cameras, images); t.setMaxIterations(10 ** 9); t.start()
import torch  # noqa: E402  (free-memory readout only)
t0, last = time.perf_counter(), 0
while t.getIteration() < steps:
    t.step()
    it = t.getIteration()
    if it % 250 == 0 and it != last:
        dev.synchronize()
        free_b, total_b = torch.cuda.mem_get_info(0)
        print(f"iter {it:6d}  points {t.getPointCount():8d}  {(it - last) * vpr / (time.perf_counter() - t0):8.1f} views/s  free {free_b / 2 ** 30:7.1f} GiB  last densify {t.getLastDensifyPruneIteration()}", flush=True)
        t0, last = time.perf_counter(), it
t.drain(); dev.synchronize()
t.destroy()
if exchange is not None:
    exchange.destroy()
print("SOAK_OK")
