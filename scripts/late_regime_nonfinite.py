#!/usr/bin/env python3
"""Which fp16 fields of the VISIBLE Splats are non-finite late in a run of the default schedule, and how many tiles those Splats cover (dev tool).
    python scripts/late_regime_nonfinite.py [config] [iterations]"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webdgs_amd import ops, synth  # noqa: E402
from webdgs_amd.trainer import Trainer  # noqa: E402

warnings.simplefilter("ignore")
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
t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))   # (iteration `iters` may be a densify event: the rebuilt passes have not rendered anything yet)
t.step(); t.drain(); dev.synchronize()
fw = t.forwardPass
n = t.getPointCount()
res = fw.getResources()
splats = res["splatBuffer"].read(np.uint16).reshape(-1, 12)[:n]
counts = res["tileCountsBuffer"].read(np.uint32)[:n] if "tileCountsBuffer" in res else None
if counts is None:
    counts = fw.getTileCountsBuffer().read(np.uint32)[:n]
vis = counts > 0
names = ["ndc.x", "ndc.y", "extent.x", "extent.y", "conic.x", "conic.y", "conic.z", "(w23.y hi)", "colour.r", "colour.g", "colour.b", "opacity"]
h = splats[vis]
expo = (h & 0x7C00) == 0x7C00
nan = expo & ((h & 0x3FF) != 0)
inf = expo & ~nan
print(f"{name} after {t.getIteration()} iterations: N={n}, visible={int(vis.sum())}, tiles={cfg.tiles_x * cfg.tiles_y}")
for i, nm in enumerate(names):
    print(f"  {nm:12s} NaN in {int(nan[:, i].sum()):6d}   infinity in {int(inf[:, i].sum()):6d}")
any_nf = expo.any(axis=1)
only_inf_extent = any_nf & ~nan.any(axis=1) & ~inf[:, [0, 1, 4, 5, 6, 8, 9, 10, 11]].any(axis=1)
c = counts[vis]
print(f"  Splats with a non-finite half: {int(any_nf.sum())}; of them with NOTHING but an infinite extent: {int(only_inf_extent.sum())}")
print(f"  tiles covered by the non-finite ones: total {int(c[any_nf].sum())}, largest box {int(c[any_nf].max()) if any_nf.any() else 0} tiles; by those with only an infinite extent: {int(c[only_inf_extent].sum())}")
