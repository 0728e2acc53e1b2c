#!/usr/bin/env python3
"""Per-kernel hipEvent timings of eager training steps (dev tool): python scripts/profile_step.py [config] [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webdgs_amd import ops, synth  # noqa: E402
from webdgs_amd.trainer import Trainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = synth.CONFIGS[name]
dev = ops.HipDevice(0)
g, sh = synth.make_gaussians(cfg)
tg, tsh = synth.make_target_scene(g, sh)
cams = synth.circle_cameras(cfg, 4)
tpc = ops.createPointCloud(dev, tg, tsh, cfg.sh_deg)
tcam = dev.createBuffer(272)
tfw = ops.TiledForwardPass(dev, tpc, tcam, dict(viewportWidth=cfg.width, viewportHeight=cfg.height))
trs = ops.TiledRasterizer(dict(device=dev, forwardPass=tfw))
images, cameras = [], []
for i in range(4):
    tcam.write(cams[i]); tfw.encode(None); trs.encode(None, cfg.width, cfg.height); dev.synchronize()
    images.append(dict(texture=dev.bufferFrom(trs.getOutputTextureView().read(np.uint8)), width=cfg.width, height=cfg.height))
    cameras.append(dict(camera=cams[i], width=cfg.width, height=cfg.height))
trs.destroy(); tfw.destroy()
vpr = int(os.environ.get("WDGS_PROFILE_VPR", "1"))  # views per step: > 1 profiles the batched step (view-batched K1 / K17, the fp32 Adam pass)
t = Trainer(dev, seed=1, use_command_buffers=False, views_per_rank=vpr)
if vpr > 1:
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg)); t.setDataset(cameras, images)
if os.environ.get("WDGS_PROFILE_FROZEN"):  # every learning rate 0: the same scene in every step (and in every build that is compared)
    t.setOptimizerHyperparameters({k: 0.0 for k in t.getOptimizerHyperparameters() if k.startswith("lr_")})
t.start()
for _ in range(3):
    t.step()
dev.setProfiling(True); dev.kernelTimes(reset=True)
for _ in range(steps):
    t.step()
dev.setProfiling(False)
st = t.forwardPass.check()
print(f"{name}: N={cfg.num_points} E={st[0]} V={st[1]} step_ms={t.getLastStepMs():.3f}")
tot = 0.0
for k, (n, ms) in sorted(dev.kernelTimes().items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:24s} launches/step={n / steps:5.1f}  ms/step={ms / steps:8.4f}  avg_us={ms / n * 1e3:9.2f}")
    tot += ms / steps
print(f"  kernel sum ms/step = {tot:.4f}")
