#!/usr/bin/env python3
"""End-to-end run of the Trainer on a synthetic scene: training steps with the reference's densify schedule, PSNR against the
ground-truth views and the point count over time.

    python scripts/train_demo.py [config] [iterations] [views] [lr_scale] [report_every] [frozen_rates]   (default: c3 1200 8 1.0 100 ''; needs an MI355X)

lr_scale multiplies the five Adam learning rates: 0.316 (= sqrt(1 - beta2) / (1 - beta1)) makes the first steps as long as
bias-corrected Adam would, which is how the start-up PSNR dip of the reference's uncorrected Adam is told from a defect.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from webdgs_amd import ops, synth  # noqa: E402
from webdgs_amd.trainer import Trainer  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
    views = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    lr_scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    every = int(sys.argv[5]) if len(sys.argv) > 5 else 100
    frozen = [k for k in (sys.argv[6].split(",") if len(sys.argv) > 6 else []) if k]   # e.g. lr_pos,lr_rot,lr_scale: those rates become 0
    cfg = synth.CONFIGS[name]
    dev = ops.HipDevice(0)
    g, sh = synth.make_gaussians(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    cams = synth.circle_cameras(cfg, views)
    # ground truth: the perturbed scene rendered by the same forward pass
    tpc = ops.createPointCloud(dev, tg, tsh, cfg.sh_deg)
    tcam = dev.createBuffer(272)
    tfw = ops.TiledForwardPass(dev, tpc, tcam, dict(viewportWidth=cfg.width, viewportHeight=cfg.height, renderMode="gaussian"))
    trs = ops.TiledRasterizer(dict(device=dev, forwardPass=tfw, format="rgba8unorm"))
    cameras, images = [], []
    for i in range(views):
        tcam.write(cams[i])
        tfw.encode(None)
        trs.encode(None, cfg.width, cfg.height)
        dev.synchronize()
        images.append(dict(texture=dev.bufferFrom(trs.getOutputTextureView().read(np.uint8)), width=cfg.width, height=cfg.height))
        cameras.append(dict(camera=cams[i], width=cfg.width, height=cfg.height))
    trs.destroy()
    tfw.destroy()

    t = Trainer(dev, seed=3)
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=True, warmupIterations=500, interval=100, stopIterations=15_000), maxBufferBytes=0))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.setMaxIterations(10 ** 9)
    if lr_scale != 1.0 or frozen:
        hp = t.getOptimizerHyperparameters()
        t.setOptimizerHyperparameters({k: (0.0 if k in frozen else v * lr_scale) for k, v in hp.items() if k.startswith("lr_")})
    t.start()

    def psnr():
        vals = []
        for i in range(views):
            t.forwardPass.setCameraBuffer(t._camera_buffers[i])
            t.forwardPass.encode(None)
            t.rasterizer.encode(None, cfg.width, cfg.height)
            vals.append(ops.imagePSNR(dev, t.rasterizer.getOutputTextureView(), images[i]["texture"], cfg.width * cfg.height))
        return float(np.mean(vals))

    t.step()  # builds the pipelines
    print(f"{name}: {cfg.num_points} Gaussians, {cfg.width}x{cfg.height}, {views} views, lr x {lr_scale}, frozen {frozen}", flush=True)
    print(f"iter {t.getIteration():5d}  points {t.getPointCount():8d}  PSNR {psnr():6.2f} dB", flush=True)
    t0, last = time.perf_counter(), 1
    while t.getIteration() < iters:
        t.step()
        it = t.getIteration()
        if it % every == 0 or it == iters:
            dev.synchronize()
            dt = time.perf_counter() - t0
            st = t.forwardPass.check()
            print(f"iter {it:5d}  points {t.getPointCount():8d}  PSNR {psnr():6.2f} dB  E {int(st[0]):9d}  {(it - last) / dt:7.1f} it/s"
                  f"  last densify {t.getLastDensifyPruneIteration()}", flush=True)
            t0, last = time.perf_counter(), it


if __name__ == "__main__":
    main()
