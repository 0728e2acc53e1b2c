#!/usr/bin/env python3
"""End-to-end run of the Trainer on a synthetic scene: training steps with the reference's densify schedule, PSNR against the
ground-truth views and the point count over time.

    python scripts/train_demo.py [config] [iterations] [views]        (default: c3 1200 8; needs an MI355X)
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
    print(f"{name}: {cfg.num_points} Gaussians, {cfg.width}x{cfg.height}, {views} views", flush=True)
    print(f"iter {t.getIteration():5d}  points {t.getPointCount():8d}  PSNR {psnr():6.2f} dB", flush=True)
    t0, last = time.perf_counter(), 1
    while t.getIteration() < iters:
        t.step()
        it = t.getIteration()
        if it % 100 == 0 or it == iters:
            dev.synchronize()
            dt = time.perf_counter() - t0
            st = t.forwardPass.check()
            print(f"iter {it:5d}  points {t.getPointCount():8d}  PSNR {psnr():6.2f} dB  E {int(st[0]):9d}  {(it - last) / dt:7.1f} it/s"
                  f"  last densify {t.getLastDensifyPruneIteration()}", flush=True)
            t0, last = time.perf_counter(), it


if __name__ == "__main__":
    main()
