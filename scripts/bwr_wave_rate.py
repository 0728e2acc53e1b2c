#!/usr/bin/env python3
"""How fast does ONE backward_rasterize wave iterate, as a function of how many waves share its SIMD?  (dev tool, GPU)

    WDGS_BWR_TIMELINE=/tmp/tl.bin WDGS_BWR_ROLES=1 python scripts/bwr_wave_rate.py

Scenes of c2's density (100 000 Gaussians per 640 x 480) on viewports of 8 ... 2048 tiles: 32 waves (each alone on its SIMD) up to 8 per SIMD.
Per launch the kernel's TIMELINE form gives every wave's life and its iterations; printed per viewport: ns per iteration of a wave (median),
waves per SIMD, the SIMD's iterations per us.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bwr_timeline  # noqa: E402
from webdgs_amd import ops, synth  # noqa: E402
from webdgs_amd.trainer import Trainer  # noqa: E402

path = os.environ["WDGS_BWR_TIMELINE"]
dev = ops.HipDevice(0)
for tx, ty in ((4, 2), (16, 8), (16, 16), (32, 16), (32, 32), (64, 32), (64, 64)):
    n = int(round(100_000 * tx * ty / 1200))
    cfg = synth.SceneConfig(2, n, 16 * tx, 16 * ty, 1, 550.0, 0.003, f"{tx}x{ty}")
    g, sh = synth.make_gaussians(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    cams = synth.circle_cameras(cfg, 1)
    tpc = ops.createPointCloud(dev, tg, tsh, cfg.sh_deg)
    tcam = dev.createBuffer(272)
    tfw = ops.TiledForwardPass(dev, tpc, tcam, dict(viewportWidth=cfg.width, viewportHeight=cfg.height))
    trs = ops.TiledRasterizer(dict(device=dev, forwardPass=tfw))
    tcam.write(cams[0]); tfw.encode(None); trs.encode(None, cfg.width, cfg.height); dev.synchronize()
    image = dict(texture=dev.bufferFrom(trs.getOutputTextureView().read(np.uint8)), width=cfg.width, height=cfg.height)
    trs.destroy(); tfw.destroy()
    t = Trainer(dev, seed=1, use_command_buffers=False)
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg)); t.setDataset([dict(camera=cams[0], width=cfg.width, height=cfg.height)], [image])
    t.setOptimizerHyperparameters({k: 0.0 for k in t.getOptimizerHyperparameters() if k.startswith("lr_")})
    t.start()
    if os.path.exists(path):
        os.remove(path)
    for _ in range(3):
        t.step()
    dev.synchronize()
    slots, tiles, rec = list(bwr_timeline.launches(path))[-1]
    ran = rec[:, 1] != 0
    it = (rec[ran, 3] & np.uint64(0xFFFF)).astype(np.int64)
    life = (rec[ran, 1] - rec[ran, 0]).astype(np.int64)
    t0 = rec[ran, 0].astype(np.int64); t1 = rec[ran, 1].astype(np.int64)
    span = (t1.max() - t0.min()) * 0.01
    ids = rec[ran, 2]
    hw = (ids & np.uint64(0xFFFFFFFF)).astype(np.int64)
    simd_key = (((ids >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64) << 20) | (hw & 0xFF30)  # xcc, se, sh, cu, simd
    work = it >= 8
    per = life[work] * 10.0 / it[work]
    n_simd = len(np.unique(simd_key))
    print(f"{tx * ty:5d} tiles N={n:6d}: waves={ran.sum():5d} on {n_simd:4d} SIMDs ({ran.sum() / n_simd:.2f} per SIMD)  iterations/wave mean={it.mean():6.1f} max={it.max():4d}  "
          f"ns/iteration of a wave p10={np.percentile(per, 10):6.0f} p50={np.percentile(per, 50):6.0f} p90={np.percentile(per, 90):6.0f}  span={span:7.2f} us  "
          f"SIMD iterations/us={it.sum() / n_simd / span:5.2f}", flush=True)
    t.destroy() if hasattr(t, "destroy") else None
