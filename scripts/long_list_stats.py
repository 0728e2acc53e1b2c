#!/usr/bin/env python3
"""How long is a PIXEL's part of a long tile list?  (dev tool)   python scripts/long_list_stats.py [config] [iterations] [tiles]

Trains [config] for [iterations] of the default schedule (default c3, 6000: the late regime of profiles/r06z_*), renders one view and evaluates, on the
host from the pass's own buffers, the longest tile lists: per pixel of the tile how many records lie inside their extent box (what a per-pixel forward
walk visits), how many of those come before the pixel saturates, and how many contribute to the backward pass (alpha >= 1/255, position < n_contrib).
The wave-per-block kernels walk every record of the list; the numbers say what a per-pixel walk would."""
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
top = int(sys.argv[3]) if len(sys.argv) > 3 else 3
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
t.flushPointCloud()
fw, rs = t.forwardPass, t.rasterizer
t.cameraBuffer.write(cams[0]); fw.encode(None); rs.encode(None, cfg.width, cfg.height); dev.synchronize()
res = fw.getResources()
E = int(fw.check()[0])
T = res["totalTiles"]
ranges = rs.getTileOffsetsBuffer().read(np.uint32, T + 1)
vals = res["tileIndicesBuffer"].read(np.uint32, E)
splats = res["splatBuffer"].read(np.uint32).reshape(-1, 6)
ncontrib = rs.getNContribTextureView().read(np.uint32).reshape(cfg.height, cfg.width)
W, H = cfg.width, cfg.height
starts = ranges[:T].astype(np.int64)
order = np.flatnonzero(starts != 0xFFFFFFFF)
ends = np.empty_like(starts)
ends[order] = np.append(starts[order][1:], E)
lens = np.where(starts != 0xFFFFFFFF, ends - starts, 0)
print(f"{name} after {t.getIteration()} iterations: N={t.getPointCount()} E={E} tiles={T}; list lengths: max={lens.max()} p99={int(np.percentile(lens, 99))} "
      f"p50={int(np.percentile(lens, 50))}; tiles above 1024 / 2048 / 4096 entries: {(lens > 1024).sum()} / {(lens > 2048).sum()} / {(lens > 4096).sum()}")


def halves(w):
    return (w & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float32), (w >> 16).astype(np.uint16).view(np.float16).astype(np.float32)


for tile in np.argsort(-lens)[:top]:
    n = int(lens[tile])
    if n == 0:
        break
    sp = splats[vals[starts[tile]:starts[tile] + n]]
    nx, ny = halves(sp[:, 0]); ex, ey = halves(sp[:, 1]); c_x, c_y = halves(sp[:, 2]); c_z, _ = halves(sp[:, 3]); _, op = halves(sp[:, 5])
    cx = (nx * 0.5 + 0.5) * W; cy = (ny * -0.5 + 0.5) * H
    ex = np.minimum(ex, 128.0); ey = np.minimum(ey, 128.0)
    tx, ty = int(tile % cfg.tiles_x), int(tile // cfg.tiles_x)
    px = tx * 16 + np.arange(16) + 0.5; py = ty * 16 + np.arange(16) + 0.5
    dx = px[None, None, :] - cx[:, None, None]; dy = py[None, :, None] - cy[:, None, None]      # [record, y, x]
    inside = (np.abs(dx) <= ex[:, None, None]) & (np.abs(dy) <= ey[:, None, None])
    power = -0.5 * (c_x[:, None, None] * dx * dx + c_z[:, None, None] * dy * dy) - c_y[:, None, None] * dx * dy
    alpha = np.minimum(0.99, op[:, None, None] * np.exp(np.minimum(power, 80.0)))
    w = np.where(inside, alpha, 0.0)
    A = np.zeros((16, 16), np.float32)
    visited = np.zeros((16, 16), np.int64)      # inside records met before the pixel saturates
    for i in range(n):
        live = ~(A > 0.99)
        visited += (inside[i] & live)
        A = np.where(live, A + w[i] * (1.0 - A), A)
    yy, xx = np.mgrid[0:16, 0:16]
    inb = ((ty * 16 + yy) < H) & ((tx * 16 + xx) < W)
    nc = np.zeros((16, 16), np.int64)
    nc[inb] = ncontrib[(ty * 16 + yy)[inb], (tx * 16 + xx)[inb]]
    pos = np.arange(n)[:, None, None]
    bwd = (inside & (alpha >= 1.0 / 255.0) & (pos < nc[None])).sum(0)
    ins = inside.sum(0)
    print(f"tile {tile} ({tx},{ty}): {n} entries; extents px: median {np.median(ex):.1f} x {np.median(ey):.1f}, opacity median {np.median(op):.3f}")
    for label, a in (("inside extent box (all)", ins), ("inside, before saturation (forward walk)", visited), ("backward-active (alpha>=1/255, pos<n_contrib)", bwd), ("n_contrib", nc)):
        a = a[inb]
        print(f"    per pixel {label:48s} max={a.max():6d} mean={a.mean():8.1f} p50={int(np.percentile(a, 50)):6d} p90={int(np.percentile(a, 90)):6d}")
    for b in range(4):
        sl = (slice((b >> 1) * 8, (b >> 1) * 8 + 8), slice((b & 1) * 8, (b & 1) * 8 + 8))
        blk_any = inside[:, sl[0], sl[1]].reshape(n, -1).any(1).sum()
        print(f"    block {b}: records with any pixel inside = {blk_any}; longest per-pixel forward walk {visited[sl].max()}, backward {bwd[sl].max()}; saturated pixels {(A[sl] > 0.99).sum()}/64")
