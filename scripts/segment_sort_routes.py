#!/usr/bin/env python3
"""Which route the per-tile depth sort (sort.hip: segment_sort) takes per tile at a config: one 10-bit pass (depth span of the tile's entries
below 2^10), two 8-bit passes (wider span), or the global-memory form (more than SEG_CAP = 2048 entries) -- counted on the host from the
forward pass's own sorted keys and range table.   python scripts/segment_sort_routes.py [config]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webdgs_amd import ops, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
cfg = synth.CONFIGS[name]
dev = ops.HipDevice(0)
g, sh = synth.make_gaussians(cfg)
cam = dev.bufferFrom(synth.circle_cameras(cfg, 8)[1])
pc = ops.createPointCloud(dev, g, sh, cfg.sh_deg)
fw = ops.TiledForwardPass(dev, pc, cam, dict(viewportWidth=cfg.width, viewportHeight=cfg.height))
rs = ops.TiledRasterizer(dict(device=dev, forwardPass=fw))
fw.encode(None)
rs.encode(None, cfg.width, cfg.height)
e = int(fw.check()[0])
keys = fw.getSortedKeysBuffer().read(np.uint32, count=e)
tile = (keys >> 16).astype(np.int64) - 1
depth = (keys & 0xFFFF).astype(np.int64)
tiles = cfg.total_tiles
counts = np.bincount(tile, minlength=tiles)
lo = np.full(tiles, 1 << 20); hi = np.full(tiles, -1)
np.minimum.at(lo, tile, depth); np.maximum.at(hi, tile, depth)
span = np.where(counts > 0, hi - lo, 0)
nonempty = counts > 1
one = nonempty & (counts <= 2048) & (span < 1024)
two = nonempty & (counts <= 2048) & (span >= 1024)
big = counts > 2048
print(f"{name}: E = {e}, tiles = {tiles}, non-trivial segments (more than one entry) = {int(nonempty.sum())}")
for label, m in (("one 10-bit pass in LDS", one), ("two 8-bit passes in LDS", two), ("two passes through global memory (> 2048 entries)", big)):
    print(f"  {label:52s} {int(m.sum()):7d} tiles  {100.0 * m.sum() / max(1, nonempty.sum()):6.2f} %   {int(counts[m].sum()):10d} entries  {100.0 * counts[m].sum() / max(1, e):6.2f} %")
print(f"  entries per non-empty tile: mean {counts[counts > 0].mean():.0f}, median {np.median(counts[counts > 0]):.0f}, max {counts.max()};  depth span: median {np.median(span[nonempty]):.0f}, 99th percentile "
      f"{np.percentile(span[nonempty], 99):.0f}, max {span.max()} (of 65535)")
rs.destroy(); fw.destroy()
