#!/usr/bin/env python3
"""Dev aid: one forward + rasterize (+ optionally backward) of a synthetic long-list scene, then the long-list work's header and block records.
   WDGS_LL_DEBUG=1 python scripts/long_list_debug.py [kind] [big] [backward 0/1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from webdgs_amd import ops  # noqa: E402
import harness  # noqa: E402
from test_gpu_nan import long_list_scene  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "sparse"
big = int(sys.argv[2]) if len(sys.argv) > 2 else 10_400
backward = len(sys.argv) > 3 and sys.argv[3] == "1"
dev = ops.HipDevice(0)
cfg, g, sh, cam, rng = long_list_scene(kind, big=big)
pipe = harness.HipPipeline(dev, cfg, g, sh, cam)
print("encode forward", flush=True)
pipe.fwd.encode(None); dev.synchronize()
print("stats after the forward pass (sort built the work):", pipe.fwd.longListStats(), flush=True)
print("encode rasterizer", flush=True)
pipe.rast.encode(None, cfg.width, cfg.height)
dev.synchronize()
print("stats after the rasterizer:", pipe.fwd.longListStats(), flush=True)
if backward:
    target = dev.bufferFrom(rng.integers(0, 255, (cfg.height, cfg.width, 4), dtype=np.uint8))
    pipe.bwd.encode(None, pipe.rast.getOutputTextureView(), target, pipe.backward_resources()); dev.synchronize()
    print("stats after the backward pass:", pipe.fwd.longListStats(), flush=True)
pipe.destroy()
print("done", flush=True)
