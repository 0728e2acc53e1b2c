#!/usr/bin/env python3
"""Wall time of the steps that carry a densify event, Python host (dev tool; the JS twin is bindings/napi/densify_timing.js):
python scripts/densify_event_steps.py [config]   -- densify every 20 iterations, pipeline depth 2, 70 steps."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webdgs_amd import ops, synth  # noqa: E402
from webdgs_amd.trainer import Trainer  # noqa: E402

cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
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
t = Trainer(dev, seed=1, pipeline_depth=2)
t.setDensifyPruneConfig(dict(schedule=dict(enabled=True, warmupIterations=20, interval=20, stopIterations=1000)))
t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg)); t.setDataset(cameras, images); t.setMaxIterations(10 ** 9); t.start()
acc = {}


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g_(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        dev.synchronize()
        acc.setdefault(label, []).append((time.perf_counter() - t0) * 1e3)
        return r
    setattr(obj, name, g_)


wrap(t, "runDensifyPruneMultiView", "runDensifyPruneMultiView"); wrap(t, "applyPointCloudSwap", "applyPointCloudSwap")
wrap(t, "_invalidate_command_buffers", "invalidateCommandBuffers"); wrap(t, "ensurePipelines", "ensurePipelines")
_cbd = ops.HipCommandBuffer.destroy


def _timed_destroy(self):
    t0 = time.perf_counter(); _cbd(self); acc.setdefault("HipCommandBuffer.destroy (bare)", []).append((time.perf_counter() - t0) * 1e3)


ops.HipCommandBuffer.destroy = _timed_destroy
slow = []
for i in range(70):
    t0 = time.perf_counter(); t.step(); dt = (time.perf_counter() - t0) * 1e3
    if dt > 1.5:
        slow.append(f"{t.getIteration()}: {dt:.2f}")
t.drain(); dev.synchronize()
for k, v in acc.items():
    print(k, f"n={len(v)}", "last few:", " ".join(f"{x:.2f}" for x in v[-6:]))
print("steps around the events (iteration: ms):", "  ".join(slow))
