#!/usr/bin/env python3
"""What a long tile list costs (dev tool): the synthetic scenes of tests/test_gpu_nan.py::test_tile_lists_past_the_reference_cap -- one 16 x 16 tile with
`big` entries -- timed kernel by kernel over eager training steps.   python scripts/long_list_scenes.py [big ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from webdgs_amd import ops  # noqa: E402
import harness  # noqa: E402
from test_gpu_nan import long_list_scene  # noqa: E402

dev = ops.HipDevice(0)
for big in [int(a) for a in sys.argv[1:]] or [10_400, 40_000]:
    for kind in ("sparse", "faint", "pile-up"):
        cfg, g, sh, cam, rng = long_list_scene(kind, big=big)
        target = dev.bufferFrom(rng.integers(0, 255, (cfg.height, cfg.width, 4), dtype=np.uint8))
        pipe = harness.HipPipeline(dev, cfg, g, sh, cam)
        pipe.train_step(target); dev.synchronize()
        dev.setProfiling(True); dev.kernelTimes(reset=True)
        steps = 5
        for _ in range(steps):
            pipe.train_step(target)
        dev.synchronize(); dev.setProfiling(False)
        got = pipe.collect_forward()
        t = {k: ms / steps for k, (n, ms) in dev.kernelTimes().items()}
        print(f"{kind:8s} longest list {big:6d}: E={got['total_entries']} max n_contrib={int(got['n_contrib'].max())}  rasterize {t.get('rasterize', 0) * 1e3:8.1f} us  "
              f"backward_rasterize {t.get('backward_rasterize', 0) * 1e3:8.1f} us  sort_segments {t.get('sort_segments', 0) * 1e3:7.1f} us  step (kernel sum) {sum(t.values()) * 1e3:8.1f} us", flush=True)
        pipe.destroy(); target.destroy()
