"""Scene shared by tests/test_gpu_dp.py and its worker processes."""
import numpy as np

from webdgs_amd import synth

import harness


def dataset(dev):
    import os
    big = os.environ.get("WDGS_DP_TEST_BIG")  # manual stress: WDGS_DP_TEST_BIG=400000
    few = os.environ.get("WDGS_DP_TEST_POINTS")  # a cloud smaller than one slice: the last rank owns nothing
    cfg = harness.small_config("c2", num_points=int(big), width=640, height=480) if big else harness.small_config("c2", num_points=int(few) if few else 6000, width=128, height=96,
                                                                                                                   s0=0.03 if few else 0.01)
    g, sh, _ = harness.scene(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    cams = synth.circle_cameras(cfg, 4)
    cameras, images = [], []
    for i in range(4):
        p = harness.HipPipeline(dev, cfg, tg, tsh, cams[i])
        p.forward()
        images.append(dict(texture=dev.bufferFrom(p.rast.getOutputTextureView().read(np.uint8)), width=cfg.width, height=cfg.height))
        cameras.append(dict(camera=cams[i], width=cfg.width, height=cfg.height))
        p.destroy()
    return cfg, g, sh, cameras, images


def view_schedule(steps, world, views_per_rank=1):
    """Global view ids per step (world * views_per_rank of them): rank r takes ids[r::world]; a single process takes all."""
    return [[(3 * s + 2 * r + 1) % 4 for r in range(world * views_per_rank)] for s in range(steps)]
