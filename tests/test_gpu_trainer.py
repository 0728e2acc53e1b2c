"""GPU: the Trainer counterpart end to end -- recorded command buffers vs eager submission, densify/prune in the loop."""
import numpy as np
import pytest

from webdgs_amd import ops, synth
from webdgs_amd.trainer import Trainer

import harness
from harness import assert_bits_equal

pytestmark = pytest.mark.gpu


def _dataset(dev, cfg, g, sh, n_views):
    tg, tsh = synth.make_target_scene(g, sh)
    cams = synth.circle_cameras(cfg, n_views)
    cameras, images = [], []
    for i in range(n_views):
        p = harness.HipPipeline(dev, cfg, tg, tsh, cams[i])
        p.forward()
        images.append(dict(texture=dev.bufferFrom(p.rast.getOutputTextureView().read(np.uint8)), width=cfg.width, height=cfg.height))
        cameras.append(dict(camera=cams[i], width=cfg.width, height=cfg.height))
        p.destroy()
    return cameras, images


def _run(dev, cfg, g, sh, cameras, images, steps, use_cb, densify=None):
    t = Trainer(dev, seed=7, use_command_buffers=use_cb)
    if densify:
        t.setDensifyPruneConfig(densify)
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.start()
    for _ in range(steps):
        t.step()
    return t


def test_recorded_command_buffers_equal_eager_submission(hip_device):
    cfg = harness.small_config("c2", num_points=6000, width=160, height=112)
    g, sh, _ = harness.scene(cfg)
    cameras, images = _dataset(hip_device, cfg, g, sh, 3)
    a = _run(hip_device, cfg, g, sh, cameras, images, 12, True)
    b = _run(hip_device, cfg, g, sh, cameras, images, 12, False)
    assert len(a._cmd_cache) >= 2, "views were recorded"
    assert a.getIteration() == b.getIteration() == 12 and a.optimizer.getIteration() == b.optimizer.getIteration() == 12
    assert_bits_equal(a.pointCloud.gaussian_3d_buffer.read(np.uint32), b.pointCloud.gaussian_3d_buffer.read(np.uint32), "gaussians after 12 steps")
    assert_bits_equal(a.pointCloud.sh_buffer.read(np.uint32), b.pointCloud.sh_buffer.read(np.uint32), "sh after 12 steps")
    for k in a.optimizer.getStateBuffers():
        assert_bits_equal(a.optimizer.getStateBuffers()[k].read(np.uint32), b.optimizer.getStateBuffers()[k].read(np.uint32), "state " + k)
    assert a.getItersPerSec() > 0 and a.getLastStepMs() > 0


def test_training_reduces_the_loss_and_densify_rebuilds(hip_device):
    """30 steps with a densify at iterations 10 and 20: the point count changes, optimizer state is carried over and the L1
    error against the ground truth goes down.  (The comparison of the rebuilt cloud, state and trajectory with the oracle's
    restatement of trainer.ts is tests/test_gpu_trainer_oracle.py.)"""
    cfg = harness.small_config("c2", num_points=5000, width=128, height=96, s0=0.01)
    g, sh, _ = harness.scene(cfg)
    dev = hip_device
    cameras, images = _dataset(dev, cfg, g, sh, 4)
    dens = dict(schedule=dict(enabled=True, warmupIterations=10, interval=10, stopIterations=25), metricViews=3, cloneThresholdCount=5,
                splitScaleThreshold=0.03, pruneOpacity=0.2, maxNewPointsPerStep=300)

    def l1(trainer):
        p = harness.HipPipeline(dev, cfg, trainer.pointCloud.gaussian_3d_buffer.read(np.uint32).reshape(-1, 6),
                                trainer.pointCloud.sh_buffer.read(np.uint32).reshape(-1, 24), cameras[0]["camera"])
        p.cfg = synth.SceneConfig(cfg.config_id, trainer.pointCloud.num_points, cfg.width, cfg.height, cfg.sh_deg, cfg.fy, cfg.s0)
        p.forward()
        img = p.rast.getOutputTextureView().read(np.uint8).astype(np.float64)
        p.destroy()
        return float(np.abs(img - images[0]["texture"].read(np.uint8).astype(np.float64)).mean())

    t = Trainer(dev, seed=3)
    t.setDensifyPruneConfig(dens)
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.start()
    assert t.getNextDensifyPruneIteration() == 10
    before = l1(t)
    counts = [t.getPointCount()]
    for i in range(30):
        t.step()
        counts.append(t.getPointCount())
    after = l1(t)
    assert after < before, (before, after)
    assert t.getLastDensifyPruneIteration() == 20
    assert counts[10] != counts[9] or counts[20] != counts[19], "densify changed the point count"
    assert counts[10] <= counts[9] + 300 and counts[20] <= counts[19] + 300, "maxNewPointsPerStep respected"
    assert t.getNextDensifyPruneIteration() is None
    assert t.optimizer.getIteration() == 30
    st = t.optimizer.getStateBuffers()
    assert st["optPosBuffer"].size == 48 * t.getPointCount()
    assert np.isfinite(st["optPosBuffer"].read(np.float32)).all()
