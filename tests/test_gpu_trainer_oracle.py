"""GPU: the ``Trainer`` class itself against the oracle's restatement of ``src/trainer.ts`` (oracle/oracle_trainer.py) -- the same
fixed view list, the reference's schedule logic, a densify/prune rebuild in the middle with several metric views accumulated
(``clear: false``) and divided by the views used, and training continued on the rebuilt cloud.  Every buffer is compared bit for bit
at every step that matters: point cloud, SH, all six optimizer-state arrays, iteration counters, metric counts, decisions."""
import numpy as np
import pytest

from webdgs_amd import ops, synth
from webdgs_amd.trainer import Trainer

import harness
from harness import assert_bits_equal

pytestmark = pytest.mark.gpu

STATE_KEYS = dict(optPosBuffer=("opt_pos", 12), optRotBuffer=("opt_rot", 12), optScaleBuffer=("opt_scale", 12), optOpacityBuffer=("opt_opacity", 3),
                  paramSH=("param_sh", 48), stateSH=("state_sh", 96))


class _FixedViews:
    """Stands in for the trainer's ``random.Random``: hands out a fixed list of view indices (the reference draws Math.random())."""

    def __init__(self, views):
        self.views, self.i = list(views), 0

    def randrange(self, n):
        v = self.views[self.i]
        self.i += 1
        assert 0 <= v < n
        return v


def _dataset(dev, orc, cfg, g, sh, n_views):
    tg, tsh = synth.make_target_scene(g, sh)
    cams = synth.circle_cameras(cfg, n_views)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    imgs = [orc.forward(tg, tsh, cams[i], st, ti)["rgba8"] for i in range(n_views)]
    cameras = [dict(camera=cams[i], width=cfg.width, height=cfg.height) for i in range(n_views)]
    images = [dict(texture=dev.bufferFrom(imgs[i]), width=cfg.width, height=cfg.height) for i in range(n_views)]
    return cams, imgs, cameras, images


def _compare(t, o, what):
    n = o.num_points
    assert t.getPointCount() == n, what
    assert_bits_equal(t.pointCloud.gaussian_3d_buffer.read(np.uint32).reshape(-1, 6)[:n], o.g, f"{what}: gaussians")
    assert_bits_equal(t.pointCloud.sh_buffer.read(np.uint32).reshape(-1, 24)[:n], o.sh, f"{what}: sh")
    bufs = t.optimizer.getStateBuffers()
    for k, (ok, width) in STATE_KEYS.items():
        assert_bits_equal(bufs[k].read(np.float32).reshape(-1, width)[:n], o.state[ok], f"{what}: optimizer state {k}")
    assert t.getIteration() == o.iteration and t.optimizer.getIteration() == o.optimizer_iteration, what


@pytest.mark.parametrize("use_cb,long_lists", [(True, None), (False, None), (True, 40)], ids=["recorded", "eager", "recorded-long-lists"])
def test_trainer_trajectory_equals_the_oracle_trainer(hip_device, orc, use_cb, long_lists):
    from oracle import oracle_trainer
    dev = hip_device
    cfg = harness.small_config("c2", num_points=5000, width=128, height=96, s0=0.01)
    g, sh, _ = harness.scene(cfg)
    cams, imgs, cameras, images = _dataset(dev, orc, cfg, g, sh, 4)
    dens = dict(schedule=dict(enabled=True, warmupIterations=12, interval=10, stopIterations=25), metricViews=3, cloneThresholdCount=5,
                splitScaleThreshold=0.03, pruneOpacity=0.2, maxNewPointsPerStep=300)
    steps = 27
    rng = np.random.default_rng(5)
    train_views = [int(v) for v in rng.integers(0, 4, steps)]
    metric_views = {12: [2, 0, 3], 22: [1, 1, 2]}  # iteration -> the views runDensifyPruneMultiView draws (a repeat is allowed)

    o = oracle_trainer.OracleTrainer(g, sh, cfg.sh_deg, list(cams), imgs, densify=dens)
    t = Trainer(dev, seed=0, use_command_buffers=use_cb)
    if long_lists:   # per-pixel lists (csrc/longlist.h) for every tile above 40 entries: most tiles, in the training and the metric passes, in recorded command buffers
        t.longLists = dict(threshold=long_lists, maxItems=8192, maxRows=65536)
    t.setDensifyPruneConfig(dens)
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.start()
    sizes = [t.getPointCount()]
    try:
        for i in range(steps):
            it = i + 1
            draws = [train_views[i]] + metric_views.get(it, [])
            t._rng = _FixedViews(draws)
            assert o.should_densify() == (it in metric_views)
            o.step(train_views[i], metric_view_ids=metric_views.get(it))
            t.step()
            assert t._rng.i == len(draws), "the trainer drew a different number of views than the schedule says"
            sizes.append(t.getPointCount())
            if it in metric_views or it in (1, 2, 11, 13, 21, 23, steps):
                _compare(t, o, f"after iteration {it}")
            if it in metric_views:
                d = o.last_densify
                assert d["used_views"] == 3 and d["rebuilt"], "the oracle densified with three metric views and rebuilt the cloud"
                assert int(d["counts_raw"].max()) >= 3, "counts were accumulated over the views (divisor > 1 matters)"
                assert t.getLastDensifyPruneIteration() == o.last_densify_iteration == it
        _compare(t, o, "end of run")
        if long_lists:
            st = t.forwardPass.longListStats()
            # (whether a block's pixels got lists of their own or its walk task took the plain walk -- lists pay only when 6 x shorter than the tile's --
            # is the scan task's decision; tests/test_gpu_nan.py::test_long_list_tasks_and_their_fallbacks pins scenes of either kind)
            assert st["threshold"] == long_lists and st["blocksWanted"] >= 40 and st["stalled"] == 0 and st["forwardQueue"] >= 2 * (st["itemsWanted"] + st["blocksWanted"]), st
        assert sizes[12] != sizes[11] and sizes[22] != sizes[21], sizes
        assert t.getNextDensifyPruneIteration() is None
    finally:
        t.destroy()


def test_multi_view_metric_accumulation_and_decisions_equal_the_oracle(hip_device, orc):
    """``runDensifyPruneMultiView`` stage by stage: per-view metric images, counts accumulated with clear:false over 4 views (one
    drawn twice), the integer division by usedViews, actions / out-counts / offsets / total and the rebuilt cloud + state."""
    from oracle import oracle_trainer
    dev = hip_device
    cfg = harness.small_config("c2", num_points=8000, width=160, height=112, s0=0.012)
    g, sh, _ = harness.scene(cfg)
    cams, imgs, cameras, images = _dataset(dev, orc, cfg, g, sh, 5)
    dens = dict(schedule=dict(enabled=True, warmupIterations=3, interval=50, stopIterations=100), metricViews=4, metricThreshold=0.35, cloneThresholdCount=4,
                splitScaleThreshold=0.03, pruneOpacity=0.25, maxNewPointsPerStep=500)
    o = oracle_trainer.OracleTrainer(g, sh, cfg.sh_deg, list(cams), imgs, densify=dens)
    t = Trainer(dev, seed=0)
    t.setDensifyPruneConfig(dens)
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.start()
    try:
        mviews = [4, 1, 4, 2]
        for it, v in enumerate([0, 3, 1], start=1):
            draws = [v] + (mviews if it == 3 else [])
            t._rng = _FixedViews(draws)
            if it == 3:  # keep the densify inputs: the trainer swaps the cloud inside step()
                captured = {}
                orig = t.densifyPrune.encodePrepare

                def spy(encoder, inputs, _orig=orig):
                    captured["metric_counts"] = inputs["metricCountsBuffer"].read(np.uint32)[: t.pointCloud.num_points].copy()
                    out = _orig(encoder, inputs)
                    n = t.pointCloud.num_points
                    captured.update(actions=out["actionBuffer"].read(np.uint32)[:n].copy(), counts=out["outCountBuffer"].read(np.uint32)[:n].copy(),
                                    offsets=out["outOffsetBuffer"].read(np.uint32)[:n].copy(), total=int(out["outTotalBuffer"].read(np.uint32)[0]),
                                    max_out=out["maxOutPoints"])
                    return out
                t.densifyPrune.encodePrepare = spy
            o.step(v, metric_view_ids=mviews if it == 3 else None)
            t.step()
        d = o.last_densify
        assert d["used_views"] == 4 and d["rebuilt"]
        assert int(d["counts_raw"].max()) > 4 or int((d["counts_raw"] % 4 != 0).sum()) > 0, "the integer division by usedViews truncates somewhere"
        assert_bits_equal(captured["metric_counts"], d["counts"], "metric counts after accumulation over 4 views and division by 4")
        assert_bits_equal(captured["actions"], d["prepared"]["actions"], "actions")
        assert_bits_equal(captured["counts"], d["prepared"]["counts"], "out counts")
        assert_bits_equal(captured["offsets"], d["prepared"]["offsets"], "out offsets")
        assert captured["total"] == d["prepared"]["total"] and captured["max_out"] == d["max_out"]
        _compare(t, o, "after the rebuild")
    finally:
        t.destroy()


def test_a_failed_recording_leaves_the_device_usable(hip_device):
    """an encode that fails between wdgs_encoder_begin and wdgs_encoder_finish must not leave the stream in capture mode."""
    from webdgs_amd import _lib
    dev = hip_device
    cfg = harness.small_config("c1", num_points=2000, width=64, height=48)
    g, sh, cam = harness.scene(cfg)
    pipe = harness.HipPipeline(dev, cfg, g, sh, cam)
    try:
        with pytest.raises(_lib.StateError):
            with dev.createCommandEncoder("doomed", record=True) as enc:
                pipe.fwd.encode(enc)
                pipe.rast.encode(enc, cfg.width, cfg.height)  # first use allocates its textures: refused inside a recording
        dev.synchronize()  # would raise WDGS_E_STATE if the capture were still open
        pipe.forward()     # eager work runs
        with dev.createCommandEncoder("fine", record=True) as enc:  # and a new recording can be opened and replayed
            pipe.fwd.encode(enc)
            pipe.rast.encode(enc, cfg.width, cfg.height)
            cmd = enc.finish()
        a = pipe.rast.getOutputTextureView().read(np.uint8).copy()
        dev.queue.submit([cmd])
        dev.synchronize()
        assert np.array_equal(a, pipe.rast.getOutputTextureView().read(np.uint8))
        cmd.destroy()
        with pytest.raises(_lib.StateError):  # uploads do not belong inside a recording
            with dev.createCommandEncoder("upload", record=True):
                pipe.camera.write(np.zeros(68, np.float32))
        dev.synchronize()
    finally:
        pipe.destroy()
