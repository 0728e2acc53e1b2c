"""GPU: tickets (wdgs_queue_mark / wdgs_queue_wait) and the Trainer's pipelined step.

trainer.ts:639-645 awaits `onSubmittedWorkDone()` inside every step.  With `pipeline_depth = 2` the Trainer keeps that promise and
awaits the PREVIOUS step's instead, so step k+1 is prepared and submitted while step k runs.  Nothing about the work changes, so the
check is bit-equality with depth 1 -- single-view and batched, across a densify rebuild -- plus the error path: a step whose tile-entry
list overflowed must leave the parameters untouched and be reported, at the latest, by the step after it."""
import numpy as np
import pytest

from webdgs_amd import ops
from webdgs_amd.trainer import Trainer

import dp_common
from harness import assert_bits_equal

pytestmark = pytest.mark.gpu


def _train(dev, depth, vpr, steps=14, densify_at=8, interval=1000, reuse_passes=True):
    cfg, g, sh, cameras, images = dp_common.dataset(dev)
    t = Trainer(dev, seed=9, views_per_rank=vpr, pipeline_depth=depth)
    t.reuse_passes = reuse_passes
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=True, warmupIterations=densify_at, interval=interval, stopIterations=10 ** 6),
                                 metricViews=3, cloneThresholdCount=5, splitScaleThreshold=0.03, pruneOpacity=0.2, maxNewPointsPerStep=300))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.start()
    in_flight = 0
    for ids in dp_common.view_schedule(steps, 1, vpr):
        t.step(ids)
        in_flight = max(in_flight, len(t._tickets))
    t.drain()
    assert not t._tickets
    dev.synchronize()
    out = dict(g=t.pointCloud.gaussian_3d_buffer.read(np.uint32), sh=t.pointCloud.sh_buffer.read(np.uint32), n=t.getPointCount(),
               state={k: b.read(np.uint32) for k, b in t.optimizer.getStateBuffers().items()}, iteration=t.getIteration(), in_flight=in_flight)
    t.destroy()
    return out


@pytest.mark.parametrize("vpr", [1, 3])
def test_pipelined_steps_leave_the_same_bits(hip_device, vpr):
    a = _train(hip_device, 2, vpr)
    b = _train(hip_device, 1, vpr)
    assert a["in_flight"] == 1 and b["in_flight"] == 0, "depth 2 keeps one step in flight after step() returns, depth 1 none"
    assert a["n"] == b["n"] != 6000 and a["iteration"] == b["iteration"] == 14, "same rebuild, same count of steps"
    assert_bits_equal(a["g"], b["g"], f"gaussians, {vpr} view(s) per step: pipelined vs awaited")
    assert_bits_equal(a["sh"], b["sh"], "sh: pipelined vs awaited")
    for k in a["state"]:
        assert_bits_equal(a["state"][k], b["state"][k], f"optimizer state {k}: pipelined vs awaited")


@pytest.mark.parametrize("vpr", [1, 2])
def test_resized_passes_equal_rebuilt_passes(hip_device, vpr):
    """applyPointCloudSwap keeps the passes and resizes them (`wdgs_tiled_*_resize`); the reference destroys and rebuilds them
    (trainer.ts:201-237).  Three rebuilds, the cloud growing and shrinking: both ways must leave the same bits."""
    a = _train(hip_device, 1, vpr, steps=16, densify_at=4, interval=4, reuse_passes=True)
    b = _train(hip_device, 1, vpr, steps=16, densify_at=4, interval=4, reuse_passes=False)
    assert a["n"] == b["n"] != 6000
    assert_bits_equal(a["g"], b["g"], "gaussians: resized vs rebuilt passes")
    assert_bits_equal(a["sh"], b["sh"], "sh: resized vs rebuilt passes")
    for k in a["state"]:
        assert_bits_equal(a["state"][k], b["state"][k], f"optimizer state {k}: resized vs rebuilt passes")


@pytest.mark.parametrize("vpr,depth", [(1, 1), (1, 2), (3, 1), (3, 2)])
def test_overflowing_step_is_reported_and_changes_nothing(hip_device, vpr, depth):
    dev = hip_device
    cfg, g, sh, cameras, images = dp_common.dataset(dev)
    t = Trainer(dev, seed=9, views_per_rank=vpr, pipeline_depth=depth, maxTileEntries=4096)  # the scene needs several times that
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.start()
    before = t.pointCloud.gaussian_3d_buffer.read(np.uint32)
    with pytest.raises(ops.CapacityError):
        for _ in range(depth):  # depth 1: the step itself raises; depth 2: the next one does
            t.step([0] * vpr)
    try:
        dev.synchronize()
    except ops.CapacityError:
        pass  # (the step that was still in flight overflowed too)
    assert_bits_equal(t.pointCloud.gaussian_3d_buffer.read(np.uint32), before, "an overflowed step must not touch the parameters")
    t.destroy()


@pytest.mark.parametrize("depth", [1, 2])
def test_lists_sized_by_the_library_grow_after_an_overflow(hip_device, depth):
    """With ``maxTileEntries`` left to the library (30 entries per Gaussian, at least 2^20) a cloud of few, large splats outruns the lists --
    what a long run of the default schedule arrives at (c3 after ~3 000 iterations).  The Trainer then doubles them, says so, and goes on
    training; a pinned capacity stays an error (the test above)."""
    import warnings
    from webdgs_amd import synth
    dev = hip_device
    cfg = synth.SceneConfig(2, 6000, 512, 384, 1, 550.0, 0.2, "few-large-splats")   # 768 tiles, splats that cover most of them
    g, sh = synth.make_gaussians(cfg)
    cams = synth.circle_cameras(cfg, 2)
    images = [dict(texture=dev.bufferFrom(np.zeros((cfg.height, cfg.width, 4), np.uint8)), width=cfg.width, height=cfg.height) for _ in cams]
    t = Trainer(dev, seed=9, pipeline_depth=depth)
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset([dict(camera=c, width=cfg.width, height=cfg.height) for c in cams], images)
    t.start()
    before = t.pointCloud.gaussian_3d_buffer.read(np.uint32).copy()
    try:
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            for i in range(8):
                t.step([i % 2])
            t.drain()
            dev.synchronize()
        grown = [w for w in seen if issubclass(w.category, RuntimeWarning) and "tile-entry lists grown" in str(w.message)]
        assert grown, "the overflow was reported as a warning"
        cap = int(t.forwardPass.getResources()["maxTileEntries"])
        needed = int(t.forwardPass.check()[0])
        assert cap > (1 << 20) and needed > (1 << 20) and cap >= needed, (cap, needed)
        assert (t.pointCloud.gaussian_3d_buffer.read(np.uint32) != before).any(), "training went on after the lists had grown"
        assert t.getIteration() >= 8 - 2 * len(grown), "only the steps that overflowed are lost"
    finally:
        t.destroy()


def test_another_owners_overflow_is_not_the_trainers_to_answer(hip_device):
    """A Viewer renders the cloud a Trainer trains, through a forward pass of its own whose lists overflow at every frame (as they do beside a
    long run of the default schedule).  The device-wide report names the passes; the Trainer grows ITS lists once, for its own overflow, says
    once that someone else's pass overflowed, and is not driven to ever larger lists by frames it does not own."""
    import warnings
    from webdgs_amd import synth
    from webdgs_amd.viewer import Viewer
    dev = hip_device
    cfg = synth.SceneConfig(2, 6000, 512, 384, 1, 550.0, 0.2, "few-large-splats")
    g, sh = synth.make_gaussians(cfg)
    cams = synth.circle_cameras(cfg, 2)
    images = [dict(texture=dev.bufferFrom(np.zeros((cfg.height, cfg.width, 4), np.uint8)), width=cfg.width, height=cfg.height) for _ in cams]
    pc = ops.createPointCloud(dev, g, sh, cfg.sh_deg)
    v = Viewer(dev, cfg.width, cfg.height)
    v.setCamera(cams[0])
    v.setPointCloud(pc)
    v.setRenderMode("gaussian")
    t = Trainer(dev, seed=9, pipeline_depth=2)
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    t.setPointCloud(pc)
    t.setDataset([dict(camera=c, width=cfg.width, height=cfg.height) for c in cams], images)
    t.start()
    try:
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            for i in range(12):
                v.render(None)
                t.step([i % 2])
            t.drain()
            try:
                dev.synchronize()
            except ops.CapacityError:
                pass  # (the last frame's report, which nobody has taken yet)
        texts = [str(w.message) for w in seen if issubclass(w.category, RuntimeWarning)]
        grown = [m for m in texts if "tile-entry lists grown" in m]
        foreign = [m for m in texts if "not this trainer's" in m]
        assert len(grown) == 1 and len(foreign) == 1, texts
        cap = int(t.forwardPass.getResources()["maxTileEntries"])
        assert (1 << 20) < cap <= (4 << 20), cap
        assert int(v.getForwardPass().getResources()["maxTileEntries"]) <= (1 << 20) + 4096, "the viewer's lists are its owner's business"
        assert t.getIteration() >= 9
        # ... and the owner gets to answer: the reports the trainer's waits consumed were left for it (ops.CapacityReports), its next read finds one
        assert dev.capacityReports.pending, "reports about the viewer's pass wait for the viewer"
        v.readFrame()
        assert int(v.getForwardPass().getResources()["maxTileEntries"]) > (1 << 20) + 4096, "the viewer enlarged its lists at its own read"
        dev.capacityReports.pending.clear()
    finally:
        t.pointCloud = None  # (the cloud is shared: destroyed below, once)
        t.destroy()
        v.destroy()
        pc.gaussian_3d_buffer.destroy(); pc.sh_buffer.destroy()


@pytest.mark.parametrize("depth", [1, 2])
def test_kernel_times_are_collected_by_ticket_waits(hip_device, depth):
    """bench.py's per-kernel leg: eager steps with an event pair around every launch; the pairs are folded into the totals by the
    waits of the steps themselves (not only by a full synchronize), and none is lost when a wait finds later kernels unfinished."""
    dev = hip_device
    cfg, g, sh, cameras, images = dp_common.dataset(dev)
    t = Trainer(dev, seed=9, pipeline_depth=depth, use_command_buffers=False)
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.start()
    t.step([0])
    dev.synchronize()
    dev.setProfiling(True)
    dev.kernelTimes(reset=True)
    for v in (0, 1, 2, 3, 0):
        t.step([v])
    t.drain()
    dev.setProfiling(False)
    times = dev.kernelTimes(reset=True)
    t.destroy()
    for name in ("project_count", "rasterize", "loss_grad", "backward_rasterize", "geometry_backward_adam"):
        assert name in times and times[name][0] == 5 and times[name][1] > 0.0, (name, times.get(name))


def test_tickets(hip_device):
    dev = hip_device
    buf = dev.createBuffer(64 << 20)
    with pytest.raises(ops.WdgsError):
        dev.queue.wait(0)
    with pytest.raises(ops.WdgsError):
        dev.queue.wait(10 ** 9)  # never issued
    tickets = []
    for i in range(20):  # more marks than the ring holds: old tickets stay waitable (they wait for the mark that took their slot)
        buf.clear()
        tickets.append(dev.queue.mark())
    assert tickets == list(range(tickets[0], tickets[0] + 20))
    dev.queue.wait(tickets[0])
    dev.queue.wait(tickets[-1])
    assert not buf.read(np.uint32, count=16).any()
    with dev.createCommandEncoder("tickets", record=True) as encoder:
        encoder.clearBuffer(buf)
        with pytest.raises(ops.WdgsError):
            dev.queue.mark()  # not while recording
        encoder.finish().destroy()
