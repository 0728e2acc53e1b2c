"""GPU: the two lanes of a device (include/webdgs.h "Lanes") and the batched step that alternates its views between them.

Running view k+1's projection / sort / loss beside view k's rasterization kernels must not change a single bit: each view's K1..K17
is the same work on its own op set, and the sums into the fp32 block are ordered across the lanes in view order.  The lanes are a
scheduling device only, so the check is equality with the same trainer run with overlap switched off."""
import numpy as np
import pytest

from webdgs_amd import ops
from webdgs_amd.trainer import Trainer

import dp_common
from harness import assert_bits_equal

pytestmark = pytest.mark.gpu


def _train(dev, overlap, steps, vpr, densify_at=None, batch_views=None, long_lists=None):
    cfg, g, sh, cameras, images = dp_common.dataset(dev)
    t = Trainer(dev, seed=5, world_size=1, rank=0, views_per_rank=vpr, overlap_views=overlap, batch_views=batch_views)
    t.longLists = long_lists
    sched = dict(enabled=False) if densify_at is None else dict(enabled=True, warmupIterations=densify_at, interval=1000, stopIterations=10 ** 6)
    t.setDensifyPruneConfig(dict(schedule=sched, metricViews=3, cloneThresholdCount=5, splitScaleThreshold=0.03, pruneOpacity=0.2, maxNewPointsPerStep=300))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.start()
    t.step([0] * vpr)            # eager (first-use allocations), lane 0 only
    taken = 1 + t.warmupCommandBuffers()  # records every (view, op set) command buffer
    replays_on_lanes = 0
    for ids in dp_common.view_schedule(steps, 1, vpr):
        before = dict(t._cmd_cache)
        t.step(ids)
        key = (lambda k, v: ("viewp", v, k)) if t.batch_views else (lambda k, v: ("view", v, k % t._lanes))
        replays_on_lanes += int(t._lanes > 1 and all(key(k, v) in before for k, v in enumerate(ids)))
    dev.synchronize()
    out = dict(g=t.pointCloud.gaussian_3d_buffer.read(np.uint32), sh=t.pointCloud.sh_buffer.read(np.uint32), n=t.getPointCount(),
               state={k: b.read(np.uint32) for k, b in t.optimizer.getStateBuffers().items()}, iteration=t.optimizer.getIteration(),
               taken=taken + steps, on_lanes=replays_on_lanes)
    t.destroy()
    return out


@pytest.mark.parametrize("batch_views", [True, False], ids=["view-batched-K1-K17", "per-view-kernels"])
@pytest.mark.parametrize("vpr,lanes", [(2, 2), (5, 2), (5, 3), (7, 4)])
def test_overlapped_views_leave_the_same_bits(hip_device, vpr, lanes, batch_views):
    """`batch_views`: K1 and K17 of all the step's views in one launch each, one op set per view (round 3), or round 2's per-view kernels
    with one op set per lane.  Either way, on lanes or not, the bits are those of the per-view, one-lane run."""
    steps = 6
    a = _train(hip_device, lanes, steps, vpr, batch_views=batch_views)
    b = _train(hip_device, False, steps, vpr, batch_views=False)
    assert a["on_lanes"] == steps and b["on_lanes"] == 0, "the overlapped run replayed on both lanes, the other one never did"
    assert a["iteration"] == b["iteration"] == a["taken"]
    assert_bits_equal(a["g"], b["g"], f"gaussians, {vpr} views per step: two lanes vs one")
    assert_bits_equal(a["sh"], b["sh"], f"sh, {vpr} views per step: two lanes vs one")
    for k in a["state"]:
        assert_bits_equal(a["state"][k], b["state"][k], f"optimizer state {k}: two lanes vs one")


def test_long_tile_lists_on_three_lanes(hip_device):
    """the long-list passes of round 4 kept their scratch in function-local statics, which views running on different lanes would have
    shared.  The scratch now belongs to the forward pass (one per view of the batch): five views per step on three lanes with per-pixel lists for
    every tile above 40 entries -- in recorded command buffers, across a densify rebuild -- leave the bits of the one-lane run with the lists off."""
    steps, vpr = 8, 5
    a = _train(hip_device, 3, steps, vpr, densify_at=9, long_lists=dict(threshold=40, maxItems=8192, maxRows=65536))
    b = _train(hip_device, False, steps, vpr, densify_at=9, long_lists=dict(threshold=0))
    assert a["on_lanes"] > 0 and a["n"] == b["n"] != 6000
    assert_bits_equal(a["g"], b["g"], "gaussians: long lists on three lanes vs none on one")
    assert_bits_equal(a["sh"], b["sh"], "sh: long lists on three lanes vs none on one")
    for k in a["state"]:
        assert_bits_equal(a["state"][k], b["state"][k], f"optimizer state {k}: long lists on three lanes vs none on one")


def test_overlap_survives_a_densify_rebuild(hip_device):
    """A rebuild drops both op sets and every recording; the steps after it go eager -> recorded -> on the lanes again."""
    steps, vpr = 8, 3
    a = _train(hip_device, True, steps, vpr, densify_at=9)
    b = _train(hip_device, False, steps, vpr, densify_at=9)
    assert a["n"] == b["n"] and a["n"] != 6000, "the rebuild happened (and the same one)"
    assert_bits_equal(a["g"], b["g"], "gaussians after a rebuild: two lanes vs one")
    assert_bits_equal(a["sh"], b["sh"], "sh after a rebuild: two lanes vs one")


def test_lane_calls_validate_their_arguments(hip_device):
    dev = hip_device
    with pytest.raises(ops.WdgsError):
        dev.selectLane(ops.MAX_LANES)
    with pytest.raises(ops.WdgsError):
        dev.laneOrder(0, -1)
    dev.laneOrder(0, 0)  # a lane is in order with itself
    dst = dev.createBuffer(4 << 20)
    with dev.createCommandEncoder("lanes", record=True) as encoder:
        encoder.clearBuffer(dst)
        with pytest.raises(ops.WdgsError):
            dev.selectLane(1)  # not while recording
        with pytest.raises(ops.WdgsError):
            dev.laneOrder(0, 1)
        encoder.finish().destroy()
    # work given to lane 1 is ordered in front of lane 0's by laneOrder, and the host's synchronize waits for both lanes
    ramp = np.arange(1 << 20, dtype=np.uint32)
    dev.selectLane(1)
    try:
        dst.write(ramp)
    finally:
        dev.selectLane(0)
    dev.laneOrder(0, 1)
    assert np.array_equal(dst.read(np.uint32), ramp)
