"""GPU: the single-view step that draws its view one step ahead and lets the kernel that updates a Gaussian project it for the next view
(`Trainer.project_ahead`, `wdgs_optimizer_step_with_geometry_project`).  Nothing about the arithmetic changes -- K1 runs on the same
re-packed Gaussian under the same camera, one kernel earlier -- so the checks are bit-equality with the plain step over a run that
crosses densify passes (where no view is drawn ahead), the random sequence consumed identically, and the form really being used."""
import numpy as np
import pytest

from webdgs_amd import ops
from webdgs_amd.trainer import Trainer

import dp_common
from harness import assert_bits_equal

pytestmark = pytest.mark.gpu


def _train(dev, ahead, steps=23, depth=2, eager=False):
    cfg, g, sh, cameras, images = dp_common.dataset(dev)
    t = Trainer(dev, seed=21, pipeline_depth=depth)
    t.project_ahead = ahead
    t.use_command_buffers = not eager
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=True, warmupIterations=7, interval=6, stopIterations=10 ** 6),
                                 metricViews=3, cloneThresholdCount=5, splitScaleThreshold=0.03, pruneOpacity=0.2, maxNewPointsPerStep=300))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.start()
    views, used_projection = [], 0
    for _ in range(steps):
        before = t._projected_view
        t.step()
        used_projection += int(before is not None)
        views.append(before)
    t.drain()
    dev.synchronize()
    keys = sorted(k for k in t._cmd_cache if k[0] == "views")
    out = dict(g=t.pointCloud.gaussian_3d_buffer.read(np.uint32), sh=t.pointCloud.sh_buffer.read(np.uint32), n=t.getPointCount(),
               state={k: b.read(np.uint32) for k, b in t.optimizer.getStateBuffers().items()}, iteration=t.getIteration(),
               opt_iteration=t.optimizer.getIteration(), rng=t._rng.getstate(), ahead_view=t._ahead_view, used_projection=used_projection, keys=keys)
    t.destroy()
    return out


@pytest.mark.parametrize("eager", [False, True], ids=["recorded", "eager"])
def test_projecting_ahead_leaves_the_same_bits(hip_device, eager):
    a = _train(hip_device, True, eager=eager)
    b = _train(hip_device, False, eager=eager)
    assert a["n"] == b["n"] != 6000 and a["iteration"] == b["iteration"] == 23 and a["opt_iteration"] == b["opt_iteration"]
    # densify passes after iterations 7, 13, 19: the steps before them draw nothing ahead, every other step does
    assert a["used_projection"] == 23 - 1 - 3 and b["used_projection"] == 0, (a["used_projection"], b["used_projection"])
    if not eager:
        assert any(k[2] for k in a["keys"]) and not b["keys"], "the recorded form that starts at the scan was used"
    assert_bits_equal(a["g"], b["g"], "gaussians: projected ahead vs plain step")
    assert_bits_equal(a["sh"], b["sh"], "sh: projected ahead vs plain step")
    for k in a["state"]:
        assert_bits_equal(a["state"][k], b["state"][k], f"optimizer state {k}: projected ahead vs plain step")
    # the random sequence: the run that draws ahead holds one more sample -- the view of step 24 -- and is otherwise where the plain run is
    import random
    r = random.Random()
    r.setstate(b["rng"])
    assert a["ahead_view"] == r.randrange(4) and r.getstate() == a["rng"]


def test_an_explicit_view_discards_or_uses_the_projection(hip_device):
    """A step with explicit view ids after steps that drew ahead: it uses the projection when it names the projected view and ignores it
    otherwise; the view drawn ahead stays in the sequence for the next drawn step."""
    cfg, g, sh, cameras, images = dp_common.dataset(hip_device)

    def run(explicit):
        t = Trainer(hip_device, seed=5)
        t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
        t.setPointCloud(ops.createPointCloud(hip_device, g, sh, cfg.sh_deg))
        t.setDataset(cameras, images)
        t.start()
        for _ in range(4):
            t.step()
        seq = [t._ahead_view]
        proj = t._projected_view
        assert proj == t._ahead_view
        t.step([proj if explicit == "same" else (proj + 1) % 4])
        assert t._projected_view is None and t._ahead_view == seq[0]
        t.step()
        t.drain()
        hip_device.synchronize()
        out = t.pointCloud.gaussian_3d_buffer.read(np.uint32).copy()
        t.destroy()
        return out, proj

    for explicit in ("same", "other"):
        got, proj = run(explicit)
        ref = Trainer(hip_device, seed=5)
        ref.project_ahead = False
        ref.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
        ref.setPointCloud(ops.createPointCloud(hip_device, g, sh, cfg.sh_deg))
        ref.setDataset(cameras, images)
        ref.start()
        for _ in range(4):
            ref.step()
        nxt = ref._rng.randrange(4)           # what the other run drew ahead
        assert nxt == proj
        ref.step([proj if explicit == "same" else (proj + 1) % 4])
        ref.step([nxt])
        ref.drain()
        hip_device.synchronize()
        assert_bits_equal(got, ref.pointCloud.gaussian_3d_buffer.read(np.uint32), f"explicit step on the {explicit} view")
        ref.destroy()
