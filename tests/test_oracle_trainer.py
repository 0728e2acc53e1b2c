"""CPU: the oracle's restatement of ``src/trainer.ts`` sequencing (oracle/oracle_trainer.py) -- schedule, multi-view metric
accumulation, the rebuild -- checked for the properties that do not need a second implementation."""
import numpy as np

from webdgs_amd import synth
from webdgs_amd.trainer import Trainer

import harness


def _scene(orc, n=3000, w=96, h=64, views=4):
    cfg = harness.small_config("c2", num_points=n, width=w, height=h, s0=0.012)
    g, sh, _ = harness.scene(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    cams = synth.circle_cameras(cfg, views)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    imgs = [orc.forward(tg, tsh, cams[i], st, ti)["rgba8"] for i in range(views)]
    return cfg, g, sh, cams, imgs, st, ti


def test_schedule_is_checked_on_the_next_iteration(orc):
    from oracle import oracle_trainer as ot
    cfg, g, sh, cams, imgs, _, _ = _scene(orc, n=500)
    o = ot.OracleTrainer(g, sh, cfg.sh_deg, list(cams), imgs, densify=dict(schedule=dict(warmupIterations=5, interval=3, stopIterations=12)))
    fired = []
    for it in range(1, 16):  # trainer.ts:593-601: warm-up, then every `interval`, inclusive stop
        if o.should_densify():
            fired.append(it)
        o.iteration += 1
    assert fired == [5, 8, 11]
    # the product's getNextDensifyPruneIteration agrees with the same rule
    t = Trainer.__new__(Trainer)
    t.densifyPruneConfig = dict(schedule=dict(enabled=True, warmupIterations=5, interval=3, stopIterations=12))
    nxt = []
    for i in range(0, 13):
        t.iteration = i
        nxt.append(t.getNextDensifyPruneIteration())
    assert nxt == [5, 5, 5, 5, 5, 8, 8, 8, 11, 11, 11, None, None]


def test_single_view_step_is_train_step_and_batches_reduce_to_it(orc):
    from oracle import oracle_trainer as ot
    cfg, g, sh, cams, imgs, st, ti = _scene(orc)
    none = dict(schedule=dict(enabled=False))
    a = ot.OracleTrainer(g, sh, cfg.sh_deg, list(cams), imgs, densify=none)
    a.step(2)
    g2, sh2 = g.copy(), sh.copy()
    state = orc.unpack(g2, sh2)
    orc.train_step(g2, sh2, state, cams[2], st, ti, imgs[2])
    assert np.array_equal(a.g, g2) and np.array_equal(a.sh, sh2)
    for k in state:
        assert np.array_equal(a.state[k].view(np.uint32), state[k].view(np.uint32)), k
    # a "batch" of one view through the fp32 path equals the fp16 path (the unpacking is exact), and 2 ranks x 1 view == 1 rank x 2 views
    b = ot.OracleTrainer(g, sh, cfg.sh_deg, list(cams), imgs, densify=none)
    c = ot.OracleTrainer(g, sh, cfg.sh_deg, list(cams), imgs, densify=none)
    for ids in ([1, 3], [0, 2], [3, 3]):
        b.step(ids, world=1)
        c.step(ids, world=2)
    assert np.array_equal(b.g, c.g) and np.array_equal(b.sh, c.sh)
    assert not np.array_equal(b.g, g)


def test_metric_counts_accumulate_over_views_and_divide_by_views_used(orc):
    from oracle import oracle_trainer as ot
    cfg, g, sh, cams, imgs, _, _ = _scene(orc, n=4000, w=128, h=96)
    dens = dict(schedule=dict(enabled=True, warmupIterations=1, interval=50, stopIterations=10), metricViews=3, metricThreshold=0.3, cloneThresholdCount=4,
                splitScaleThreshold=0.03, pruneOpacity=0.2, maxNewPointsPerStep=200)
    o = ot.OracleTrainer(g, sh, cfg.sh_deg, list(cams), imgs, densify=dens)
    o.step(0, metric_view_ids=[1, 2, 1, 3])  # the fourth draw is never reached: metricViews = 3
    d = o.last_densify
    assert d["used_views"] == 3 and [p["view"] for p in d["per_view"]] == [1, 2, 1]
    per = [p["counts_after"].astype(np.int64) for p in d["per_view"]]
    assert np.array_equal(per[2], d["counts_raw"]) and (per[1] >= per[0]).all() and (per[2] >= per[1]).all()
    assert np.array_equal(per[2] - per[1], per[0]), "the same view drawn twice adds the same counts again (clear:false)"
    assert np.array_equal(d["counts"], d["counts_raw"] // 3)
    assert (d["counts_raw"] % 3 != 0).any(), "the integer division truncates somewhere"
    assert d["out_n"] <= o.densify["maxNewPointsPerStep"] + 4000
    # the metrics camera keeps the pose and rescales the intrinsics to the metrics canvas
    mc = d["per_view"][0]["camera"]
    assert np.array_equal(mc[0:16], cams[1][0:16]) and tuple(mc[64:66]) == (64.0, 48.0)
    assert abs(float(mc[66]) - float(cams[1][66]) * 0.5) < 1e-3
    assert np.array_equal(Trainer.metrics_camera(cams[1], 64, 48).view(np.uint32), mc.view(np.uint32))
