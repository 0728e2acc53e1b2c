"""GPU: Gaussians whose fp16 fields hold NaN or infinity -- what a long run of the reference's default schedule produces (its K17 deviates from
the gradient of its own forward pass, DESIGN.md section 2, and Adam then drives some Gaussians out of the number range).  Such a Gaussian passes
every rejection test of K1 that is written as a comparison (a comparison with NaN is false), lands in tile 0 and is walked by every pixel of it:
the 10 600-entry list of profiles/r06z_timelines_late_regime.txt is made of them.  The whole step must still equal the oracle.

Equality here is bit for bit with ONE allowance: a NaN equals a NaN whatever its sign and payload.  Which NaN an operation returns is the one thing
two IEEE machines do not agree on (x86 returns the first operand's payload and generates the negative "indefinite", the GPU propagates by source
operand priority and generates the positive one; for a fused multiply-add the x86 choice even depends on the instruction form the compiler
picked), and nothing downstream can tell them apart: a NaN converts to the fixed-point 0, to the texel 0, and stays a NaN in every sum."""
import numpy as np
import pytest

from webdgs_amd import synth

import harness

pytestmark = pytest.mark.gpu

NAN16, INF16 = 0x7E00, 0x7C00


def differences(a, b, what, fp=None):
    """'' if a == b bit for bit (NaNs of the float type `fp` counting as equal), else a one-line description."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape:
        return f"{what}: shape {a.shape} vs {b.shape}"
    if fp is not None:
        fa, fb = a.view(fp).reshape(-1), b.view(fp).reshape(-1)
        ua, ub = fa.view(np.uint16 if fp == np.float16 else np.uint32), fb.view(np.uint16 if fp == np.float16 else np.uint32)
        bad = np.flatnonzero((ua != ub) & ~(np.isnan(fa) & np.isnan(fb)))
        if bad.size:
            return f"{what}: {bad.size} of {fa.size} values differ; first at {bad[0]}: {fa[bad[0]]!r} ({ua[bad[0]]:#x}) vs {fb[bad[0]]!r} ({ub[bad[0]]:#x})"
        return ""
    av, bv = a.view(np.uint8).reshape(-1), b.view(np.uint8).reshape(-1)
    if not np.array_equal(av, bv):
        idx = np.unique(np.flatnonzero(av != bv) // a.dtype.itemsize)
        return f"{what}: {idx.size} of {a.size} elements differ; first at {idx[0]}: {a.reshape(-1)[idx[0]]!r} vs {b.reshape(-1)[idx[0]]!r}"
    return ""


def step_differences(orc, dev, cfg, g, sh, cam, target, steps=2, pipeline_factory=None, keep=False):
    """Runs `steps` training steps through the operator classes and through the oracle; returns the list of stages that differ (all of them, in order)."""
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    ref_g, ref_sh = g.copy(), sh.copy()
    ref_state = orc.unpack(ref_g, ref_sh)
    pipe = (pipeline_factory or harness.HipPipeline)(dev, cfg, g, sh, cam)
    tbuf = dev.bufferFrom(target)
    out = []
    try:
        for step in range(steps):
            ref = orc.train_step(ref_g, ref_sh, ref_state, cam, st, ti, target)
            pipe.train_step(tbuf)
            dev.synchronize()
            got = pipe.collect_forward()
            n = cfg.num_points
            e = ref["total_entries"]
            acc = harness.acc_to_reference_layout(pipe.bwd.getAccumulatorsBuffer().read(np.int32), n)
            state = pipe.read_state()
            checks = [(np.array([int(got["stats"][0])]), np.array([e]), "E", None),
                      # (a culled Gaussian's Splat is whatever an earlier step left there: only the visible rows are defined)
                      (got["splats"][ref["tile_counts"] > 0], ref["splats"][ref["tile_counts"] > 0], "splats", np.float16), (got["tile_counts"], ref["tile_counts"], "tile counts", None),
                      (got["sorted_keys"], ref["sorted_keys"][:e], "sorted keys", None), (got["sorted_values"], ref["sorted_values"][:e], "sorted values", None),
                      (got["tile_ranges"], ref["tile_ranges"], "tile ranges", None),
                      (got["rgba8"], ref["rgba8"], "image", None), (got["final_T"], ref["final_T"], "final T", np.float32), (got["n_contrib"], ref["n_contrib"], "n_contrib", None),
                      (pipe.bwd.getLossTextureView().read(np.float32), ref["loss_grad"].reshape(-1), "loss gradient", np.float32)]
            checks += [(acc[i], ref[k], k, None) for i, k in enumerate(("grad_means", "grad_conics", "grad_opacity", "grad_colors"))]
            checks += [(pipe.bwd.getGradientsBuffer().read(np.uint32).reshape(-1, 8)[:n], ref["gradients"], "packed gradients", np.float16),
                       (pipe.pc.gaussian_3d_buffer.read(np.uint32).reshape(-1, 6)[:n], ref_g, "re-packed Gaussians", np.float16),
                       (pipe.pc.sh_buffer.read(np.uint32).reshape(-1, 24)[:n], ref_sh, "re-packed SH", np.float16)]
            checks += [(state[k], ref_state[k], f"optimizer state {k}", np.float32) for k in ("opt_pos", "opt_rot", "opt_scale", "opt_opacity", "param_sh", "state_sh")]
            for a, b, what, fp in checks:
                d = differences(a, b, f"step {step}: {what}", fp)
                if d:
                    out.append(d)
        return out
    finally:
        if not keep:
            pipe.destroy()


def poisoned(cfg, field, value, every=7):
    g, sh, cam = harness.scene(cfg)
    gh = g.view(np.uint16).reshape(-1, 12).copy()
    shh = sh.view(np.uint16).reshape(-1, 48).copy()
    rows = np.arange(0, cfg.num_points, every)
    if field == "sh":
        shh[rows, 1] = value
    else:
        for c in dict(position=[0, 1, 2], x=[0], z=[2], opacity=[3], rotation=[4, 5, 6, 7], scale=[8, 9, 10], one_scale=[9])[field]:
            gh[rows, c] = value
    return gh.view(np.uint32).reshape(-1, 6), shh.view(np.uint32).reshape(-1, 24), cam


@pytest.mark.parametrize("value", [NAN16, INF16, 0xFE00, 0xFC00], ids=["nan", "inf", "-nan", "-inf"])
@pytest.mark.parametrize("field", ["position", "x", "z", "opacity", "rotation", "scale", "one_scale", "sh"])
def test_step_with_non_finite_gaussians(hip_device, orc, field, value):
    cfg = harness.small_config("c1", num_points=700, width=64, height=48)
    g, sh, cam = poisoned(cfg, field, value)
    rng = np.random.default_rng(5)
    target = rng.integers(0, 255, (cfg.height, cfg.width, 4), dtype=np.uint8)
    diffs = step_differences(orc, hip_device, cfg, g, sh, cam, target, steps=2)
    assert not diffs, "\n".join(diffs)


@pytest.mark.parametrize("field,value,every", [("position", NAN16, 2), ("x", NAN16, 2), ("z", 0xFE00, 3), ("position", INF16, 2), ("opacity", NAN16, 2), ("z", NAN16, 1)],
                         ids=["nan-positions", "nan-x", "-nan-z", "inf-positions", "nan-opacity", "all-nan-z"])
def test_a_pile_of_non_finite_gaussians_behind_a_dead_block(hip_device, orc, field, value, every):
    """The late regime's tile 0 in small: 1 300-4 000 Gaussians with a NaN (or infinite) field behind the real ones of their tile -- dozens of chunks whose
    records a dead block drops (raster.hip: EXACT, `dead`), and a tile list that `segment_sort` sorts through global memory (more than 2 048 entries:
    sort.hip, seg_pass_global).  A NaN position gives a NaN depth, which sorts last; a NaN opacity leaves the depth a number; an infinite position makes
    it an infinity or a NaN by the camera's row.  Either way: two steps equal to the oracle's."""
    cfg = harness.small_config("c1", num_points=4000, width=64, height=48)
    g, sh, cam = poisoned(cfg, field, value, every=every)
    rng = np.random.default_rng(6)
    target = rng.integers(0, 255, (cfg.height, cfg.width, 4), dtype=np.uint8)
    diffs = step_differences(orc, hip_device, cfg, g, sh, cam, target, steps=2)
    assert not diffs, "\n".join(diffs)


def _rows(rows):
    """(x, y, z, opacity_raw, log_sigma) -> packed Gaussians and SH (DC only)."""
    g = np.zeros((len(rows), 12), np.uint16)
    sh = np.zeros((len(rows), 48), np.uint16)
    f16 = lambda v: np.float16(v).view(np.uint16)   # noqa: E731
    for i, (x, y, z, o, ls) in enumerate(rows):
        g[i, 0:4] = [f16(x), f16(y), f16(z), f16(o)]
        g[i, 4] = f16(1.0)
        g[i, 8:11] = f16(ls)
        sh[i, 0:3] = [f16(1.5 - (i % 7) * 0.3), f16(0.3), f16(-0.7 + (i % 5) * 0.2)]
    return g.view(np.uint32).reshape(len(rows), 6), sh.view(np.uint32).reshape(len(rows), 24)


def _tile_centre(cfg, tx, ty, z):
    """World position (identity camera) that projects onto the centre of tile (tx, ty)."""
    px, py = tx * 16 + 8.0, ty * 16 + 8.0
    return (px - cfg.width / 2) * z / cfg.fy, (py - cfg.height / 2) * z / cfg.fy


def long_list_scene(kind, big=10_400, second=4_097):
    """Gaussians, SH, camera and config of a 96 x 64 scene with one tile of `big` entries, one of `second`, short lists elsewhere (see the test below)."""
    cfg = harness.small_config("c1", num_points=1, width=96, height=64, fy=90.0, sh_deg=0)
    rng = np.random.default_rng(17)
    rows = []
    for (tx, ty, count) in ((2, 1, big), (4, 2, second)):
        for i in range(count):
            z = 2.0 + 6.0 * rng.random()
            cx, cy = _tile_centre(cfg, tx, ty, z)
            if kind == "faint":   # ~40 px footprint at an opacity just above the 1/128 cull: covers the tile and its neighbours
                rows.append((cx + rng.uniform(-2, 2) * z / cfg.fy, cy + rng.uniform(-2, 2) * z / cfg.fy, z, rng.uniform(-4.7, -4.0), np.log(0.12 * z)))
            else:                 # footprints of a pixel or two (the 0.3 px dilation of the 2D covariance is all of it) at opacities of 1-3 %, anywhere in the tile
                rows.append((cx + rng.uniform(-7.5, 7.5) * z / cfg.fy, cy + rng.uniform(-7.5, 7.5) * z / cfg.fy, z, rng.uniform(-4.5, -3.5), np.log(0.001 * z)))
    for i in range(300):          # short lists everywhere
        z = 2.0 + 6.0 * rng.random()
        rows.append((rng.uniform(-0.5, 0.5) * z, rng.uniform(-0.33, 0.33) * z, z, rng.uniform(-1.0, 3.0), np.log(rng.uniform(0.01, 0.05) * z)))
    g, sh = _rows(rows)
    if kind == "pile-up":
        gh = g.view(np.uint16).reshape(-1, 12).copy()
        gh[60:big, 0:3] = NAN16       # everything of the first pile but its first 60 Gaussians
        gh[big:big + second:3, 8:11] = NAN16
        g = gh.view(np.uint32).reshape(-1, 6)
    cfg = harness.small_config("c1", num_points=len(rows), width=96, height=64, fy=90.0, sh_deg=0)
    return cfg, g, sh, synth.identity_camera(cfg), rng


@pytest.mark.parametrize("kind", ["sparse", "faint", "pile-up"])
def test_tile_lists_past_the_reference_cap(hip_device, orc, kind):
    """One tile with more than 10 000 entries (the reference stages at most 32 x 256 = 8 192 per tile, SURVEY Q3: lifted here), one with 4 097, short
    neighbours.  "sparse": thousands of small splats scattered over the tile -- a pixel sees few of them and never saturates, the block walks the
    whole list; "faint": splats that cover the whole tile at an alpha near the 1/255 cut -- the pixels saturate after some hundred; "pile-up": the
    late regime of a long run (profiles/r08g_long_list_stats.txt): ~60 real splats in front of thousands of Gaussians with NaN positions, which all
    land in tile 0 behind them.  Two training steps against the oracle, stage by stage."""
    cfg, g, sh, cam, rng = long_list_scene(kind)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    ref = orc.forward(g, sh, cam, st, ti)
    lens = np.bincount(ref["sorted_keys"][:ref["total_entries"]] >> 16, minlength=ti[2] + 2)[1:]
    assert lens.max() > 10_000 and (lens > 4_096).sum() >= (1 if kind == "pile-up" else 2), np.sort(lens)[-4:]
    if kind == "sparse":
        assert ref["n_contrib"].max() > 8_192, "a pixel whose last contributor lies beyond the reference's 8 192-entry cap"
    target = rng.integers(0, 255, (cfg.height, cfg.width, 4), dtype=np.uint8)
    diffs = step_differences(orc, hip_device, cfg, g, sh, cam, target, steps=2)
    assert not diffs, "\n".join(diffs)


def test_trainer_trajectory_with_non_finite_gaussians_equals_the_oracle_trainer(hip_device, orc):
    """The Trainer across two densify / prune rebuilds (metric views at half resolution, decisions, the rebuilt cloud and optimizer state) on a cloud
    in which every ninth Gaussian has a NaN or an infinity somewhere: the metric passes walk the poisoned tile lists too (K24 counts with the
    backward's alpha test), a NaN opacity is never pruned, a NaN scale never splits."""
    from oracle import oracle_trainer
    from webdgs_amd import ops
    from webdgs_amd.trainer import Trainer
    from test_gpu_trainer_oracle import _FixedViews, _dataset, STATE_KEYS
    dev = hip_device
    cfg = harness.small_config("c2", num_points=5000, width=128, height=96, s0=0.01)
    g, sh, _ = harness.scene(cfg)
    gh = g.view(np.uint16).reshape(-1, 12).copy()
    for k, (cols, value) in enumerate([([0, 1, 2], NAN16), ([3], NAN16), ([8, 9, 10], INF16), ([9], NAN16), ([4, 5, 6, 7], NAN16), ([2], 0xFC00)]):
        for c in cols:
            gh[k * 9 + 4::54, c] = value
    g = gh.view(np.uint32).reshape(-1, 6)
    clean, _, _ = harness.scene(cfg)
    cams, imgs, cameras, images = _dataset(dev, orc, cfg, clean, sh, 4)
    dens = dict(schedule=dict(enabled=True, warmupIterations=6, interval=5, stopIterations=12), metricViews=3, cloneThresholdCount=5,
                splitScaleThreshold=0.03, pruneOpacity=0.2, maxNewPointsPerStep=300)
    steps = 14
    rng = np.random.default_rng(9)
    train_views = [int(v) for v in rng.integers(0, 4, steps)]
    metric_views = {6: [2, 0, 3], 11: [1, 1, 2]}
    o = oracle_trainer.OracleTrainer(g, sh, cfg.sh_deg, list(cams), imgs, densify=dens)
    t = Trainer(dev, seed=0)
    t.setDensifyPruneConfig(dens)
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset(cameras, images)
    t.start()

    def compare(what):
        n = o.num_points
        assert t.getPointCount() == n, what
        out = [differences(t.pointCloud.gaussian_3d_buffer.read(np.uint32).reshape(-1, 6)[:n], o.g, f"{what}: gaussians", np.float16),
               differences(t.pointCloud.sh_buffer.read(np.uint32).reshape(-1, 24)[:n], o.sh, f"{what}: sh", np.float16)]
        bufs = t.optimizer.getStateBuffers()
        out += [differences(bufs[k].read(np.float32).reshape(-1, width)[:n], o.state[ok], f"{what}: optimizer state {k}", np.float32) for k, (ok, width) in STATE_KEYS.items()]
        out = [d for d in out if d]
        assert not out, "\n".join(out)

    try:
        for i in range(steps):
            it = i + 1
            draws = [train_views[i]] + metric_views.get(it, [])
            t._rng = _FixedViews(draws)
            o.step(train_views[i], metric_view_ids=metric_views.get(it))
            t.step()
            compare(f"after iteration {it}")
            if it in metric_views:
                assert o.last_densify["rebuilt"], "the cloud was rebuilt"
        nan_rows = np.isnan(o.g.view(np.float16).reshape(-1, 12)[:, :11].astype(np.float32)).any(axis=1).sum()
        assert nan_rows > 300, f"non-finite Gaussians survive the rebuilds and spread ({nan_rows})"
    finally:
        t.destroy()


def _pipeline_with_long_lists(threshold, items=0, rows=0, seen=None):
    def make(dev, cfg, g, sh, cam):
        p = harness.HipPipeline(dev, cfg, g, sh, cam)
        p.fwd.setLongLists(threshold, items, rows)
        if seen is not None:
            seen.append(p)
        return p
    return make


def test_long_lists_follow_a_growing_viewport_and_a_change_of_render_mode(hip_device, orc):
    """The per-tile marks of the long lists and the stamps of non-finite tiles are sized by the tile grid: a viewport that grows past what the pass was
    built for re-allocates both (csrc/api.hip: set_viewport).  With every tile above 48 entries a long one: the image and n_contrib after the change
    equal the oracle's, and so does a frame in point-cloud mode in between (that kernel knows nothing of the lists: the marks must not keep its blocks
    from being drawn) and the Gaussian frame after it."""
    from harness import assert_bits_equal
    cfg = harness.small_config("c1", num_points=6000, width=64, height=48, s0=0.004, fy=140.0)
    g, sh, cam = harness.scene(cfg)
    pipe = harness.HipPipeline(hip_device, cfg, g, sh, cam)
    try:
        pipe.fwd.setLongLists(48, 8192, 65536)
        pipe.forward()
        st = pipe.fwd.longListStats()
        assert st["blocksWanted"] >= 4 and st["itemsWanted"] <= st["maxItems"] and not st["stalled"], st
        big = harness.small_config("c1", num_points=6000, width=208, height=112, s0=0.004, fy=140.0)
        cam2 = synth.identity_camera(big)
        pipe.camera.write(cam2)
        pipe.fwd.setViewport(big.width, big.height)
        pipe.cfg = big
        rs, ti = synth.render_settings(big), synth.tile_info(big.width, big.height, 0)
        ref = orc.forward(g, sh, cam2, rs, ti)
        for what in ("after the viewport change", "again, after a frame in point-cloud mode"):
            pipe.forward()
            got = pipe.collect_forward()
            st = pipe.fwd.longListStats()
            assert st["blocksWanted"] >= 4 and st["itemsWanted"] <= st["maxItems"] and st["forwardQueue"] >= 2 * (st["itemsWanted"] + st["blocksWanted"]) and not st["stalled"], st
            assert_bits_equal(got["rgba8"], ref["rgba8"], "image " + what)
            assert_bits_equal(got["n_contrib"], ref["n_contrib"], "n_contrib " + what)
            assert not differences(got["final_T"], ref["final_T"], "final T " + what, np.float32)
            if what.startswith("after"):
                pipe.fwd.setRenderMode("pointcloud")
                pipe.forward()
                points = pipe.collect_forward()["rgba8"]
                ps = synth.render_settings(big, gaussian_mode=0.0)   # (point size: the pass's default, 3 px)
                assert_bits_equal(points, orc.forward(g, sh, cam2, ps, ti)["rgba8"], "point-cloud frame between two Gaussian frames with long lists")
                pipe.fwd.setRenderMode("gaussian")
    finally:
        pipe.destroy()


@pytest.mark.parametrize("case", ["every-tile", "no-item-slots", "no-rows", "off"])
def test_long_list_tasks_and_their_fallbacks(hip_device, orc, case):
    """The per-pixel lists of long tiles (csrc/longlist.h) at a threshold of 48 entries, so that most tiles of an ordinary scene take them: built by the
    sort, counted / scanned / filled / walked by tasks inside the rasterization kernel, walked backwards by the backward kernel's helpers.  With room
    for everything; with item slots for a few tiles only (the others stay with the wave-per-block walk: the no-room branch of the build); with a row
    pool that runs out (the blocks that find it empty are walked the plain way by their walk task); and switched off.  Two steps against the oracle."""
    cfg = harness.small_config("c1", num_points=9000, width=160, height=112, s0=0.004, fy=140.0)
    g, sh, cam = harness.scene(cfg)
    rng = np.random.default_rng(23)
    target = rng.integers(0, 255, (cfg.height, cfg.width, 4), dtype=np.uint8)
    threshold, items, rows = dict([("every-tile", (48, 8192, 65536)), ("no-item-slots", (48, 64, 65536)), ("no-rows", (48, 8192, 96)), ("off", (0, 0, 0))])[case]
    seen = []
    diffs = step_differences(orc, hip_device, cfg, g, sh, cam, target, steps=2, pipeline_factory=_pipeline_with_long_lists(threshold, items, rows, seen), keep=True)
    try:
        st = seen[0].fwd.longListStats()
        assert st["stalled"] == 0, st
        if case == "every-tile":
            assert st["blocksWanted"] >= 100 and st["itemsWanted"] <= st["maxItems"] and st["rowsWanted"] == st["rowsUsed"] > 0, st
            assert st["forwardQueue"] >= 2 * (st["itemsWanted"] + st["blocksWanted"]), st
        elif case == "no-item-slots":
            assert st["itemsWanted"] > st["maxItems"], st
        elif case == "no-rows":
            assert st["rowsUsed"] > st["maxRows"], ("blocks asked a pool that had run out", st)
        else:
            assert st["threshold"] == 0 and st["blocksWanted"] == 0, st
    finally:
        seen[0].destroy()
    assert not diffs, "\n".join(diffs)
