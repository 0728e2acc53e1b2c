"""GPU: operator-level behaviour through the C ABI -- primitives (scan, sort) on edge-case inputs, golden fixtures,
densify/prune parity, error behaviour mirrored from the reference, data-parallel step equivalence."""
import os

import numpy as np
import pytest

from webdgs_amd import _lib, ops, synth

import harness
from harness import assert_bits_equal

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- primitives
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4095, 4096, 4097, 100_003, 2_500_000])
def test_prefix_scanner_matches_cumsum(hip_device, n):
    rng = np.random.default_rng(n)
    x = rng.integers(0, 2049, n, dtype=np.uint32)
    sc = ops.get_prefix_scanner(n, hip_device)
    try:
        sc.input_buffer.write(x)
        sc.set_count(n)
        sc.scan(None)
        got = sc.output_buffer.read(np.uint32, count=n)
        ref = np.concatenate([[0], np.cumsum(x.astype(np.uint64))[:-1]]).astype(np.uint32)  # wrapping u32, > 2 097 152 elements allowed (Q1 lifted)
        assert_bits_equal(got, ref, f"exclusive scan n={n}")
    finally:
        sc.destroy()


def test_prefix_scanner_rejects_oversize_count(hip_device):
    sc = ops.get_prefix_scanner(1000, hip_device)
    try:
        with pytest.raises(_lib.CapacityError):
            sc.set_count(1001)
    finally:
        sc.destroy()


@pytest.mark.parametrize("n,bits", [(0, 32), (1, 32), (64, 32), (4096, 32), (4097, 32), (200_001, 32), (1_000_003, 29), (300_000, 8), (300_000, 17)])
def test_dynamic_sorter_is_a_stable_sort(hip_device, n, bits):
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 2**bits, max(n, 1), dtype=np.uint64).astype(np.uint32)[:n]
    if n > 100:
        keys[: n // 2] = keys[0]  # long runs of equal keys: stability matters
    vals = np.arange(n, dtype=np.uint32)[::-1].copy()
    stats = hip_device.createBuffer(16)
    stats.write(np.array([n, 0, 0, 0], np.uint32))
    so = ops.get_dynamic_sorter(max(n, 1) + 5000, hip_device, stats)
    try:
        if n:
            so.ping_pong[0]["sort_depths_buffer"].write(keys)
            so.ping_pong[0]["sort_indices_buffer"].write(vals)
        so.sort(None, key_bits=bits)
        out = so.ping_pong[so.final_out_index]
        gk, gv = out["sort_depths_buffer"].read(np.uint32, count=n), out["sort_indices_buffer"].read(np.uint32, count=n)
        order = np.argsort(keys, kind="stable")
        assert_bits_equal(gk, keys[order], "sorted keys")
        assert_bits_equal(gv, vals[order], "payload follows a STABLE sort")
    finally:
        so.destroy()


# ----------------------------------------------------------------------------- golden fixtures through the HIP path
def test_hip_matches_golden_train_step(hip_device):
    d = np.load(os.path.join(HERE, "golden", "train_step.npz"))
    from tests_golden_cfg import GOLDEN_CFG as cfg
    pipe = harness.HipPipeline(hip_device, cfg, d["in_gaussians"], d["in_sh"], d["camera"])
    try:
        tbuf = hip_device.bufferFrom(d["target"])
        pipe.train_step(tbuf)
        hip_device.synchronize()
        got = pipe.collect_forward()
        vis = d["tile_counts"] > 0
        assert_bits_equal(got["tile_counts"], d["tile_counts"], "tile_counts")
        assert_bits_equal(got["splats"][vis], d["splats"][vis], "splats")
        assert_bits_equal(got["sorted_keys"], d["sorted_keys"], "sorted keys")
        assert_bits_equal(got["sorted_values"], d["sorted_values"], "sorted values")
        assert_bits_equal(got["tile_ranges"], d["tile_ranges"], "tile ranges")
        for k in ("rgba8", "final_T", "n_contrib"):
            assert_bits_equal(got[k], d[k], k)
        n = cfg.num_points
        assert_bits_equal(pipe.bwd.getGradientsBuffer().read(np.uint32).reshape(-1, 8)[:n], d["gradients"], "gradients")
        st = pipe.read_state()
        for k in st:
            assert_bits_equal(st[k], d["state1_" + k], "state " + k)
        assert_bits_equal(pipe.pc.gaussian_3d_buffer.read(np.uint32).reshape(-1, 6), d["out_gaussians"], "re-packed gaussians")
        assert_bits_equal(pipe.pc.sh_buffer.read(np.uint32).reshape(-1, 24), d["out_sh"], "re-packed sh")
    finally:
        pipe.destroy()


def test_hip_matches_golden_densify(hip_device):
    d = np.load(os.path.join(HERE, "golden", "densify.npz"))
    from tests_golden_cfg import GOLDEN_CFG as cfg
    dev = hip_device
    n = cfg.num_points
    mw, mh = cfg.width // 2, cfg.height // 2
    pc = ops.createPointCloud(dev, d["in_gaussians"], d["in_sh"], cfg.sh_deg)
    cam = dev.bufferFrom(d["metrics_camera"])
    fwd = ops.TiledForwardPass(dev, pc, cam, dict(viewportWidth=mw, viewportHeight=mh, renderMode="gaussian"))
    rast = ops.TiledRasterizer(dict(device=dev, forwardPass=fwd, format="rgba8unorm"))
    mp = ops.TiledBackwardPass(dev, pc, dict(viewportWidth=mw, viewportHeight=mh, trainingConfig={}))
    dp = ops.DensifyPrunePass(dev, dict(strategy="gpu_rebuild", numViews=1, cloneThreshold=6, splitThreshold=0.05, pruneThreshold=0.15,
                                        maxNewPointsPerStep=40, maxBufferBytes=128 * 1024 * 1024))
    try:
        # GT down-sample (K31)
        golden_target = np.load(os.path.join(HERE, "golden", "train_step.npz"))["target"]
        src = dev.bufferFrom(golden_target)
        small = dev.createBuffer(4 * mw * mh)
        ops.downsampleRGBA8(dev, src, cfg.width, cfg.height, small, mw, mh)
        assert_bits_equal(small.read(np.uint8).reshape(mh, mw, 4), d["gt_small"], "bilinear down-sample")
        # metric view
        fwd.encode(None)
        rast.encode(None, mw, mh)
        assert_bits_equal(rast.getOutputTextureView().read(np.uint8).reshape(mh, mw, 4), d["metric_rgba8"], "metric render")
        mp.getMetricCountsBuffer().clear()
        mp.computeMetricMap(None, rast.getOutputTextureView(), small, dict(threshold=0.5))
        assert_bits_equal(mp.getMetricMinMaxBuffer().read(np.uint32), d["metric_minmax"], "metric min/max")
        assert_bits_equal(mp.getMetricMapTextureView().read(np.uint32).reshape(mh, mw), d["metric_flags"], "metric flags")
        mp.computeMetricCounts(None, dict(splatBuffer=fwd.getResources()["splatBuffer"], tileOffsetsBuffer=rast.getTileOffsetsBuffer(),
                                          tileIndicesBuffer=fwd.getSortedIndicesBuffer(), nContribTexture=rast.getNContribTextureView()), dict(clear=False))
        assert_bits_equal(mp.getMetricCountsBuffer().read(np.uint32)[:n], d["metric_counts"], "metric counts")
        mp.normalizeMetricCounts(None, dict(divisor=1))
        # decide / cap / scan / total
        prep = dp.encodePrepare(None, dict(pointCloud=pc, metricCountsBuffer=mp.getMetricCountsBuffer()))
        assert prep["maxOutPoints"] == int(d["max_out"][0])
        total = dp.readTotal()
        assert total == int(d["total"][0])
        assert_bits_equal(prep["actionBuffer"].read(np.uint32)[:n], d["actions"], "actions")
        assert_bits_equal(prep["outCountBuffer"].read(np.uint32)[:n], d["out_counts"], "out counts")
        assert_bits_equal(prep["outOffsetBuffer"].read(np.uint32)[:n], d["out_offsets"], "out offsets")
        # scatter
        out_n = min(total, prep["maxOutPoints"])
        names = dict(optPosBuffer="opt_pos", optRotBuffer="opt_rot", optScaleBuffer="opt_scale", optOpacityBuffer="opt_opacity", paramSH="param_sh", stateSH="state_sh")
        in_state = {k: dev.bufferFrom(d["in_" + v]) for k, v in names.items()}
        out_pc = ops.allocatePointCloudLike(dev, pc, dict(numPoints=out_n))
        out_state = ops.allocateOptimizerStateBuffers(dev, out_n)
        dp.encodeScatter(None, dict(pointCloud=pc, optimizerState=in_state, outOffsetBuffer=prep["outOffsetBuffer"], outNumPoints=out_n, resetNewOptimizerState=True),
                         dict(outPointCloud=out_pc, outOptimizerState=out_state))
        dev.synchronize()
        assert_bits_equal(out_pc.gaussian_3d_buffer.read(np.uint32).reshape(-1, 6), d["out_gaussians"], "scattered gaussians")
        assert_bits_equal(out_pc.sh_buffer.read(np.uint32).reshape(-1, 24), d["out_sh"], "scattered sh")
        for k, v in names.items():
            ref = d["out_" + v]
            assert_bits_equal(out_state[k].read(np.float32).reshape(ref.shape), ref, "scattered " + v)
        with pytest.raises(_lib.WdgsError):  # densify-prune.ts:478-480
            dp.encodeScatter(None, dict(pointCloud=pc, optimizerState=in_state, outNumPoints=out_n + 1), dict(outPointCloud=out_pc, outOptimizerState=out_state))
        with pytest.raises(NotImplementedError):
            dp.applyActions()
    finally:
        for o in (dp, mp, rast, fwd):
            o.destroy()


# ----------------------------------------------------------------------------- error behaviour
def test_texture_getters_throw_before_first_encode(hip_device):
    cfg = harness.small_config("c1", num_points=100, width=64, height=64)
    g, sh, cam = harness.scene(cfg)
    pipe = harness.HipPipeline(hip_device, cfg, g, sh, cam)
    try:
        for getter in (pipe.rast.getOutputTextureView, pipe.rast.getAlphaTextureView, pipe.rast.getNContribTextureView, pipe.rast.getTileOffsetsBuffer):
            with pytest.raises(_lib.StateError):  # tiled-rasterizer.ts:308-330
                getter()
        with pytest.raises(_lib.StateError):
            pipe.rast.encode(None, cfg.width, cfg.height)  # forward pass not encoded yet
        pipe.forward()
        assert pipe.rast.getOutputTextureView().size == 4 * 64 * 64
        pipe.fwd.destroy(); pipe.fwd.destroy()  # double destroy is a no-op (tiled-forward-pass.ts:518-533)
        pipe.fwd = None
    finally:
        pipe.destroy()


def test_tile_entry_overflow_is_a_hard_error(hip_device):
    """The reference overruns its buffers silently when E > maxTileEntries; here it is reported."""
    cfg = harness.small_config("c1", num_points=20_000)
    g, sh, cam = harness.scene(cfg)
    pipe = harness.HipPipeline(hip_device, cfg, g, sh, cam, max_tile_entries=4096)
    try:
        pipe.fwd.encode(None)
        with pytest.raises(_lib.CapacityError):
            pipe.fwd.check()
        hip_device.synchronize()  # the overflow word is consumed by the check that reported it
        pipe.fwd.encode(None)
        with pytest.raises(_lib.CapacityError):
            hip_device.synchronize()  # the deferred check of queue.onSubmittedWorkDone reports it too
        hip_device.synchronize()
        # sticky across encodes: a later encode that fits must not hide an earlier one that overflowed (multi-view steps)
        pipe.fwd.encode(None)
        small = harness.HipPipeline(hip_device, harness.small_config("c1", num_points=50), g[:50], sh[:50], cam, max_tile_entries=4096)
        small.fwd.encode(None)
        with pytest.raises(_lib.CapacityError):
            hip_device.synchronize()
        small.destroy()
    finally:
        pipe.destroy()
    hip_device.synchronize()


def test_viewport_change_follows_the_forward_pass(hip_device, orc):
    """setViewport resizes the whole grid (fixes SURVEY Q19: the reference's rasterizer keeps a stale 1x1 grid)."""
    cfg = harness.small_config("c1", num_points=5000, width=64, height=48)
    g, sh, cam = harness.scene(cfg)
    pipe = harness.HipPipeline(hip_device, cfg, g, sh, cam)
    try:
        pipe.forward()
        big = harness.small_config("c1", num_points=5000, width=208, height=112)
        cam2 = synth.identity_camera(big)
        pipe.camera.write(cam2)
        pipe.fwd.setViewport(big.width, big.height)
        pipe.cfg = big
        pipe.forward()
        got = pipe.collect_forward()
        ref = orc.forward(g, sh, cam2, synth.render_settings(big), synth.tile_info(big.width, big.height, 0))
        assert_bits_equal(got["rgba8"], ref["rgba8"], "image after viewport change")
        assert_bits_equal(got["n_contrib"], ref["n_contrib"], "n_contrib after viewport change")
    finally:
        pipe.destroy()


def test_compat_caps_reproduces_reference_capacity(hip_device, orc):
    """compatCaps=True: maxTileEntries = min(30N, 32Mi, 2097152) rounded to 3840 and at most 32 x 256 splats per tile (Q2, Q3)."""
    cfg = harness.small_config("c1", num_points=3000, width=64, height=64, s0=0.08)
    g, sh, cam = harness.scene(cfg)
    pipe = harness.HipPipeline(hip_device, cfg, g, sh, cam, compat_caps=True)
    try:
        assert pipe.fwd.getResources()["maxTileEntries"] == 94208  # 90 000 -> x3840 = 92 160 -> sorter partitions of 4096
        pipe.forward()
        got = pipe.collect_forward()
        st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
        ref = orc.forward(g, sh, cam, st, ti, max_batches=32)
        assert_bits_equal(got["rgba8"], ref["rgba8"], "image under the 8192-per-tile cap")
        assert_bits_equal(got["n_contrib"], ref["n_contrib"], "n_contrib under the cap")
    finally:
        pipe.destroy()


def test_point_cloud_render_mode(hip_device, orc):
    cfg = harness.small_config("c1", num_points=4000, width=96, height=80)
    g, sh, cam = harness.scene(cfg)
    pipe = harness.HipPipeline(hip_device, cfg, g, sh, cam)
    try:
        pipe.fwd.setRenderMode("pointcloud")
        pipe.fwd.setPointSize(2.0)
        pipe.forward()
        got = pipe.collect_forward()
        st = synth.render_settings(cfg, gaussian_mode=0.0)
        st[4] = 2.0
        ref = orc.forward(g, sh, cam, st, synth.tile_info(cfg.width, cfg.height, 0))
        assert_bits_equal(got["rgba8"], ref["rgba8"], "point-cloud mode image")
        assert_bits_equal(got["n_contrib"], ref["n_contrib"], "point-cloud mode n_contrib")
    finally:
        pipe.destroy()


# ----------------------------------------------------------------------------- data-parallel step
def test_dp_step_with_one_view_equals_reference_step(hip_device):
    """accumulate -> stepF32 on one view must equal the fp16-gradient step bit for bit (batch semantics reduce to the reference)."""
    cfg = harness.small_config("c2", num_points=8000, width=160, height=120)
    g, sh, cam = harness.scene(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    dev = hip_device
    a = harness.HipPipeline(dev, cfg, g, sh, cam)
    b = harness.HipPipeline(dev, cfg, g, sh, cam)
    tp = harness.HipPipeline(dev, cfg, tg, tsh, cam)
    try:
        tp.forward()
        target = dev.bufferFrom(tp.rast.getOutputTextureView().read(np.uint8))
        a.train_step(target)
        b.opt = ops.Optimizer(dev, b.pc)
        b.fwd.encode(None); b.rast.encode(None, cfg.width, cfg.height)
        b.bwd.encode(None, b.rast.getOutputTextureView(), target, b.backward_resources())
        n = cfg.num_points
        acc, vis = dev.createBuffer(4 * 14 * n), dev.createBuffer(4 * n)
        ops.accumulateGradients(dev, n, b.bwd.getGradientsBuffer(), b.fwd.getResources()["tileCountsBuffer"], acc, vis)
        b.opt.stepF32(None, b.pc, acc, vis)
        dev.synchronize()
        sa, sb = a.read_state(), b.read_state()
        for k in sa:
            assert_bits_equal(sa[k], sb[k], "DP state " + k)
        assert_bits_equal(a.pc.gaussian_3d_buffer.read(np.uint32), b.pc.gaussian_3d_buffer.read(np.uint32), "DP gaussians")
        assert (vis.read(np.uint32) > 0).sum() == int(a.fwd.check()[1])
    finally:
        for p in (a, b, tp):
            p.destroy()


def test_geometry_backward_that_accumulates_equals_the_separate_launches(hip_device):
    """A batched step's K17 (encodeRaster + encodeGeometry(accumulate=...)) adds the view's gradient to the step's fp32 block and folds the
    overflow word into the guard itself.  Three views from three cameras, an overflowing one among them: block, visibility counts,
    guard word and the packed gradients must equal encode + storeGradients / accumulateGradients + guardAccumulate bit for bit."""
    dev = hip_device
    cfg = harness.small_config("c2", num_points=8000, width=160, height=120)
    g, sh, _ = harness.scene(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    cams = synth.circle_cameras(cfg, 3)
    n = cfg.num_points
    results = []
    for fused in (False, True):
        sums, vis, guard = dev.createBuffer(4 * 14 * n), dev.createBuffer(4 * n), dev.createBuffer(16)
        guard.write(np.array([7, 0, 0, 0], np.uint32))  # a stale word: the first view must overwrite it
        grads = []
        for k, cam in enumerate(cams):
            tp = harness.HipPipeline(dev, cfg, tg, tsh, cam)
            tp.forward()
            target = dev.bufferFrom(tp.rast.getOutputTextureView().read(np.uint8))
            tp.destroy()
            p = harness.HipPipeline(dev, cfg, g, sh, cam, max_tile_entries=4096 if k == 1 else 0)  # view 1 overflows its tile list
            try:
                p.fwd.encode(None); p.rast.encode(None, cfg.width, cfg.height)
                res, stats, counts = p.backward_resources(), p.fwd.getStatsBuffer(), p.fwd.getResources()["tileCountsBuffer"]
                if fused:
                    p.bwd.encodeRaster(None, p.rast.getOutputTextureView(), target, res)
                    p.bwd.encodeGeometry(None, p.camera, dict(sums=sums, visible=vis, tileCounts=counts, guard=guard, stats=stats, first=(k == 0)))
                else:
                    p.bwd.encode(None, p.rast.getOutputTextureView(), target, res)
                    (ops.storeGradients if k == 0 else ops.accumulateGradients)(dev, n, p.bwd.getGradientsBuffer(), counts, sums, vis)
                    ops.guardAccumulate(dev, guard, stats, 8, overwrite=(k == 0))
                try:
                    dev.synchronize()
                except _lib.CapacityError:
                    assert k == 1
                grads.append(p.bwd.getGradientsBuffer().read(np.uint32))
            finally:
                p.destroy()
        results.append(dict(sums=sums.read(np.uint32), vis=vis.read(np.uint32), guard=guard.read(np.uint32, count=1), grads=np.stack(grads)))
    a, b = results
    assert int(a["guard"][0]) == 1 and a["vis"].max() == 3 and a["sums"].any()
    for k in a:
        assert_bits_equal(a[k], b[k], f"fused K17 vs separate launches: {k}")


def test_resized_passes_behave_like_fresh_ones(hip_device):
    """wdgs_tiled_forward_resize / wdgs_tiled_backward_resize (what applyPointCloudSwap uses instead of destroy + construct): a pass
    taken from 5 000 to 9 000 Gaussians (re-allocates, with headroom), on to 9 700 (fits the headroom) and down to 3 000 (fits) must
    produce, stage by stage, the bits of a pass constructed for that cloud -- forward, composite, gradients, accumulators."""
    dev = hip_device
    big_cfg = harness.small_config("c1", num_points=9_700, width=160, height=128)
    g, sh, cam = harness.scene(big_cfg)
    target = None
    pipe = harness.HipPipeline(dev, harness.small_config("c1", num_points=5_000, width=160, height=128), g[:5_000], sh[:5_000], cam)
    try:
        pipe.forward()
        for n in (9_000, 9_700, 3_000):
            cfg = harness.small_config("c1", num_points=n, width=160, height=128)
            pc = ops.createPointCloud(dev, g[:n], sh[:n], cfg.sh_deg)
            assert pipe.fwd.setPointCloud(pc) and pipe.bwd.setPointCloud(pc)
            pipe.pc, pipe.cfg = pc, cfg
            fresh = harness.HipPipeline(dev, cfg, g[:n], sh[:n], cam)
            try:
                for p in (pipe, fresh):
                    p.forward()
                a, b = pipe.collect_forward(), fresh.collect_forward()
                assert a["total_entries"] == b["total_entries"] > 0
                assert pipe.fwd.getResources()["maxTileEntries"] == fresh.fwd.getResources()["maxTileEntries"]
                for k in a:
                    harness.assert_bits_equal(np.asarray(a[k]), np.asarray(b[k]), f"{n} Gaussians, resized vs fresh pass: {k}")
                if target is None:
                    target = dev.bufferFrom(np.full(cfg.width * cfg.height, 0xFF406080, np.uint32))
                for p in (pipe, fresh):
                    p.bwd.encode(None, p.rast.getOutputTextureView(), target, p.backward_resources())
                dev.synchronize()
                harness.assert_bits_equal(pipe.bwd.getGradientsBuffer().read(np.uint32), fresh.bwd.getGradientsBuffer().read(np.uint32), f"{n}: gradients")
                harness.assert_bits_equal(pipe.bwd.getAccumulatorsBuffer().read(np.int32), fresh.bwd.getAccumulatorsBuffer().read(np.int32), f"{n}: accumulators")
            finally:
                fresh.destroy()
        other_deg = ops.createPointCloud(dev, g[:100], sh[:100], 3 if big_cfg.sh_deg != 3 else 1)
        assert not pipe.fwd.setPointCloud(other_deg), "another SH degree: the pass cannot follow and says so"
    finally:
        pipe.destroy()


def test_a_projection_is_consumed_once(hip_device, orc):
    """``projectViews`` + ``encodeProjected`` (the view-batched K1): the scan works in place on K1's workgroup sums, so the rest of a pass
    can run once per projection -- a second ``encodeProjected``, or one after a plain ``encode`` has overwritten the projection, is refused
    (it would index with offsets scanned twice), and ``isProjected`` says which state the pass is in.  The projected pass equals the plain one."""
    cfg = harness.small_config("c1", num_points=4000)
    g, sh, cam = harness.scene(cfg)
    pipe = harness.HipPipeline(hip_device, cfg, g, sh, cam)
    try:
        fw = pipe.fwd
        assert not fw.isProjected()
        ops.projectViews([fw], [pipe.camera], pipe.pc)
        assert fw.isProjected()
        fw.encodeProjected(None)
        assert not fw.isProjected()
        with pytest.raises(ops.StateError):
            fw.encodeProjected(None)
        st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
        ref = orc.forward(g, sh, cam, st, ti)
        e = ref["total_entries"]
        harness.assert_bits_equal(fw.getSortedIndicesBuffer().read(np.uint32)[:e], ref["sorted_values"][:e], "sorted indices of the projected pass")
        ops.projectViews([fw], [pipe.camera], pipe.pc)
        fw.encode(None)   # K1 again: the projection is gone
        assert not fw.isProjected()
        with pytest.raises(ops.StateError):
            fw.encodeProjected(None)
        harness.assert_bits_equal(fw.getSortedIndicesBuffer().read(np.uint32)[:e], ref["sorted_values"][:e], "sorted indices after the plain encode")
        # A RECORDED encodeProjected consumes a projection on every replay: the check travels with the command buffer to queue.submit (a replay
        # makes no encode call; without the check a recording that starts at the scan runs on sums that were already scanned in place -- the
        # device fault of round 3's project-ahead experiment)
        dev = hip_device
        with dev.createCommandEncoder("rest of a projected pass", record=True) as enc:
            fw.encodeProjected(enc)   # recording needs no projection: nothing runs now
            cmd = enc.finish()
        try:
            with pytest.raises(ops.StateError, match="holds no projection"):
                dev.queue.submit([cmd])
            ops.projectViews([fw], [pipe.camera], pipe.pc)
            dev.queue.submit([cmd])
            assert not fw.isProjected()
            harness.assert_bits_equal(fw.getSortedIndicesBuffer().read(np.uint32)[:e], ref["sorted_values"][:e], "sorted indices of a replayed projected pass")
            with pytest.raises(ops.StateError, match="holds no projection"):
                dev.queue.submit([cmd])   # the same projection twice
            ops.projectViews([fw], [pipe.camera], pipe.pc)
            fw.encode(None)               # a foreign encode of the pass between the projection and the replay (what a host preview between steps does)
            with pytest.raises(ops.StateError, match="holds no projection"):
                dev.queue.submit([cmd])
            ops.projectViews([fw], [pipe.camera], pipe.pc)
            dev.queue.submit([cmd])
            harness.assert_bits_equal(fw.getSortedIndicesBuffer().read(np.uint32)[:e], ref["sorted_values"][:e], "sorted indices after the refused submits")
        finally:
            dev.synchronize()
            cmd.destroy()
    finally:
        pipe.destroy()
