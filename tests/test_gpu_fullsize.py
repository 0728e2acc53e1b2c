"""GPU: BASELINE-size workloads (1 M Gaussians, 1920x1080, SH 3) checked through size-independent properties, because
the CPU oracle needs seconds per stage at this size: sortedness + stability, permutation, ranges partition, count
identities, image/gradient checksums equal across runs and across submission modes, and one oracle spot check of K1."""
import hashlib
import subprocess
import os

import numpy as np
import pytest

from webdgs_amd import ops, synth

import harness

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8)).hexdigest()


@pytest.fixture(scope="module")
def c3_forward(hip_device):
    cfg = synth.CONFIGS["c3"]
    g, sh = synth.make_gaussians(cfg)
    cam = synth.circle_cameras(cfg, 8)[3]
    pipe = harness.HipPipeline(hip_device, cfg, g, sh, cam)
    pipe.forward()
    out = pipe.collect_forward()
    yield cfg, g, sh, cam, pipe, out
    pipe.destroy()


def test_c3_forward_structure(c3_forward, orc):
    cfg, g, sh, cam, pipe, out = c3_forward
    counts, offsets, keys, vals, ranges = out["tile_counts"], out["tile_offsets"], out["sorted_keys"], out["sorted_values"], out["tile_ranges"]
    e = out["total_entries"]
    assert e > 3_000_000 and int(out["stats"][2]) == 0
    assert int(counts.astype(np.uint64).sum()) == e
    assert np.array_equal(offsets[1:], np.cumsum(counts.astype(np.uint64))[:-1].astype(np.uint32))
    assert int(out["stats"][1]) == int((counts > 0).sum())
    assert np.all(keys[:-1] <= keys[1:]), "sorted"
    same = keys[:-1] == keys[1:]
    assert np.all(vals[:-1][same] < vals[1:][same]), "stable: ties in ascending Gaussian index"
    assert np.array_equal(np.bincount(vals, minlength=cfg.num_points).astype(np.uint32), counts), "permutation of the emitted entries"
    assert np.array_equal(keys & 0xFFFF, out["depths"][vals] >> 16)
    tile_of = (keys >> 16).astype(np.int64) - 1
    first = np.full(cfg.total_tiles, 0xFFFFFFFF, np.uint32)
    idx = np.flatnonzero(np.concatenate([[True], tile_of[1:] != tile_of[:-1]]))
    first[tile_of[idx]] = idx.astype(np.uint32)
    assert np.array_equal(ranges[:-1], first) and ranges[-1] == e
    # K1 spot check against the oracle on the first 50k Gaussians (same camera)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    rs, rd, rc, _ = orc.project_count(g[:50_000].copy(), sh[:50_000].copy(), cam, st, ti)
    assert np.array_equal(rc, counts[:50_000])
    vis = rc > 0
    assert np.array_equal(rs[vis], out["splats"][:50_000][vis]) and np.array_equal(rd[vis], out["depths"][:50_000][vis])
    # image sanity: n_contrib never exceeds the tile's entry count; T in [0,1]
    assert out["final_T"].min() >= 0.0 and out["final_T"].max() <= 1.0
    per_tile = np.bincount(tile_of, minlength=cfg.total_tiles)
    tile_max = np.zeros(cfg.total_tiles, np.int64)
    yy, xx = np.mgrid[0:cfg.height, 0:cfg.width]
    np.maximum.at(tile_max, (yy // 16) * cfg.tiles_x + xx // 16, out["n_contrib"].astype(np.int64))
    assert np.all(tile_max <= per_tile)


def test_c3_train_step_is_deterministic_and_mode_independent(hip_device):
    """Two eager runs and one recorded (HIP graph) run of 3 training steps give identical bits: integer accumulation makes the
    backward order-free, and the recorded command buffer replays the same kernels."""
    cfg = synth.CONFIGS["c3"]
    g, sh = synth.make_gaussians(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    cams = synth.circle_cameras(cfg, 2)
    from webdgs_amd.trainer import Trainer
    dev = hip_device
    tp = harness.HipPipeline(dev, cfg, tg, tsh, cams[0])
    images, cameras = [], []
    for i in range(2):
        tp.camera.write(cams[i]); tp.forward()
        images.append(dict(texture=dev.bufferFrom(tp.rast.getOutputTextureView().read(np.uint8)), width=cfg.width, height=cfg.height))
        cameras.append(dict(camera=cams[i], width=cfg.width, height=cfg.height))
    tp.destroy()
    digests = []
    for use_cb in (False, False, True):
        t = Trainer(dev, seed=5, use_command_buffers=use_cb)
        t.keep_gradients = True   # (the fused step fills the packed-gradient buffer only on request; its digest is compared below)
        t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg)); t.setDataset(cameras, images); t.start()
        for _ in range(4):
            t.step()
        digests.append((_digest(t.pointCloud.gaussian_3d_buffer.read(np.uint32)), _digest(t.optimizer.getStateBuffers()["optPosBuffer"].read(np.uint32)),
                        _digest(t.backwardPass.getGradientsBuffer().read(np.uint32))))
        changed = (t.pointCloud.gaussian_3d_buffer.read(np.uint32).reshape(-1, 6) != g).any(axis=1).sum()
        assert changed > 50_000  # only Gaussians inside some pixel's first n_contrib entries receive a gradient (~180 k here)
        t.applyPointCloudSwap(dict(pointCloud=ops.createPointCloud(dev, g[:10], sh[:10], cfg.sh_deg)))  # frees the big buffers
    assert digests[0] == digests[1] == digests[2]


def test_napi_addon_renders_on_the_gpu():
    """The N-API binding (bindings/napi) drives the same library from node: tiny forward + composite."""
    addon = os.path.join(ROOT, "bindings", "napi", "webdgs_napi.node")
    if not os.path.exists(addon):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "bindings", "napi")])
    out = subprocess.run(["node", os.path.join(ROOT, "bindings", "napi", "smoke.js"), "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "napi gpu smoke: E=" in out.stdout and "visible=64" in out.stdout


def test_c3_full_size_train_step_equals_the_oracle(hip_device, orc):
    """BASELINE config c3 at full size -- 1 M Gaussians, 1920x1080, SH 3, ~6 M tile entries -- one complete training step
    (project, scan, emit, sort, ranges, composite, loss, backward raster, geometry backward, Adam, re-pack) compared with the
    oracle bit for bit.  (The oracle needs a few seconds per step on the GPU box's host cores.)"""
    cfg = synth.CONFIGS["c3"]
    g, sh = synth.make_gaussians(cfg)
    cam = synth.circle_cameras(cfg, 8)[5]
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    tg, tsh = synth.make_target_scene(g, sh)
    tp = harness.HipPipeline(hip_device, cfg, tg, tsh, cam)
    tp.forward()
    target = tp.rast.getOutputTextureView().read(np.uint8).reshape(cfg.height, cfg.width, 4).copy()
    tp.destroy()
    ref_g, ref_sh = g.copy(), sh.copy()
    ref_state = orc.unpack(ref_g, ref_sh)
    ref = orc.train_step(ref_g, ref_sh, ref_state, cam, st, ti, target)
    assert ref["total_entries"] > 5_000_000
    pipe = harness.HipPipeline(hip_device, cfg, g, sh, cam)
    try:
        pipe.train_step(hip_device.bufferFrom(target))
        hip_device.synchronize()
        got = pipe.collect_forward()
        n = cfg.num_points
        assert int(got["stats"][0]) == ref["total_entries"]
        harness.assert_bits_equal(got["sorted_values"], ref["sorted_values"][:ref["total_entries"]], "c3 sort order")
        harness.assert_bits_equal(got["rgba8"], ref["rgba8"], "c3 image")
        harness.assert_bits_equal(got["final_T"], ref["final_T"], "c3 final T")
        harness.assert_bits_equal(got["n_contrib"], ref["n_contrib"], "c3 n_contrib")
        harness.assert_bits_equal(pipe.bwd.getLossTextureView().read(np.float32).reshape(cfg.height, cfg.width, 4), ref["loss_grad"], "c3 loss gradient image")
        gm, gc, go, gcol = harness.acc_to_reference_layout(pipe.bwd.getAccumulatorsBuffer().read(np.int32), n)
        harness.assert_bits_equal(gm, ref["grad_means"], "c3 mean accumulators")
        harness.assert_bits_equal(gc, ref["grad_conics"], "c3 conic accumulators")
        harness.assert_bits_equal(go, ref["grad_opacity"], "c3 opacity accumulators")
        harness.assert_bits_equal(gcol, ref["grad_colors"], "c3 colour accumulators")
        harness.assert_bits_equal(pipe.bwd.getGradientsBuffer().read(np.uint32).reshape(-1, 8)[:n], ref["gradients"], "c3 packed gradients")
        state = pipe.read_state()
        for k in ref_state:
            harness.assert_bits_equal(state[k], ref_state[k], "c3 optimizer state " + k)
        harness.assert_bits_equal(pipe.pc.gaussian_3d_buffer.read(np.uint32).reshape(-1, 6), ref_g, "c3 re-packed Gaussians")
        harness.assert_bits_equal(pipe.pc.sh_buffer.read(np.uint32).reshape(-1, 24), ref_sh, "c3 re-packed SH")
    finally:
        pipe.destroy()


def test_c3_full_size_densify_equals_the_oracle(hip_device, orc):
    """The densify/prune rebuild at BASELINE size: half-resolution metric view of 1 M Gaussians, metric map + counts,
    decide / cap (maxNewPointsPerStep = 50 000 as in config c5) / scans / total, and the fused scatter of the point cloud and all
    optimizer state -- every array equal to the oracle's."""
    cfg = synth.CONFIGS["c3"]
    g, sh = synth.make_gaussians(cfg)
    cam = synth.circle_cameras(cfg, 8)[2]
    n = cfg.num_points
    mw, mh = cfg.width // 2, cfg.height // 2
    mst, mti = synth.render_settings(cfg, mw, mh), synth.tile_info(mw, mh, 0)
    mcam = synth.camera_block(cam[0:16].reshape(4, 4).T.astype(np.float64), mw, mh, cfg.fy * mh / cfg.height)
    tg, tsh = synth.make_target_scene(g, sh)
    # oracle side
    target = orc.forward(tg, tsh, cam, synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0))["rgba8"]
    mfw = orc.forward(g, sh, mcam, mst, mti)
    gt_small = orc.downsample_bilinear(target, mw, mh)
    err, mm, flags = orc.metric_map(mfw["rgba8"], gt_small, 0.5)
    counts = np.zeros(n, np.uint32)
    bst = mst.copy(); bst[5] = 0.0
    cap = max(int(mfw["total_entries"]), 1)
    orc.metric_count(bst, mfw["tile_ranges"], mfw["sorted_values"][:cap].copy(), cap, mfw["splats"], flags, mfw["n_contrib"], counts)
    counts_raw = counts.copy()
    orc.metric_normalize(counts, 1)
    max_out = n + 50_000
    prep = orc.densify_prepare(g, counts, max_out, clone_threshold=40, prune_opacity=0.12, split_scale=0.012)
    a = prep["actions"]
    assert min((a == i).sum() for i in range(4)) > 100, "the case must exercise keep, clone, split and prune"
    out_n = min(prep["total"], max_out)
    state = orc.unpack(g.copy(), sh.copy())
    og, osh, ost = orc.densify_scatter(g, sh, state, prep, out_n)
    # HIP side
    dev = hip_device
    pc = ops.createPointCloud(dev, g, sh, cfg.sh_deg)
    fwd = ops.TiledForwardPass(dev, pc, dev.bufferFrom(mcam), dict(viewportWidth=mw, viewportHeight=mh, renderMode="gaussian"))
    rast = ops.TiledRasterizer(dict(device=dev, forwardPass=fwd, format="rgba8unorm"))
    mp = ops.TiledBackwardPass(dev, pc, dict(viewportWidth=mw, viewportHeight=mh, trainingConfig={}))
    dp = ops.DensifyPrunePass(dev, dict(strategy="gpu_rebuild", numViews=1, cloneThreshold=40, splitThreshold=0.012, pruneThreshold=0.12,
                                        maxNewPointsPerStep=50_000, maxBufferBytes=0))
    opt = ops.Optimizer(dev, pc)
    try:
        small = dev.createBuffer(4 * mw * mh)
        ops.downsampleRGBA8(dev, dev.bufferFrom(target), cfg.width, cfg.height, small, mw, mh)
        harness.assert_bits_equal(small.read(np.uint8).reshape(mh, mw, 4), gt_small, "down-sampled ground truth")
        fwd.encode(None)
        rast.encode(None, mw, mh)
        harness.assert_bits_equal(rast.getOutputTextureView().read(np.uint8).reshape(mh, mw, 4), mfw["rgba8"], "metric render")
        mp.getMetricCountsBuffer().clear()
        mp.computeMetricMap(None, rast.getOutputTextureView(), small, dict(threshold=0.5))
        harness.assert_bits_equal(mp.getMetricMapTextureView().read(np.uint32).reshape(mh, mw), flags, "metric flags")
        mp.computeMetricCounts(None, dict(splatBuffer=fwd.getResources()["splatBuffer"], tileOffsetsBuffer=rast.getTileOffsetsBuffer(),
                                          tileIndicesBuffer=fwd.getSortedIndicesBuffer(), nContribTexture=rast.getNContribTextureView()), dict(clear=False))
        harness.assert_bits_equal(mp.getMetricCountsBuffer().read(np.uint32)[:n], counts_raw, "metric counts")
        mp.normalizeMetricCounts(None, dict(divisor=1))
        p = dp.encodePrepare(None, dict(pointCloud=pc, metricCountsBuffer=mp.getMetricCountsBuffer()))
        assert p["maxOutPoints"] == max_out and dp.readTotal() == prep["total"]
        harness.assert_bits_equal(p["actionBuffer"].read(np.uint32)[:n], prep["actions"], "actions")
        harness.assert_bits_equal(p["outCountBuffer"].read(np.uint32)[:n], prep["counts"], "out counts")
        harness.assert_bits_equal(p["outOffsetBuffer"].read(np.uint32)[:n], prep["offsets"], "out offsets")
        out_pc = ops.allocatePointCloudLike(dev, pc, dict(numPoints=out_n))
        out_state = ops.allocateOptimizerStateBuffers(dev, out_n)
        dp.encodeScatter(None, dict(pointCloud=pc, optimizerState=opt.getStateBuffers(), outOffsetBuffer=p["outOffsetBuffer"], outNumPoints=out_n,
                                    resetNewOptimizerState=True), dict(outPointCloud=out_pc, outOptimizerState=out_state))
        dev.synchronize()
        harness.assert_bits_equal(out_pc.gaussian_3d_buffer.read(np.uint32).reshape(-1, 6), og, "scattered gaussians")
        harness.assert_bits_equal(out_pc.sh_buffer.read(np.uint32).reshape(-1, 24), osh, "scattered sh")
        names = dict(optPosBuffer="opt_pos", optRotBuffer="opt_rot", optScaleBuffer="opt_scale", optOpacityBuffer="opt_opacity", paramSH="param_sh", stateSH="state_sh")
        for k, v in names.items():
            harness.assert_bits_equal(out_state[k].read(np.float32).reshape(ost[v].shape), ost[v], "scattered " + v)
    finally:
        for o in (opt, dp, mp, rast, fwd):
            o.destroy()


def test_c5_full_size_forward_and_train_step_equal_the_oracle(hip_device, orc):
    """BASELINE config c5's shape -- 5 M Gaussians at 3840x2160, SH 3, ~38 M tile entries (beyond every capacity cap of the
    reference) -- one complete training step against the oracle, bit for bit."""
    cfg = synth.CONFIGS["c5"]
    g, sh = synth.make_gaussians(cfg)
    cam = synth.identity_camera(cfg)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    target = np.full((cfg.height, cfg.width, 4), 255, np.uint8)
    target[..., :3] = (np.add.outer(np.arange(cfg.height), np.arange(cfg.width)) % 251).astype(np.uint8)[..., None]
    ref_g, ref_sh = g.copy(), sh.copy()
    ref_state = orc.unpack(ref_g, ref_sh)
    ref = orc.train_step(ref_g, ref_sh, ref_state, cam, st, ti, target)
    assert ref["total_entries"] > 30_000_000
    pipe = harness.HipPipeline(hip_device, cfg, g, sh, cam)
    try:
        pipe.train_step(hip_device.bufferFrom(target))
        hip_device.synchronize()
        got = pipe.collect_forward()
        n = cfg.num_points
        assert int(got["stats"][0]) == ref["total_entries"]
        harness.assert_bits_equal(got["sorted_values"], ref["sorted_values"][:ref["total_entries"]], "c5 sort order")
        harness.assert_bits_equal(got["rgba8"], ref["rgba8"], "c5 image")
        harness.assert_bits_equal(got["n_contrib"], ref["n_contrib"], "c5 n_contrib")
        harness.assert_bits_equal(pipe.bwd.getGradientsBuffer().read(np.uint32).reshape(-1, 8)[:n], ref["gradients"], "c5 packed gradients")
        harness.assert_bits_equal(pipe.pc.gaussian_3d_buffer.read(np.uint32).reshape(-1, 6), ref_g, "c5 re-packed Gaussians")
        harness.assert_bits_equal(pipe.pc.sh_buffer.read(np.uint32).reshape(-1, 24), ref_sh, "c5 re-packed SH")
    finally:
        pipe.destroy()


def test_c5_from_a_loaded_ply_with_the_50k_densify_equals_the_oracle_trainer(hip_device, orc, tmp_path):
    """BASELINE config c5 as written (the 8-GPU part aside): 5 M Gaussians LOADED FROM A .ply (written by exportPly, parsed by loadPly:
    the round trip must reproduce the fp16 cloud bit for bit), a 4K training step, then the densify/prune rebuild with
    maxNewPointsPerStep = 50 000 over two half-resolution metric views -- the Trainer against the oracle's restatement of trainer.ts,
    every buffer bit for bit."""
    from oracle import oracle_trainer
    from webdgs_amd import loaders
    from webdgs_amd.trainer import Trainer
    from test_gpu_trainer_oracle import _FixedViews, _compare
    cfg = synth.CONFIGS["c5"]
    g0, sh0 = synth.make_gaussians(cfg)
    path = tmp_path / "c5.ply"
    path.write_bytes(loaders.exportPly(g0, sh0, cfg.sh_deg))
    assert path.stat().st_size > 1_000_000_000
    pcd = loaders.loadPly(path.read_bytes())
    path.unlink()
    g, sh = pcd.gaussians, pcd.sh
    assert pcd.num_points == cfg.num_points and pcd.sh_deg == cfg.sh_deg
    harness.assert_bits_equal(g, g0, "c5 Gaussians after the .ply round trip")
    harness.assert_bits_equal(sh, sh0, "c5 SH after the .ply round trip")
    del g0, sh0
    dev = hip_device
    cams = synth.circle_cameras(cfg, 2)
    tg, tsh = synth.make_target_scene(g, sh)
    imgs = []
    for c in cams:  # ground truth by the HIP forward (equal to the oracle's: test_c5_train_step...), 33 MB each
        tp = harness.HipPipeline(dev, cfg, tg, tsh, c)
        tp.forward()
        imgs.append(tp.rast.getOutputTextureView().read(np.uint8).reshape(cfg.height, cfg.width, 4).copy())
        tp.destroy()
    del tg, tsh
    dens = dict(schedule=dict(enabled=True, warmupIterations=1, interval=1000, stopIterations=10), metricViews=2, metricThreshold=0.5, cloneThresholdCount=1,
                splitScaleThreshold=0.01, pruneOpacity=0.03, maxNewPointsPerStep=50_000, maxBufferBytes=2 ** 40)
    # (thresholds chosen from an oracle dry run of this scene so that additions exceed prunes by more than 50 k: the cap binds)
    o = oracle_trainer.OracleTrainer(g, sh, cfg.sh_deg, list(cams), imgs, densify=dens)
    t = Trainer(dev, seed=0)
    t.setDensifyPruneConfig(dens)
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset([dict(camera=cams[i], width=cfg.width, height=cfg.height) for i in range(2)],
                 [dict(texture=dev.bufferFrom(imgs[i]), width=cfg.width, height=cfg.height) for i in range(2)])
    t.start()
    try:
        t._rng = _FixedViews([1, 0, 1])
        o.step(1, metric_view_ids=[0, 1])
        t.step()
        d = o.last_densify
        a = d["prepared"]["actions"]
        assert d["used_views"] == 2 and d["rebuilt"] and min((a == i).sum() for i in range(4)) > 1000, [(a == i).sum() for i in range(4)]
        assert d["prepared"]["total"] >= d["max_out"] == cfg.num_points + 50_000 == o.num_points, "the 50k cap binds"
        _compare(t, o, "c5: step + 50k-capped densify")
    finally:
        t.destroy()


def test_c5_ply_goes_through_the_typescript_side_loaders_like_the_python_ones(tmp_path):
    """c5's cloud arrives as a .ply: the 1.2 GB file of 5 M Gaussians (SH degree 3) parsed by ``bindings/ts/loaders.js`` (node, no GPU) gives the
    packed Gaussians / SH the Python host's ``loadPly`` gives, and its ``exportPly`` writes the same file back -- compared by sha256."""
    import json
    import shutil
    from webdgs_amd import loaders
    if shutil.which("node") is None:
        pytest.skip("node is not on this box")
    cfg = synth.CONFIGS["c5"]
    g0, sh0 = synth.make_gaussians(cfg)
    data = loaders.exportPly(g0, sh0, cfg.sh_deg)
    path = tmp_path / "c5.ply"
    path.write_bytes(data)
    assert path.stat().st_size > 1_000_000_000
    want = dict(type="full", num_points=cfg.num_points, sh_deg=cfg.sh_deg, gaussians=hashlib.sha256(g0.tobytes()).hexdigest(), sh=hashlib.sha256(sh0.tobytes()).hexdigest(),
                exported=hashlib.sha256(data).hexdigest(), exported_bytes=len(data))
    del data
    r = subprocess.run(["node", "--max-old-space-size=8192", os.path.join(ROOT, "bindings", "napi", "ply_roundtrip_run.js"), str(path)], capture_output=True, text=True, timeout=900)
    path.unlink()
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert {k: got[k] for k in want} == want, (got, want)


@pytest.mark.parametrize("form", ["c3_as_written", "c4_rank_step"])
def test_c3_as_written_through_the_typescript_side_host_equals_the_python_host(hip_device, tmp_path, form):
    """BASELINE config c3 as written -- 1 M Gaussians, 1920x1080, SH degree 3, the full train loop with the reference's default densify/prune
    schedule (warm-up 500, interval 100: two rebuilds in 620 iterations, ten half-resolution metric views each) -- driven by
    ``bindings/ts/trainer.js`` under node and by ``webdgs_amd.trainer`` on the same view draws: final cloud and all six optimizer-state arrays
    sha256-equal, the same point counts after every step.  Second form: the per-rank step of BASELINE c4 at the same size -- 8 views per step on 3
    lanes, view-batched K1 / K17, the sliced exchange through the library's RCCL communicator in a world of one -- for 12 steps."""
    import json
    import shutil
    from webdgs_amd import parallel
    from webdgs_amd.trainer import Trainer
    from test_gpu_trainer_oracle import _FixedViews
    batched = form == "c4_rank_step"
    node = shutil.which("node")
    if not node or not os.path.exists(os.path.join(ROOT, "bindings", "napi", "webdgs_napi.node")):
        pytest.skip("node or the N-API addon is not available")
    dev = hip_device
    cfg = synth.CONFIGS["c3"]
    views, steps = 8, (12 if batched else 620)
    g, sh = synth.make_gaussians(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    cams = synth.circle_cameras(cfg, views)
    imgs = []
    for c in cams:  # ground truth by the HIP forward (equal to the oracle's: test_c3_forward_structure)
        tp = harness.HipPipeline(dev, cfg, tg, tsh, c)
        tp.forward()
        imgs.append(tp.rast.getOutputTextureView().read(np.uint8).reshape(cfg.height, cfg.width, 4).copy())
        tp.destroy()
    del tg, tsh
    rng = np.random.default_rng(7)
    draws = []
    for i in range(steps):
        draws += [int(v) for v in rng.integers(views, size=8 if batched else 1)]
        if i + 1 in (500, 600):
            draws += [int(v) for v in rng.integers(views, size=10)]
    opts = dict(views_per_step=8, lanes=3, comm="capi", pipeline_depth=2) if batched else dict(pipeline_depth=2)
    g.tofile(tmp_path / "gaussians.bin")
    sh.tofile(tmp_path / "sh.bin")
    np.ascontiguousarray(cams, np.float32).tofile(tmp_path / "cameras.bin")
    np.stack(imgs).tofile(tmp_path / "images.bin")
    (tmp_path / "meta.json").write_text(json.dumps(dict(num_points=cfg.num_points, sh_deg=cfg.sh_deg, width=cfg.width, height=cfg.height, views=views, steps=steps,
                                                        draws=draws, densify={}, hash_only=True, skip_probes=True, **opts)))
    r = subprocess.run([node, os.path.join(ROOT, "bindings", "napi", "trainer_run.js"), str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "TRAINER_RUN_OK" in r.stdout, f"exit code {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    out = json.loads((tmp_path / "out_meta.json").read_text())
    for f in ("gaussians.bin", "sh.bin", "images.bin"):
        (tmp_path / f).unlink()

    exchange = parallel.CapiExchange(dev) if batched else None
    t = Trainer(dev, seed=0, pipeline_depth=2, views_per_rank=8 if batched else 1, overlap_views=3 if batched else None, exchange=exchange)
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset([dict(camera=cams[i], width=cfg.width, height=cfg.height) for i in range(views)],
                 [dict(texture=dev.bufferFrom(imgs[i]), width=cfg.width, height=cfg.height) for i in range(views)])
    t.start()
    t._rng = _FixedViews(draws)
    sizes = [t.getPointCount()]
    try:
        for _ in range(steps):
            t.step()
            sizes.append(t.getPointCount())
        dev.synchronize()
        n = t.getPointCount()
        assert out["sizes"] == sizes and out["num_points"] == n and len(set(sizes)) == (1 if batched else 3), (sorted(set(out["sizes"])), sorted(set(sizes)))
        assert out["iteration"] == t.getIteration() == steps and out["last_densify"] == t.getLastDensifyPruneIteration() == (None if batched else 600)
        if batched:
            assert "RCCL" in out["exchange"], "the sliced step ran through the library's communicator"
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
        assert out["hashes"]["gaussians"] == sha(t.pointCloud.gaussian_3d_buffer.read(np.uint32)[: n * 6]), "node vs python: gaussians"
        assert out["hashes"]["sh"] == sha(t.pointCloud.sh_buffer.read(np.uint32)[: n * 24]), "node vs python: sh"
        words = dict(optPosBuffer=12, optRotBuffer=12, optScaleBuffer=12, optOpacityBuffer=3, paramSH=48, stateSH=96)
        for k, b in t.optimizer.getStateBuffers().items():
            assert out["hashes"][f"state_{k}"] == sha(b.read(np.uint32)[: n * words[k]]), f"node vs python: state {k}"
    finally:
        t.destroy()
        if exchange is not None:
            exchange.destroy()


def test_c2_long_run_is_the_same_whichever_way_it_is_driven(hip_device):
    """BASELINE c2 (100 k Gaussians, 640x480) for 1 250 iterations at the reference's densify schedule -- eight densify events,
    the cloud shrinking from 100 k -- driven two ways: as the reference drives it (every step awaited, passes destroyed and rebuilt
    at every swap, one lane) and as bench.py drives it (a step awaits the previous one, passes resized, command buffers re-recorded).
    Integer accumulation and fixed summation orders make the whole trajectory deterministic, so point cloud, point count and
    optimizer state must come out bit-identical."""
    from webdgs_amd.trainer import Trainer
    import bench
    dev = hip_device
    cfg = synth.CONFIGS["c2"]
    g, sh = synth.make_gaussians(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    cameras, images = bench.make_dataset(dev, cfg, tg, tsh, synth.circle_cameras(cfg, 8))

    def run(depth, reuse, command_buffers, vpr=1, lanes=1, iterations=1250):
        t = Trainer(dev, seed=77, pipeline_depth=depth, use_command_buffers=command_buffers, views_per_rank=vpr, overlap_views=lanes)
        t.reuse_passes = reuse
        t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
        t.setDataset(cameras, images)
        t.setMaxIterations(10 ** 9)
        t.start()
        sizes = [t.getPointCount()]
        while t.getIteration() < iterations:
            t.step()
            if t.getPointCount() != sizes[-1]:
                sizes.append(t.getPointCount())
        t.drain()
        dev.synchronize()
        out = dict(sizes=sizes, g=_digest(t.pointCloud.gaussian_3d_buffer.read(np.uint32)), sh=_digest(t.pointCloud.sh_buffer.read(np.uint32)),
                   state={k: _digest(b.read(np.uint32)) for k, b in t.optimizer.getStateBuffers().items()})
        t.destroy()
        return out

    a = run(2, True, True)
    b = run(1, False, True)
    assert len(a["sizes"]) >= 8 and a["sizes"][-1] != cfg.num_points, a["sizes"]
    assert a == b, (a["sizes"], b["sizes"])
    # the batched step (c4's shape, four views per step): three lanes, pipelined, resized passes vs one lane, awaited, rebuilt passes
    c = run(2, True, True, vpr=4, lanes=3, iterations=720)
    d = run(1, False, True, vpr=4, lanes=1, iterations=720)
    assert len(c["sizes"]) >= 3 and c == d, (c["sizes"], d["sizes"])


def test_c3_batched_step_on_lanes_equals_one_lane(hip_device):
    """BASELINE c3 / c4's per-rank step at full size -- 1 M Gaussians, 1920x1080, four views per step -- where the kernels of different
    lanes really do run side by side for hundreds of microseconds: three lanes with the pipelined submission must leave the bits of
    one lane with every step awaited (cloud and optimizer state after 6 steps), with the view-batched K1 / K17 kernels and with the
    per-view ones."""
    from webdgs_amd.trainer import Trainer
    import bench
    dev = hip_device
    cfg = synth.CONFIGS["c3"]
    g, sh = synth.make_gaussians(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    cameras, images = bench.make_dataset(dev, cfg, tg, tsh, synth.circle_cameras(cfg, 4))

    def run(lanes, depth, batch_views=None):
        t = Trainer(dev, seed=5, views_per_rank=4, overlap_views=lanes, pipeline_depth=depth, batch_views=batch_views)
        t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
        t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
        t.setDataset(cameras, images)
        t.start()
        t.step([0, 1, 2, 3])
        t.warmupCommandBuffers()
        on_lanes = 0
        for ids in ([3, 1, 0, 2], [2, 2, 1, 0], [0, 3, 3, 1], [1, 0, 2, 3], [3, 2, 1, 0], [0, 1, 2, 3]):
            key = (lambda k, v: ("viewp", v, k)) if t.batch_views else (lambda k, v: ("view", v, k % t._lanes))
            on_lanes += int(t._lanes > 1 and all(key(k, v) in t._cmd_cache for k, v in enumerate(ids)))
            t.step(ids)
        t.drain()
        dev.synchronize()
        out = dict(g=_digest(t.pointCloud.gaussian_3d_buffer.read(np.uint32)), sh=_digest(t.pointCloud.sh_buffer.read(np.uint32)),
                   state={k: _digest(b.read(np.uint32)) for k, b in t.optimizer.getStateBuffers().items()})
        t.destroy()
        return out, on_lanes

    a, lanes_a = run(3, 2)                       # view-batched K1 / K17 (round 3), one op set per view, three lanes
    b, lanes_b = run(1, 1, batch_views=False)    # round 2's per-view kernels, one lane, every step awaited
    c, lanes_c = run(3, 2, batch_views=False)    # ... and those on three lanes
    assert lanes_a == 6 and lanes_b == 0 and lanes_c == 6
    assert a == b and c == b
