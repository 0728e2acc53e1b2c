"""GPU: the TypeScript-side host beyond the one-view step -- loaders, image ingest and Viewer driven by node
(bindings/napi/viewer_run.js) against the Python host on the same files: a JS-loaded .ply renders ``==`` the Python-loaded one in both
render modes and on a resized canvas; a viewer that shares its PointCloud with a running trainer shows the TRAINED colours without any
hand-over call (deferred SH writes: ADVICE r3 medium), in both hosts; state handles kept across steps are current when read; and
bindings/napi/bench.js prints its line."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from webdgs_amd import images, loaders, ops, synth
from webdgs_amd.trainer import Trainer
from webdgs_amd.viewer import Viewer, encodePNG

import harness
from harness import assert_bits_equal
from test_gpu_trainer_oracle import _FixedViews

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "bindings", "napi", "webdgs_napi.node")


def _need_node():
    if not NODE or not os.path.exists(ADDON):
        pytest.skip("node or the N-API addon is not available")


def _camera_json(cfg, n):
    out = []
    for i, blk in enumerate(synth.circle_cameras(cfg, n)):
        view = blk[0:16].reshape(4, 4).T.astype(np.float64)
        rot_rows = view[:3, :3]
        out.append(dict(id=i, img_name=f"gt_{i}.png", width=cfg.width, height=cfg.height, fx=cfg.fy, fy=cfg.fy, position=list(-rot_rows.T @ view[:3, 3]), rotation=rot_rows.tolist()))
    return out


def test_js_loaded_ply_renders_like_the_python_loaded_one_and_the_viewer_follows_training(hip_device, orc, tmp_path):
    _need_node()
    dev = hip_device
    cfg = harness.small_config("c2", num_points=6000, width=144, height=96, s0=0.01)
    g, sh, _ = harness.scene(cfg)
    (tmp_path / "scene.ply").write_bytes(loaders.exportPly(g, sh, cfg.sh_deg))
    cams_json = _camera_json(cfg, 3)
    (tmp_path / "cams.json").write_text(json.dumps(cams_json))
    cams = loaders.loadCameraJson((tmp_path / "cams.json").read_bytes())
    tg, tsh = synth.make_target_scene(g, sh)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    os.mkdir(tmp_path / "gt")
    for i, c in enumerate(cams):
        (tmp_path / "gt" / f"gt_{i}.png").write_bytes(encodePNG(orc.forward(tg, tsh, loaders.cameraUniforms(c, cfg.width, cfg.height), st, ti)["rgba8"]))
    steps, draws, resized = 7, [0, 2, 1, 1, 0, 2, 2, 0, 1], (100, 72)
    (tmp_path / "meta.json").write_text(json.dumps(dict(width=cfg.width, height=cfg.height, resized=resized, view_camera=1, steps=steps, draws=draws)))
    r = subprocess.run([NODE, os.path.join(ROOT, "bindings", "napi", "viewer_run.js"), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "VIEWER_RUN_OK" in r.stdout, f"exit code {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    out = json.loads((tmp_path / "out.json").read_text())

    def js_frame(name, w=cfg.width, h=cfg.height):
        return np.fromfile(tmp_path / name, np.uint8).reshape(h, w, 4)

    # ---- the Python host on the same files
    data = loaders.loadPointCloud((tmp_path / "scene.ply").read_bytes())
    assert (out["type"], out["num_points"], out["sh_deg"]) == (data.type, data.num_points, data.sh_deg) and np.array_equal(data.gaussians, g)
    pc = ops.createPointCloud(dev, data.gaussians, data.sh, data.sh_deg)
    v = Viewer(dev, cfg.width, cfg.height)
    t = None
    try:
        v.setCamera(cams[1])
        v.setPointCloud(pc)
        v.render(None)
        assert_bits_equal(js_frame("out_frame_points.rgba"), v.readFrame(), "point-cloud mode: JS-loaded .ply vs Python-loaded")
        v.setRenderMode("gaussian")
        v.render(None)
        gauss = v.readFrame().copy()
        assert_bits_equal(js_frame("out_frame_gaussian.rgba"), gauss, "gaussian mode: JS-loaded .ply vs Python-loaded")
        assert_bits_equal(gauss, orc.forward(g, sh, loaders.cameraUniforms(cams[1], cfg.width, cfg.height), st, ti)["rgba8"], "... and both are the oracle's image")
        assert_bits_equal(images.decodePNG((tmp_path / "out_frame.png").read_bytes()), gauss, "the PNG the JS viewer saved")
        v.setPointSize(5.0)
        v.setRenderMode("pointcloud")
        v.render(None)
        assert_bits_equal(js_frame("out_frame_points5.rgba"), v.readFrame(), "setPointSize + setRenderMode")
        v.resize(*resized)
        v.setRenderMode("gaussian")
        v.render(None)
        assert_bits_equal(js_frame("out_frame_resized.rgba", *resized), v.readFrame(), "resized canvas")
        v.resize(cfg.width, cfg.height)

        # ---- train the shared cloud; the viewer is never told (no flushPointCloud)
        gt = images.loadImages([str(tmp_path / "gt" / f) for f in os.listdir(tmp_path / "gt")], dev)
        assert out["images"] == [[im.name, im.width, im.height] for im in gt] and out["cameras"] == 3
        t = Trainer(dev, seed=0)
        assert t.deferred_sh, "deferred SH writes are the Trainer's default: that is the case under test"
        t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
        t.setPointCloud(pc)
        t.setDataset(cams, gt)
        t.start()
        t._rng = _FixedViews(draws)
        for _ in range(steps):
            t.step()
        v.render(None)
        trained = v.readFrame().copy()
        assert_bits_equal(js_frame("out_frame_trained.rgba"), trained, "unflushed viewer beside a running trainer: node vs Python")
        assert not np.array_equal(trained, gauss), "training changed the image"
        import torch
        from webdgs_amd import parallel
        stale = parallel._tensor_at(dev, pc.sh_buffer.ptr, pc.num_points * 24, torch.int32).cpu().numpy().view(np.uint32)
        rows = pc.sh_buffer.read(np.uint32)[: pc.num_points * 24]
        assert out["stale_rows_seen"] and not np.array_equal(stale, rows), "the rows really are stale until a hand-over (otherwise this test shows nothing)"
        assert_bits_equal(np.fromfile(tmp_path / "out_sh.bin", np.uint32), rows, "host read of the SH rows through the hook: node vs Python")
        assert_bits_equal(np.fromfile(tmp_path / "out_gaussians.bin", np.uint32), pc.gaussian_3d_buffer.read(np.uint32)[: pc.num_points * 6], "trained Gaussians: node vs Python")
        # what the unflushed viewer showed IS the trained cloud: a fresh pass on a flushed copy of the cloud, no dc source, renders the same image
        flushed = ops.createPointCloud(dev, pc.gaussian_3d_buffer.read(np.uint32)[: pc.num_points * 6].reshape(-1, 6), rows.reshape(-1, 24), pc.sh_deg)
        v2 = Viewer(dev, cfg.width, cfg.height)
        v2.setCamera(cams[1])
        v2.setPointCloud(flushed)
        v2.setRenderMode("gaussian")
        v2.render(None)
        assert_bits_equal(v2.readFrame(), trained, "unflushed viewer == viewer of the flushed cloud")
        v2.destroy()
        # a state handle fetched before further steps is current when read
        kept = t.optimizer.getStateBuffers()
        for _ in range(2):
            t.step()
        pos_kept = kept["optPosBuffer"].read(np.uint32)[: pc.num_points * 12]
        assert_bits_equal(pos_kept, t.optimizer.getStateBuffers()["optPosBuffer"].read(np.uint32)[: pc.num_points * 12], "kept state handle == fresh getStateBuffers()")
        assert_bits_equal(np.fromfile(tmp_path / "out_state_pos_kept.bin", np.uint32), pos_kept, "kept state handle: node vs Python")
        t.destroy()
        t = None
        assert pc.dc_words is None and out["dc_words_after_trainer"], "the optimizer's destruction hands the rows back"
        v.render(None)
        assert_bits_equal(js_frame("out_frame_after_trainer.rgba"), v.readFrame(), "viewer after the trainer is gone")
    finally:
        if t is not None:
            t.destroy()
        v.destroy()


def _quat_wxyz(R):
    """Unit quaternion (w, x, y, z) of a proper rotation matrix (COLMAP's images.bin stores the world -> camera rotation that way)."""
    t = np.trace(R)
    if t > 0:
        s = 2.0 * np.sqrt(t + 1.0)
        return np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = 2.0 * np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k])
    q = np.zeros(4)
    q[0], q[1 + i], q[1 + j], q[1 + k] = (R[k, j] - R[j, k]) / s, 0.25 * s, (R[j, i] + R[i, j]) / s, (R[k, i] + R[i, k]) / s
    return q


def test_a_colmap_dataset_with_jpeg_ground_truth_trains_alike_in_both_hosts(hip_device, orc, tmp_path):
    """north_star: "outputs match ... on identical COLMAP/PLY inputs".  A COLMAP reconstruction as it lies on disk -- points3D.bin, images.bin,
    cameras.bin and JPEG images -- goes through each host's OWN loaders (node: loaders.js + jpeg.js; Python: loaders.py + Pillow), each host trains the
    cloud on it, and the trained clouds and the viewer frames must be equal byte for byte."""
    _need_node()
    Image = pytest.importorskip("PIL.Image")
    import io
    import struct
    dev = hip_device
    cfg = harness.small_config("c2", num_points=7000, width=160, height=112, s0=0.01)
    g, sh, _ = harness.scene(cfg)
    g16 = g.view(np.float16).reshape(-1, 12).astype(np.float64)
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (cfg.num_points, 3), dtype=np.uint8)
    pts = struct.pack("<Q", cfg.num_points)
    for i in range(cfg.num_points):   # id, xyz, rgb, error, track length + track
        track = [(1, 2)] * (i % 3)
        pts += struct.pack("<Q3d3BdQ", i + 1, *g16[i, :3], *rgb[i], 0.5, len(track)) + b"".join(struct.pack("<II", *t) for t in track)
    (tmp_path / "points3D.bin").write_bytes(pts)
    blocks = synth.circle_cameras(cfg, 3)
    imgs_bin = struct.pack("<Q", 3)
    for i, blk in enumerate(blocks):
        view = blk[0:16].reshape(4, 4).T.astype(np.float64)
        R, t = view[:3, :3], view[:3, 3]
        imgs_bin += struct.pack("<I7dI", 10 + i, *_quat_wxyz(R), *t, 1) + f"frame_{i:02d}.jpg".encode() + b"\0" + struct.pack("<Q", i) + b"\0" * (24 * i)
    (tmp_path / "images.bin").write_bytes(imgs_bin)
    (tmp_path / "cameras.bin").write_bytes(struct.pack("<Q", 1) + struct.pack("<IiQQ4d", 1, 1, cfg.width, cfg.height, cfg.fy, cfg.fy, cfg.width / 2, cfg.height / 2))
    cams = loaders.mergeColmap(loaders.loadColmapImagesBin(imgs_bin), loaders.loadColmapCamerasBin((tmp_path / "cameras.bin").read_bytes()))
    data = loaders.loadPointCloud(pts)
    assert data.type == "normal" and data.sh_deg == 0
    # ground truth: the same cloud with brighter colours, rendered by the oracle under the LOADED cameras, stored as JPEG (4:2:0, one progressive)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    st[1] = 0.0
    tg, tsh = synth.make_target_scene(data.gaussians, data.sh)
    os.mkdir(tmp_path / "gt")
    for i, c in enumerate(cams):
        ref = orc.forward(tg, tsh, loaders.cameraUniforms(c, cfg.width, cfg.height), st, ti)["rgba8"]
        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(ref[..., :3]), "RGB").save(buf, format="JPEG", quality=90, subsampling=2, progressive=(i == 1))
        (tmp_path / "gt" / f"frame_{i:02d}.jpg").write_bytes(buf.getvalue())
    steps, draws, resized = 6, [1, 0, 2, 2, 1, 0, 0, 1], (96, 64)
    (tmp_path / "meta.json").write_text(json.dumps(dict(width=cfg.width, height=cfg.height, resized=resized, view_camera=2, steps=steps, draws=draws,
                                                        cloud_file="points3D.bin", camera_files=["images.bin", "cameras.bin"])))
    r = subprocess.run([NODE, os.path.join(ROOT, "bindings", "napi", "viewer_run.js"), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "VIEWER_RUN_OK" in r.stdout, f"exit code {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    out = json.loads((tmp_path / "out.json").read_text())
    assert (out["type"], out["num_points"], out["sh_deg"], out["cameras"]) == ("normal", cfg.num_points, 0, 3)

    pc = ops.createPointCloud(dev, data.gaussians, data.sh, data.sh_deg)
    gt = images.loadImages([str(tmp_path / "gt" / f) for f in os.listdir(tmp_path / "gt")], dev)
    assert out["images"] == [[im.name, im.width, im.height] for im in gt]
    v = Viewer(dev, cfg.width, cfg.height)
    t = Trainer(dev, seed=0)
    try:
        v.setCamera(cams[2])
        v.setPointCloud(pc)
        v.setRenderMode("gaussian")
        v.render(None)
        assert_bits_equal(np.fromfile(tmp_path / "out_frame_gaussian.rgba", np.uint8).reshape(cfg.height, cfg.width, 4), v.readFrame(), "COLMAP cloud under a COLMAP camera: node vs Python")
        t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
        t.setPointCloud(pc)
        t.setDataset(cams, gt)
        t.start()
        t._rng = _FixedViews(draws)
        for _ in range(steps):
            t.step()
        v.render(None)
        assert_bits_equal(np.fromfile(tmp_path / "out_frame_trained.rgba", np.uint8).reshape(cfg.height, cfg.width, 4), v.readFrame(), "after training on the JPEG ground truth: node vs Python")
        n = pc.num_points
        assert_bits_equal(np.fromfile(tmp_path / "out_gaussians.bin", np.uint32), pc.gaussian_3d_buffer.read(np.uint32)[: n * 6], "trained Gaussians: node vs Python")
        assert_bits_equal(np.fromfile(tmp_path / "out_sh.bin", np.uint32), pc.sh_buffer.read(np.uint32)[: n * 24], "trained SH rows: node vs Python")
        assert not np.array_equal(pc.gaussian_3d_buffer.read(np.uint32)[: n * 6], data.gaussians.reshape(-1)), "training moved the cloud"
    finally:
        t.destroy()
        v.destroy()


def test_keep_gradients_switch_is_obeyed_after_set_point_cloud(hip_device, orc):
    """``keep_gradients`` flipped AFTER ``setPointCloud`` (the passes survive swaps) must not leave getGradientsBuffer() stale."""
    dev = hip_device
    cfg = harness.small_config("c1", num_points=3000, width=96, height=64)
    g, sh, cam = harness.scene(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    target = orc.forward(tg, tsh, cam, st, ti)["rgba8"]
    t = Trainer(dev, seed=0)
    try:
        t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
        t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
        t.setDataset([dict(camera=cam, width=cfg.width, height=cfg.height)], [dict(texture=dev.bufferFrom(target), width=cfg.width, height=cfg.height)])
        t.start()
        ref_g, ref_sh = g.copy(), sh.copy()
        state = orc.unpack(ref_g, ref_sh)
        for _ in range(3):   # eager, recording, replay -- gradient output off
            t.step()
            orc.train_step(ref_g, ref_sh, state, cam, st, ti, target)
        assert not t.keep_gradients and len(t._cmd_cache) == 1
        t.keep_gradients = True
        t.step()
        ref = orc.train_step(ref_g, ref_sh, state, cam, st, ti, target)
        assert_bits_equal(t.backwardPass.getGradientsBuffer().read(np.uint32).reshape(-1, 8)[: cfg.num_points], ref["gradients"], "gradients of the step after the switch")
        t.step()
        t.step()   # recorded again, with the output on
        for _ in range(2):
            ref = orc.train_step(ref_g, ref_sh, state, cam, st, ti, target)
        assert_bits_equal(t.backwardPass.getGradientsBuffer().read(np.uint32).reshape(-1, 8)[: cfg.num_points], ref["gradients"], "gradients of a replayed step after the switch")
        assert_bits_equal(t.pointCloud.gaussian_3d_buffer.read(np.uint32).reshape(-1, 6), ref_g, "the cloud")
    finally:
        t.destroy()


@pytest.mark.parametrize("depth,vpr", [(1, 1), (2, 1), (1, 3), (2, 3)])
def test_tile_entry_lists_grow_after_an_overflow_in_both_hosts_alike(hip_device, depth, vpr):
    """A cloud of few, large splats outruns the tile-entry lists the library sized for it (what a long run of the default schedule arrives at).
    Both Trainers double the lists, warn and go on; the steps lost to the overflow are the same ones, so the clouds end up byte-identical."""
    import hashlib
    import warnings
    _need_node()
    r = subprocess.run([NODE, os.path.join(ROOT, "bindings", "napi", "capacity_run.js"), str(depth), str(vpr)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["grown"] >= 1 and out["cap"] >= out["needed"] > (1 << 20), out

    dev = hip_device
    cfg = synth.SceneConfig(2, 6000, 512, 384, 1, 550.0, 0.2, "few-large-splats")
    g, sh = synth.make_gaussians(cfg)
    cams = synth.circle_cameras(cfg, 2)
    # the Viewers: a first frame that outruns the library-sized lists is rendered again around larger ones by readFrame() -- in both hosts the picture
    # a pass with lists pinned large enough gives
    vpc = ops.createPointCloud(dev, g, sh, cfg.sh_deg)
    v = Viewer(dev, cfg.width, cfg.height)
    v.setCamera(cams[0]); v.setPointCloud(vpc); v.setRenderMode("gaussian")
    v.render(None)
    frame = v.readFrame().copy()
    vcap = int(v.getForwardPass().getResources()["maxTileEntries"])
    v.destroy()
    cam = dev.bufferFrom(np.asarray(cams[0], np.float32))
    fw = ops.TiledForwardPass(dev, vpc, cam, dict(viewportWidth=cfg.width, viewportHeight=cfg.height, renderMode="gaussian", maxTileEntries=8 << 20))
    rs = ops.TiledRasterizer(dict(device=dev, forwardPass=fw))
    fw.encode(None); rs.encode(None, cfg.width, cfg.height)
    whole = rs.getOutputTextureView().read(np.uint8).reshape(cfg.height, cfg.width, 4).copy()
    assert int(fw.check()[0]) > (1 << 20)
    rs.destroy(); fw.destroy(); vpc.gaussian_3d_buffer.destroy(); vpc.sh_buffer.destroy()
    assert vcap > (1 << 20), "the Python viewer grew its lists"
    assert_bits_equal(frame, whole, "the Python viewer's frame after the growth vs a pass with room from the start")
    assert out["viewer"]["cap"] == vcap and out["viewer"]["frame"] == hashlib.sha256(whole.tobytes()).hexdigest(), "the JS viewer's frame likewise"
    images = [dict(texture=dev.bufferFrom(np.zeros((cfg.height, cfg.width, 4), np.uint8)), width=cfg.width, height=cfg.height) for _ in cams]
    t = Trainer(dev, seed=0, pipeline_depth=depth, views_per_rank=vpr, overlap_views=2 if vpr > 1 else None)
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    t.setDataset([dict(camera=c, width=cfg.width, height=cfg.height) for c in cams], images)
    t.start()
    try:
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            for i in range(8):
                t.step([(i + k) % 2 for k in range(vpr)])
            t.drain()
            dev.synchronize()
        grown = sum("tile-entry lists grown" in str(w.message) for w in seen)
        n = t.getPointCount()
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
        assert (out["grown"], out["iteration"], out["cap"]) == (grown, t.getIteration(), int(t.forwardPass.getResources()["maxTileEntries"])), (out, grown, t.getIteration())
        assert out["gaussians"] == sha(t.pointCloud.gaussian_3d_buffer.read(np.uint32)[: n * 6]), "node vs python after the growth: gaussians"
        assert out["sh"] == sha(t.pointCloud.sh_buffer.read(np.uint32)[: n * 24]), "node vs python after the growth: sh"
    finally:
        t.destroy()


@pytest.mark.parametrize("depth", [1, 2])
def test_a_live_trainer_follows_its_host_in_both_hosts_alike(hip_device, orc, tmp_path, depth):
    """What a host changes under a running Trainer -- loss weights, learning rates, a dataset of another image size, stop / start, the densify
    schedule switched on -- invalidates recordings and passes in different ways in the two hosts (the JS Trainer rebuilds its backward pass for new
    loss weights, the Python one updates it in place ...): same sequence, same view draws, clouds and optimizer state sha256-equal."""
    import hashlib
    _need_node()
    dev = hip_device
    cfg = harness.small_config("c2", num_points=5000, width=128, height=96, s0=0.01)
    g, sh, _ = harness.scene(cfg)
    tg, tsh = synth.make_target_scene(g, sh)
    views, size_a, size_b = 4, (128, 96), (96, 64)

    def dataset(w, h):
        c = synth.SceneConfig(cfg.config_id, cfg.num_points, w, h, cfg.sh_deg, cfg.fy * w / cfg.width, cfg.s0, "views")
        cams = synth.circle_cameras(c, views)
        st, ti = synth.render_settings(c), synth.tile_info(w, h, 0)
        return cams, [orc.forward(tg, tsh, cams[i], st, ti)["rgba8"] for i in range(views)]

    cams_a, imgs_a = dataset(*size_a)
    cams_b, imgs_b = dataset(*size_b)
    dens = dict(schedule=dict(enabled=True, warmupIterations=4, interval=50, stopIterations=40), metricViews=3, metricDownscale=2, metricThreshold=0.5,
                cloneThresholdCount=5, splitScaleThreshold=0.03, pruneOpacity=0.2, maxNewPointsPerStep=300, maxBufferBytes=128 * 1024 * 1024)
    rng = np.random.default_rng(11)
    # 4 + 3 + 3 + 3 steps, stop / start, 2 steps, schedule on, 6 steps: the restarted count reaches 4 on its 4th step -> one event of 3 metric views
    draws = [int(v) for v in rng.integers(views, size=13 + 2)] + [int(v) for v in rng.integers(views, size=2)] + [int(v) for v in rng.integers(views, size=3)] + \
            [int(v) for v in rng.integers(views, size=4)] + [int(v) for v in rng.integers(views, size=2)]   # (+ 2 steps on the resized cloud)
    g.tofile(tmp_path / "gaussians.bin"); sh.tofile(tmp_path / "sh.bin")
    for tag, cams, imgs in (("a", cams_a, imgs_a), ("b", cams_b, imgs_b)):
        np.ascontiguousarray(cams, np.float32).tofile(tmp_path / f"cameras_{tag}.bin")
        np.stack(imgs).tofile(tmp_path / f"images_{tag}.bin")
    (tmp_path / "meta.json").write_text(json.dumps(dict(num_points=cfg.num_points, sh_deg=cfg.sh_deg, views=views, a=size_a, b=size_b, draws=draws, densify=dens,
                                                        pipeline_depth=depth)))
    r = subprocess.run([NODE, os.path.join(ROOT, "bindings", "napi", "host_changes_run.js"), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, f"exit code {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    out = json.loads(r.stdout.strip().splitlines()[-1])

    t = Trainer(dev, seed=0, pipeline_depth=depth)
    t.setDensifyPruneConfig(dict(schedule=dict(enabled=False)))
    t.setPointCloud(ops.createPointCloud(dev, g, sh, cfg.sh_deg))
    sets = {}
    for tag, cams, imgs, (w, h) in (("a", cams_a, imgs_a, size_a), ("b", cams_b, imgs_b, size_b)):
        sets[tag] = ([dict(camera=cams[i], width=w, height=h) for i in range(views)], [dict(texture=dev.bufferFrom(imgs[i]), width=w, height=h) for i in range(views)])
    t.setDataset(*sets["a"])
    t.start()
    t._rng = _FixedViews(draws)
    seen = []
    try:
        def steps(n):
            for _ in range(n):
                t.step()
        steps(4)
        t.setTrainingConfig(dict(lambda_l1=0.6, lambda_dssim=0.4))
        steps(3)
        hp = t.getOptimizerHyperparameters()
        t.setOptimizerHyperparameters(dict(lr_pos=hp["lr_pos"] * 2, lr_color=hp["lr_color"] * 0.5))
        steps(3)
        t.setDataset(*sets["b"])
        steps(3)
        seen.append(t.getIteration())
        t.stop(); t.start()
        seen.append(t.getIteration())
        steps(2)
        t.setDensifyPruneConfig(dens)
        steps(6)
        t.drain()
        dev.synchronize()
        assert t._rng.i == len(draws) - 2
        n = t.getPointCount()
        assert (out["num_points"], out["iteration"], out["iterations_seen"], out["last_densify"]) == (n, t.getIteration(), seen, t.getLastDensifyPruneIteration()), out
        assert n != cfg.num_points and t.getLastDensifyPruneIteration() == 4, "the schedule switched on mid-run fired"
        assert out["training_config"] == t.getTrainingConfig() and out["lr_pos"] == pytest.approx(t.getOptimizerHyperparameters()["lr_pos"], rel=1e-7)
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
        assert out["hashes"]["gaussians"] == sha(t.pointCloud.gaussian_3d_buffer.read(np.uint32)[: n * 6]), "node vs python: gaussians"
        assert out["hashes"]["sh"] == sha(t.pointCloud.sh_buffer.read(np.uint32)[: n * 24]), "node vs python: sh"
        words = dict(optPosBuffer=12, optRotBuffer=12, optScaleBuffer=12, optOpacityBuffer=3, paramSH=48, stateSH=96)
        for k, b in t.optimizer.getStateBuffers().items():
            assert out["hashes"][f"state_{k}"] == sha(b.read(np.uint32)[: n * words[k]]), f"node vs python: state {k}"
        # the debug helper of trainer.ts:194: a swap to a zero-filled cloud of another size, applied by the host at a step boundary; two more steps
        t.requestResizeTo(3000)
        t.applyPointCloudSwap(t.consumePointCloudSwapRequest())
        steps(2)
        t.drain()
        dev.synchronize()
        assert t._rng.i == len(draws), "the schedule of view draws was used up exactly"
        rz = out["resized"]
        assert (rz["num_points"], rz["iteration"]) == (t.getPointCount(), t.getIteration()) == (3000, 10)
        assert rz["gaussians"] == sha(t.pointCloud.gaussian_3d_buffer.read(np.uint32)[: 3000 * 6]) and rz["sh"] == sha(t.pointCloud.sh_buffer.read(np.uint32)[: 3000 * 24])
        assert rz["stateSH"] == sha(t.optimizer.getStateBuffers()["stateSH"].read(np.uint32)[: 3000 * 96])
    finally:
        t.destroy()


def test_bench_js_prints_its_line(tmp_path):
    _need_node()
    for extra in ([], ["--views-per-step", "3", "--lanes", "2", "--views", "4"]):
        r = subprocess.run([NODE, os.path.join(ROOT, "bindings", "napi", "bench.js"), "--config", "c1", "--steps", "6", "--warmup", "2", "--min-seconds", "0.05"] + extra,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, f"exit code {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["unit"] == "iters/s" and line["value"] > 0 and line["n_gpus"] == 1 and line["steps"] == 6 and line["ms_per_step"] > 0
        assert line["ms_per_step_awaiting_every_step"] > 0 and line["timed_blocks"]["blocks"] >= 2 and "node" in line["host"]
        assert line["config"]["global_batch_views"] == (3 if extra else 1) and line["config"]["tile_entries_E"] > 0
        assert line["kernel_ms_per_step"] and any(k.startswith("backward_rasterize") for k in line["kernel_ms_per_step"])
