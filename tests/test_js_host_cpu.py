"""CPU: the TypeScript-side host's loaders, camera, image ingest, scene generator and slice arithmetic (bindings/ts/loaders.js, camera.js,
images.js, synth.js, parallel.js) run by node WITHOUT a GPU and compared bit for bit with the Python host (webdgs_amd/loaders.py, images.py,
synth.py, parallel.py), which tests/test_loaders.py and tests/test_images.py pin to the reference's formats."""
import json
import math
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from webdgs_amd import images, loaders, parallel, synth
from webdgs_amd.viewer import encodePNG

from test_loaders import _images_bin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")


def _normal_ply():
    header = ("ply\nformat binary_little_endian 1.0\ncomment " + "x" * 70 + "\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n")
    rows = [(1.5, -2.25, 3.0, 255, 0, 128), (0.1, 0.2, 0.3, 10, 20, 30), (7, 8, 9, 1, 2, 3), (70000.0, -1e-9, 6.1e-5, 0, 255, 77)]
    return header.encode() + b"".join(struct.pack("<3f3B", *v) for v in rows)


def _points3d():
    pts = [(1, (0.5, 1.5, -2.0), (255, 128, 0), 0.1, [(1, 2), (3, 4)]), (2, (10.0, 20.0, 30.0), (1, 2, 3), 0.2, []), (9, (1e-7, -65520.0, 0.33333), (9, 99, 199), 1.5, [(5, 6)])]
    data = struct.pack("<Q", len(pts))
    for pid, xyz, rgb, err, track in pts:
        data += struct.pack("<Q3d3BdQ", pid, *xyz, *rgb, err, len(track)) + b"".join(struct.pack("<II", *t) for t in track)
    return data


@pytest.fixture(scope="module")
def run(tmp_path_factory):
    d = tmp_path_factory.mktemp("jshost")
    clouds = {}
    for deg in (0, 3):
        cfg = synth.SceneConfig(11 + deg, 301, 64, 64, deg, 100.0, 0.01)
        g, sh = synth.make_gaussians(cfg)
        clouds[f"full{deg}.ply"] = loaders.exportPly(g, sh, deg)
    clouds["normal.ply"] = _normal_ply()
    clouds["points3D.bin"] = _points3d()
    names = ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(6)] + ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    bad = {"baddeg.ply": ("ply\nformat binary_little_endian 1.0\nelement vertex 1\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n").encode() + b"\0" * (4 * len(names)),
           "short.ply": _normal_ply()[:-4], "trunc.bin": struct.pack("<Q", 5) + b"\0" * 10}
    for k, v in {**clouds, **bad}.items():
        (d / k).write_bytes(v)
    cfg = synth.CONFIGS["c2"]
    cams_json = []
    for i, blk in enumerate(synth.circle_cameras(cfg, 3)):
        view = blk[0:16].reshape(4, 4).T.astype(np.float64)
        rot_rows = view[:3, :3]
        cams_json.append(dict(id=i, img_name=f"v{i}.png", width=cfg.width, height=cfg.height, fx=123.0, fy=cfg.fy + i, position=list(-rot_rows.T @ view[:3, 3]), rotation=rot_rows.tolist()))
    (d / "cams.json").write_text(json.dumps(cams_json))
    th = 0.3
    (d / "images.bin").write_bytes(_images_bin([(7, (math.cos(th / 2), 0.0, math.sin(th / 2), 0.0), (0.5, -1.0, 2.0), 3, "a.png", 2), (8, (0.5, 0.5, -0.5, 0.5), (1e-3, 7.25, -3.0), 4, "b b.png", 0),
                                                (9, (1, 0, 0, 0), (0, 0, 0), 99, "c.png", 1)]))
    (d / "cameras.bin").write_bytes(struct.pack("<Q", 2) + struct.pack("<IiQQ3d", 3, 0, 640, 480, 500.0, 320.0, 240.0) + struct.pack("<IiQQ4d", 4, 1, 800, 600, 700.0, 710.0, 400.0, 300.0))
    os.mkdir(d / "images")
    rng = np.random.default_rng(5)
    frames = {}
    for name, (h, w) in {"frame_10.png": (9, 14), "frame_9.PNG": (9, 14), "Frame_2.png": (5, 7)}.items():
        a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        a[::2, :, :3] = (np.arange(w, dtype=np.uint8) * 5)[None, :, None]
        frames[name] = a
        (d / "images" / name).write_bytes(encodePNG(a))
    try:  # other colour types and filters, written by Pillow when it is there
        import io

        from PIL import Image
        a = rng.integers(0, 256, (11, 13, 4), dtype=np.uint8)
        for mode, arr in {"RGB": a[..., :3], "L": a[..., 0], "LA": a[..., [0, 3]]}.items():
            buf = io.BytesIO()
            Image.fromarray(np.ascontiguousarray(arr), mode).save(buf, format="PNG", optimize=True)
            (d / "images" / f"pil_{mode}.png").write_bytes(buf.getvalue())
        buf = io.BytesIO()
        Image.fromarray(a[..., :3], "RGB").quantize(16).save(buf, format="PNG")
        (d / "images" / "pil_P.png").write_bytes(buf.getvalue())
        # JPEG ground truth (what COLMAP datasets hold): samplings, qualities, progressive, restart markers, greyscale, sizes that are no multiple
        # of the MCU -- the JS decoder (bindings/ts/jpeg.js) must give the texels Pillow's libjpeg-turbo gives
        yy, xx = np.mgrid[0:45, 0:70]
        smooth = np.stack([xx * 255 // 69, yy * 255 // 44, (xx + yy) * 7 % 256], -1).astype(np.uint8)
        smooth[12:30, 20:50] = rng.integers(0, 256, (18, 30, 3), dtype=np.uint8)
        k = 0
        for sub in (0, 1, 2):
            for prog in (False, True):
                for q, pic in ((35, smooth), (92, smooth[:37, :53]), (75, smooth[:16, :17])):
                    buf = io.BytesIO()
                    Image.fromarray(np.ascontiguousarray(pic), "RGB").save(buf, format="JPEG", quality=q, subsampling=sub, progressive=prog, optimize=bool(k & 1))
                    (d / "images" / f"jpg_{k:02d}.jpg").write_bytes(buf.getvalue())
                    k += 1
        for kw in (dict(restart_marker_blocks=3), dict(restart_marker_rows=1, progressive=True)):
            buf = io.BytesIO()
            Image.fromarray(smooth, "RGB").save(buf, format="JPEG", quality=85, subsampling=2, **kw)
            (d / "images" / f"jpg_{k:02d}.jpeg").write_bytes(buf.getvalue())
            k += 1
        for prog in (False, True):
            buf = io.BytesIO()
            Image.fromarray(np.ascontiguousarray(smooth[:33, :47, 0]), "L").save(buf, format="JPEG", quality=80, progressive=prog)
            (d / "images" / f"jpg_{k:02d}.JPG").write_bytes(buf.getvalue())
            k += 1
    except ImportError:
        pass
    (d / "images" / "notes.txt").write_bytes(b"hello")
    (d / "images" / "broken.png").write_bytes(b"\x89PNG\r\n\x1a\nxxxx")
    (d / "images" / "photo.jpg").write_bytes(b"\xff\xd8\xff\xe0 not really a jpeg")
    meta = dict(clouds=list(clouds), bad_clouds={k: True for k in bad}, header_ply="normal.ply", synth=[dict(config="c1", points=3000, cameras=5), dict(config="c3", points=2000, cameras=8)],
                slices=[[1000, 1], [1000, 8], [40, 2], [1_000_000, 8], [63, 4], [0, 2]])
    (d / "meta.json").write_text(json.dumps(meta))
    r = subprocess.run([NODE, os.path.join(ROOT, "bindings", "napi", "host_cpu_run.js"), str(d)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "HOST_CPU_RUN_OK" in r.stdout, f"exit code {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    return d, json.loads((d / "out.json").read_text()), clouds, frames, r.stderr


def test_point_clouds_load_to_the_same_fp16_words(run):
    d, out, clouds, _, _ = run
    for name, data in clouds.items():
        pc = loaders.loadPointCloud(data)
        assert out["clouds"][name] == dict(type=pc.type, num_points=pc.num_points, sh_deg=pc.sh_deg), name
        assert np.array_equal(np.fromfile(d / f"out_{name}.g", np.uint32).reshape(-1, 6), pc.gaussians), name
        assert np.array_equal(np.fromfile(d / f"out_{name}.sh", np.uint32).reshape(-1, 24), pc.sh), name
    assert out["export_round_trip"]
    pc0 = loaders.loadPointCloud(clouds["full0.ply"])
    assert (d / "out_export.ply").read_bytes() == loaders.exportPly(pc0.gaussians, pc0.sh, pc0.sh_deg), "the two exporters write the same file"
    assert "Unsupported SH degree" in out["errors"]["baddeg.ply"] and "too short" in out["errors"]["short.ply"] and out["errors"]["trunc.bin"].startswith("Failed to load pointcloud")


def test_ply_header_surface_of_plyreader(run):
    _, out, clouds, _, _ = run
    n, props, off = loaders.decodeHeader(clouds["normal.ply"])
    h = out["header"]
    assert h["count"] == n == 4 and h["names"] == list(props) and h["types"] == props and props["uchar"] == "list"
    assert h["first_offset"] == 15 and h["first_vertex"]["x"] == 1.5 and h["first_vertex"]["red"] == 1.0 and abs(h["first_vertex"]["blue"] - 128 / 255) < 1e-15
    assert h["nsh"] == [1, 4, 9, 16] and "Unsupported SH degree" in h["nsh_error"]


def test_cameras_and_uniform_blocks(run):
    d, out, _, _, _ = run
    cj = loaders.loadCameraJson((d / "cams.json").read_bytes())
    merged = loaders.mergeColmap(loaders.loadColmapImagesBin((d / "images.bin").read_bytes()), loaders.loadColmapCamerasBin((d / "cameras.bin").read_bytes()))
    for got, want in zip(out["cameras"]["json"] + out["cameras"]["merged"], cj + merged):
        for k, v in want.items():
            if k in ("position", "rotation"):
                # (values, not bits: JSON carries -0 as 0; the blocks built from these cameras are compared bit for bit below)
                assert np.array_equal(np.asarray(got[k], np.float32), np.asarray(v, np.float32)), k
            else:
                assert got[k] == v, (k, got[k], v)
    assert len(out["cameras"]["merged"]) == 3 and "width" not in out["cameras"]["merged"][2] and out["cameras"]["only_images"] == 3
    assert [c["fy"] for c in out["cameras"]["only_cameras"]] == [500.0, 710.0]
    blocks = np.fromfile(d / "out_blocks.f32", np.float32).reshape(-1, 68)
    want = []
    for c in cj + merged:
        want += [loaders.cameraUniforms(c) if c.get("width") else None, loaders.cameraUniforms(c, 333, 201)]
    want.append(loaders.cameraUniforms({}, 200, 100))
    for i, w in enumerate(want):
        if w is not None:   # (a camera without width / height has no default canvas: only the explicit size is comparable)
            assert np.array_equal(blocks[i].view(np.uint32), w.view(np.uint32)), f"camera block {i}"
    cc = out["camera_class"]
    assert cc["equals_uniforms"] and cc["writes"] == 2 and math.isclose(cc["default_focal"], 0.5 * 201 / math.tan(math.radians(22.5)), rel_tol=1e-6)
    assert "Unsupported camera file format" in out["errors"]["camera"]
    raw = json.loads((d / "cams.json").read_text())
    for p, j in zip(out["presets"], raw):   # mat3.create(...rotation.flat()) + mat4.fromMat3: the nine numbers in file order as columns
        r = np.asarray(p["rotation"], np.float32).reshape(4, 4)
        assert np.array_equal(r[:3, :3], np.asarray(j["rotation"], np.float32)) and r[3, 3] == 1 and np.allclose(p["position"], j["position"][:3])


def test_images_are_filtered_ordered_decoded_and_dropped_alike(run):
    d, out, _, frames, stderr = run
    files = [str(d / "images" / f) for f in os.listdir(d / "images")]
    want = images.loadImages(files)
    assert [i["name"] for i in out["images"]] == [w.name for w in want]
    for i, w in enumerate(want):
        assert (out["images"][i]["width"], out["images"][i]["height"]) == (w.width, w.height)
        assert np.array_equal(np.fromfile(d / f"out_image_{i}.rgba", np.uint8).reshape(w.height, w.width, 4), w.bitmap), w.name
    names = [i["name"] for i in out["images"]]
    try:
        import PIL  # noqa: F401
        assert sum(n.lower().endswith((".jpg", ".jpeg")) for n in names) == 22, "every JPEG written above was decoded by both hosts (and compared above)"
    except ImportError:
        pass
    assert names.index("Frame_2.png") < names.index("frame_9.PNG") < names.index("frame_10.png") and "broken.png" not in names and "photo.jpg" not in names
    assert "Failed to load image broken.png" in stderr and "Failed to load image photo.jpg" in stderr and out["png_round_trip"]


def test_scene_generator_draws_the_same_bits(run):
    d, _, _, _, _ = run
    for name, pts, ncam in (("c1", 3000, 5), ("c3", 2000, 8)):
        cfg = synth.CONFIGS[name]
        g, sh = synth.make_gaussians(cfg, pts)
        tg, tsh = synth.make_target_scene(g, sh)
        for ext, a in (("g", g), ("sh", sh), ("tg", tg), ("tsh", tsh)):
            assert np.array_equal(np.fromfile(d / f"out_synth_{name}.{ext}", np.uint32).reshape(a.shape), a), (name, ext)
        cams = np.concatenate([synth.circle_cameras(cfg, ncam), synth.identity_camera(cfg)[None]])
        assert np.array_equal(np.fromfile(d / f"out_synth_{name}.cams", np.uint32).reshape(-1, 68), cams.view(np.uint32)), name


def test_slice_arithmetic_of_the_data_parallel_step(run):
    _, out, _, _, _ = run
    for s in out["slices"]:
        assert s["slice"] == parallel.slice_points(s["n"], s["w"])
        assert [(o["first"], o["count"]) for o in s["owned"]] == [parallel.owned_range(s["n"], s["w"], r) for r in range(s["w"])]
    assert out["shard"] == parallel.shard_views([5, 6, 7, 8, 9], 1, 2)
