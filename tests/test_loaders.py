"""CPU: on-disk formats -- PLY / COLMAP loaders, PLY export, camera loaders, camera block builder."""
import json
import math
import struct

import numpy as np
import pytest

from webdgs_amd import loaders, synth


def _f16(x):
    return np.array(x, np.float64).astype(np.float16)


def test_ply_export_load_round_trip_all_degrees():
    for deg in (0, 1, 2, 3):
        cfg = synth.SceneConfig(11, 257, 64, 64, deg, 100.0, 0.01)
        g, sh = synth.make_gaussians(cfg)
        pc = loaders.loadPointCloud(loaders.exportPly(g, sh, deg))
        assert (pc.type, pc.num_points, pc.sh_deg) == ("full", 257, deg)
        assert np.array_equal(pc.gaussians, g) and np.array_equal(pc.sh, sh)


def test_ply_header_quirks_and_normal_cloud_colour_scaling():
    # header longer than one 50-byte chunk, a list property (ignored), uchar colours divided by 255 twice (Q22)
    header = ("ply\nformat binary_little_endian 1.0\ncomment " + "x" * 70 + "\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n")
    body = b"".join(struct.pack("<3f3B", *v) for v in [(1.5, -2.25, 3.0, 255, 0, 128), (0.1, 0.2, 0.3, 10, 20, 30), (7, 8, 9, 1, 2, 3)])
    n, props, off = loaders.decodeHeader(header.encode() + body)
    assert n == 3 and list(props)[:6] == ["x", "y", "z", "red", "green", "blue"] and props["uchar"] == "list"
    pc = loaders.loadPly(header.encode() + body)
    assert pc.type == "normal" and pc.sh_deg == 0 and pc.num_points == 3
    g = pc.gaussians.view(np.float16).reshape(-1, 12)
    assert np.array_equal(g[0, :3], _f16([1.5, -2.25, 3.0]))
    assert np.array_equal(g[:, 3:12], np.tile(_f16([1, 1, 0, 0, 0, -5, -5, -5, 0]), (3, 1)))  # load-pointcloud.ts:255-265
    s = pc.sh.view(np.float16).reshape(-1, 48)
    expect = _f16([((255 / 255.0) / 255.0 - 0.5) / loaders.C0, ((0 / 255.0) / 255.0 - 0.5) / loaders.C0, ((128 / 255.0) / 255.0 - 0.5) / loaders.C0])
    assert np.array_equal(s[0, :3], expect) and np.all(s[:, 3:] == 0)
    with pytest.raises(ValueError):
        loaders.loadPly(header.encode() + body[:-4])


def test_ply_unsupported_sh_degree_raises():
    names = ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(6)] + ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    header = "ply\nformat binary_little_endian 1.0\nelement vertex 1\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with pytest.raises(ValueError, match="Unsupported SH degree"):  # 2 rest coefficients per colour -> degree sqrt(3)-1
        loaders.loadPly(header.encode() + b"\0" * (4 * len(names)))


def test_colmap_points3d_bin():
    pts = [(1, (0.5, 1.5, -2.0), (255, 128, 0), 0.1, [(1, 2), (3, 4)]), (2, (10.0, 20.0, 30.0), (1, 2, 3), 0.2, [])]
    data = struct.pack("<Q", len(pts))
    for pid, xyz, rgb, err, track in pts:
        data += struct.pack("<Q3d3BdQ", pid, *xyz, *rgb, err, len(track)) + b"".join(struct.pack("<II", *t) for t in track)
    pc = loaders.loadPointCloud(data)
    assert pc.type == "normal" and pc.num_points == 2
    g = pc.gaussians.view(np.float16).reshape(-1, 12)
    assert np.array_equal(g[1, :3], _f16([10, 20, 30]))
    s = pc.sh.view(np.float16).reshape(-1, 48)
    assert np.array_equal(s[0, :3], _f16([(255 / 255.0 - 0.5) / loaders.C0, (128 / 255.0 - 0.5) / loaders.C0, (0 - 0.5) / loaders.C0]))
    with pytest.raises(ValueError, match="Failed to load pointcloud"):
        loaders.loadPointCloud(struct.pack("<Q", 5) + b"\0" * 10)


def _images_bin(entries):
    data = struct.pack("<Q", len(entries))
    for image_id, q, t, cam_id, name, npts in entries:
        data += struct.pack("<I7dI", image_id, *q, *t, cam_id) + name.encode() + b"\0" + struct.pack("<Q", npts) + b"\0" * (24 * npts)
    return data


def test_colmap_cameras_and_images_merge():
    th = 0.3
    q = (math.cos(th / 2), 0.0, math.sin(th / 2), 0.0)  # w,x,y,z: rotation about y
    imgs = loaders.loadColmapImagesBin(_images_bin([(7, q, (0.5, -1.0, 2.0), 3, "a.png", 2), (8, (1, 0, 0, 0), (0, 0, 0), 9, "b.png", 0)]))
    cams = loaders.loadColmapCamerasBin(struct.pack("<Q", 2) + struct.pack("<IiQQ3d", 3, 0, 640, 480, 500.0, 320.0, 240.0) + struct.pack("<IiQQ4d", 4, 1, 800, 600, 700.0, 710.0, 400.0, 300.0))
    assert cams[0]["fx"] == cams[0]["fy"] == 500.0 and cams[1]["fy"] == 710.0
    merged = loaders.mergeColmap(imgs, cams)
    assert merged[0]["id"] == 7 and merged[0]["width"] == 640 and merged[0]["img_name"] == "a.png" and "width" not in merged[1]
    R = merged[0]["rotation"].astype(np.float64).reshape(4, 4).T[:3, :3]  # row-major 3x3 world->camera
    Ry = np.array([[math.cos(th), 0, math.sin(th)], [0, 1, 0], [-math.sin(th), 0, math.cos(th)]])
    assert np.allclose(R, Ry, atol=1e-6)
    assert np.allclose(merged[0]["position"], -Ry.T @ np.array([0.5, -1.0, 2.0]), atol=1e-6)
    with pytest.raises(ValueError, match="Unsupported COLMAP camera model"):
        loaders.loadColmapCamerasBin(struct.pack("<Q", 1) + struct.pack("<IiQQ", 1, 2, 10, 10) + b"\0" * 64)
    assert loaders.loadColmapImagesBin(b"\0\0") == []


def test_camera_json_and_uniform_block_match_the_synthetic_builder():
    cfg = synth.CONFIGS["c2"]
    for blk in synth.circle_cameras(cfg, 5):
        view = blk[0:16].reshape(4, 4).T.astype(np.float64)
        rot_rows = view[:3, :3]
        centre = -rot_rows.T @ view[:3, 3]
        js = json.dumps([dict(id=1, img_name="x", width=cfg.width, height=cfg.height, fx=123.0, fy=cfg.fy, position=list(centre), rotation=rot_rows.tolist())])
        # load-camera.ts:148-155 copies rotation[i][c] into column c, row i of the mat4: the matrix IS json.rotation (world->camera)
        cam = loaders.loadCameraJson(js.encode())[0]
        got = loaders.cameraUniforms(cam)
        assert np.allclose(got, blk, rtol=2e-6, atol=2e-6)
        assert got[66] == got[67], "fx is ignored: focal.x = focal.y (SURVEY Q18)"
    # default fovY = 45 degrees when intrinsics are missing (camera.ts:134)
    d = loaders.cameraUniforms(dict(position=np.zeros(3, np.float32), rotation=np.eye(4, dtype=np.float32).reshape(-1)), 200, 100)
    assert math.isclose(d[67], 0.5 * 100 / math.tan(math.radians(22.5)), rel_tol=1e-6)
