"""On-disk formats either side of the hot path: PLY / COLMAP point clouds, COLMAP and JSON
cameras, the 272-byte camera block, PLY export.

Host-side restatement of ``/root/reference/src/utils/plyreader.ts``, ``load-pointcloud.ts``, ``load-camera.ts`` and
``src/camera/camera.ts`` (citations below are relative to ``src/``), vectorised with numpy.  Values go through the same
number types as in the browser: JS numbers are binary64, ``Float16Array`` stores round binary64 -> binary16 once
(``@petamoriken/float16`` 3.8.7, round-to-nearest-even), wgpu-matrix (3.2.0) matrices are ``Float32Array``.  Quirks kept:
``uchar`` properties are divided by 255 on read AND again as colours; only ``float`` and ``uchar`` properties
advance the read offset (``plyreader.ts:63-72``); ``fx``, ``cx``, ``cy`` are ignored by the camera (Q18).
"""
from __future__ import annotations

import dataclasses
import json
import math
import re
import struct
from typing import Optional

import numpy as np

from . import synth

C0 = 0.28209479177387814


@dataclasses.dataclass
class PointCloudData:
    """Host copy of a ``PointCloud`` (``utils/load-pointcloud.ts:16-23``): upload with ``ops.createPointCloud``."""

    type: str                 # 'full' (splats) or 'normal' (points with default splat parameters)
    num_points: int
    sh_deg: int
    gaussians: np.ndarray     # uint32[N, 6]  = 12 fp16
    sh: np.ndarray            # uint32[N, 24] = 48 fp16, [k][rgb]


def _f16_words(a64: np.ndarray, halves_per_row: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        h = np.ascontiguousarray(a64.astype(np.float16))
    return h.view(np.uint32).reshape(-1, halves_per_row // 2)


# ----------------------------------------------------------------------------- PLY
def decodeHeader(data: bytes):
    """``decodeHeader`` (``utils/plyreader.ts:1-54``): returns (vertexCount, propertyTypes in declaration order, vertex byte offset).  The header
    is the text in front of the word ``end_header``; the payload starts one byte after that word (the reference's own offset rule).  A
    ``property`` line is taken as ``property <type> <name>`` -- so ``property list uchar int vertex_indices`` declares a property named
    ``uchar`` of type ``list``, exactly as the reference's three-word match reads it; only ``float`` and ``uchar`` TYPES are ever read."""
    end = data.find(b"end_header")
    if end < 0:
        raise ValueError("PLY header: 'end_header' not found")
    vertex_count, props = 0, {}
    for raw in data[:end].decode("utf-8", "replace").split("\n"):
        words = raw.split()
        if words[:2] == ["element", "vertex"]:
            digits = re.search(r"\d+", raw)
            if digits:
                vertex_count = int(digits.group(0))
        elif words and words[0] == "property" and len(words) >= 3:
            props[words[2]] = words[1]
    return vertex_count, props, end + len(b"end_header") + 1


def nShCoeffs(deg: float) -> int:
    if deg in (0, 1, 2, 3):
        return (int(deg) + 1) ** 2
    raise ValueError(f"Unsupported SH degree: {deg}")


def _normal_defaults(n: int, xyz: np.ndarray) -> np.ndarray:
    g = np.zeros((n, 12), np.float64)
    g[:, 0:3] = xyz
    g[:, 3], g[:, 4] = 1.0, 1.0       # raw opacity 1.0, quaternion (1,0,0,0)
    g[:, 8:11] = -5.0                 # log-sigma
    return g


def loadPly(data: bytes) -> PointCloudData:
    """``loadPly`` (``utils/load-pointcloud.ts:156-307``)."""
    n, props, voff = decodeHeader(data)
    fields = [(name, "<f4" if t == "float" else "u1") for name, t in props.items() if t in ("float", "uchar")]
    dt = np.dtype(fields)
    if len(data) < voff + n * dt.itemsize:
        raise ValueError(f"PLY payload too short: need {n * dt.itemsize} bytes after the header, have {len(data) - voff}")
    v = np.frombuffer(data, dtype=dt, count=n, offset=voff)

    def col(name: str) -> np.ndarray:  # readRawVertex: float as is, uchar / 255.0 (binary64)
        a = v[name].astype(np.float64)
        return a / 255.0 if props[name] == "uchar" else a

    is_full = "rot_0" in props and "scale_0" in props
    sh = np.zeros((n, 48), np.float64)
    if is_full:
        n_rest = sum(1 for p in props if p.startswith("f_rest_"))
        per_color = n_rest / 3
        sh_deg = math.sqrt(per_color + 1) - 1
        num_coefs = nShCoeffs(sh_deg)
        per_color = int(per_color)
        order = [f"f_dc_{c}" for c in range(3)] + [f"f_rest_{c * per_color + i}" for i in range(per_color) for c in range(3)]
        g = np.zeros((n, 12), np.float64)
        for k, name in enumerate(("x", "y", "z", "opacity", "rot_0", "rot_1", "rot_2", "rot_3", "scale_0", "scale_1", "scale_2")):
            g[:, k] = col(name)
        for o in range(num_coefs):
            for j in range(3):
                sh[:, o * 3 + j] = col(order[o * 3 + j])
        return PointCloudData("full", n, int(sh_deg), _f16_words(g, 12), _f16_words(sh, 48))
    g = _normal_defaults(n, np.stack([col("x"), col("y"), col("z")], axis=1))
    if "red" in props:
        rgb = np.stack([col("red"), col("green"), col("blue")], axis=1) / 255.0      # second division: SURVEY Q22
    elif "diffuse_red" in props:
        rgb = np.stack([col("diffuse_red"), col("diffuse_green"), col("diffuse_blue")], axis=1) / 255.0
    else:
        rgb = np.full((n, 3), 0.5)
    sh[:, 0:3] = (rgb - 0.5) / C0
    return PointCloudData("normal", n, 0, _f16_words(g, 12), _f16_words(sh, 48))


def loadColmapBin(data: bytes) -> PointCloudData:
    """COLMAP ``points3D.bin`` (``utils/load-pointcloud.ts:54-154``): id u64, xyz f64 x3, rgb u8 x3, error f64, track (u64 + 8 B each)."""
    (n,) = struct.unpack_from("<Q", data, 0)
    off = 8
    xyz = np.zeros((n, 3), np.float64)
    rgb = np.zeros((n, 3), np.float64)
    for i in range(n):
        off += 8
        xyz[i] = struct.unpack_from("<3d", data, off); off += 24
        rgb[i] = struct.unpack_from("<3B", data, off); off += 3
        off += 8
        (track_len,) = struct.unpack_from("<Q", data, off); off += 8
        off += track_len * 8
    g = _normal_defaults(n, xyz)
    sh = np.zeros((n, 48), np.float64)
    sh[:, 0:3] = (rgb / 255.0 - 0.5) / C0
    return PointCloudData("normal", n, 0, _f16_words(g, 12), _f16_words(sh, 48))


def loadPointCloud(data: bytes) -> PointCloudData:
    """``loadPointCloud`` (``utils/load-pointcloud.ts:29-52``): 'ply' magic, else COLMAP points3D.bin."""
    if data[:3] == b"ply":
        return loadPly(data)
    try:
        return loadColmapBin(data)
    except Exception as exc:  # the reference rethrows with this prefix
        raise ValueError(f"Failed to load pointcloud: {exc}") from exc


def exportPly(gaussians: np.ndarray, sh: np.ndarray, sh_deg: int) -> bytes:
    """Binary little-endian 3DGS PLY of a 'full' cloud (the reference has no exporter).  ``loadPly(exportPly(...))``
    returns the same fp16 words: fp16 -> fp32 is exact and the loader rounds back to the same fp16."""
    g = np.ascontiguousarray(gaussians).view(np.float16).reshape(-1, 12).astype(np.float32)
    s = np.ascontiguousarray(sh).view(np.float16).reshape(-1, 48).astype(np.float32)
    n, k = g.shape[0], (sh_deg + 1) ** 2
    per_color = k - 1
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(3 * per_color)] + \
            ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    out = np.zeros((n, len(names)), np.float32)
    idx = {nm: i for i, nm in enumerate(names)}
    out[:, idx["x"]], out[:, idx["y"]], out[:, idx["z"]], out[:, idx["opacity"]] = g[:, 0], g[:, 1], g[:, 2], g[:, 3]
    for j in range(4):
        out[:, idx[f"rot_{j}"]] = g[:, 4 + j]
    for j in range(3):
        out[:, idx[f"scale_{j}"]] = g[:, 8 + j]
        out[:, idx[f"f_dc_{j}"]] = s[:, j]
    for i in range(per_color):           # loader order: f_rest_{rgb * per_color + i} <- coefficient (i+1), channel rgb
        for c in range(3):
            out[:, idx[f"f_rest_{c * per_color + i}"]] = s[:, (i + 1) * 3 + c]
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + "".join(f"property float {nm}\n" for nm in names) + "end_header\n"
    return header.encode("ascii") + out.astype("<f4").tobytes()


# ----------------------------------------------------------------------------- cameras
def _f32(a) -> np.ndarray:
    return np.asarray(a, np.float64).astype(np.float32)


def loadCameraJson(data: bytes) -> list:
    """``loadCameraJson`` (``utils/load-camera.ts:138-168``): rotation rows become columns of a column-major mat4."""
    j = json.loads(data.decode("utf-8"))
    out = []
    for c in (j if isinstance(j, list) else [j]):
        r = c["rotation"]
        rot = np.zeros(16, np.float32)
        rot[0:3] = (r[0][0], r[1][0], r[2][0])
        rot[4:7] = (r[0][1], r[1][1], r[2][1])
        rot[8:11] = (r[0][2], r[1][2], r[2][2])
        rot[15] = 1.0
        out.append(dict(id=c.get("id"), img_name=c.get("img_name"), width=c.get("width"), height=c.get("height"), fx=c.get("fx"), fy=c.get("fy"),
                        position=_f32(c["position"][:3]), rotation=rot))
    return out


def _from_quat(x: float, y: float, z: float, w: float) -> np.ndarray:
    """wgpu-matrix 3.2.0 ``mat4.fromQuat`` on Float32Array inputs: column-major rotation matrix, computed in binary64, stored f32."""
    x, y, z, w = (float(np.float32(v)) for v in (x, y, z, w))
    x2, y2, z2 = x + x, y + y, z + z
    xx, yx, yy, zx, zy, zz, wx, wy, wz = x * x2, y * x2, y * y2, z * x2, z * y2, z * z2, w * x2, w * y2, w * z2
    return _f32([1 - yy - zz, yx + wz, zx - wy, 0, yx - wz, 1 - xx - zz, zy + wx, 0, zx + wy, zy - wx, 1 - xx - yy, 0, 0, 0, 0, 1])


def loadColmapImagesBin(data: bytes) -> list:
    """COLMAP ``images.bin`` (``utils/load-camera.ts:171-240``): world position C = -R^T t, rotation kept as R (world->camera)."""
    if len(data) < 8:
        return []
    (n,) = struct.unpack_from("<Q", data, 0)
    off, out = 8, []
    for _ in range(n):
        (image_id,) = struct.unpack_from("<I", data, off); off += 4
        qw, qx, qy, qz, tx, ty, tz = struct.unpack_from("<7d", data, off); off += 56
        (camera_id,) = struct.unpack_from("<I", data, off); off += 4
        end = data.index(b"\0", off)
        name = data[off:end].decode("latin-1"); off = end + 1
        (npts,) = struct.unpack_from("<Q", data, off); off += 8 + npts * 24
        rot = _from_quat(qx, qy, qz, qw)
        m = rot.astype(np.float64).reshape(4, 4)           # m[c][r]; transpose(R) applied to T as vec3.transformMat4 does
        t = _f32([tx, ty, tz]).astype(np.float64)
        rt = m.T                                            # Rt[c][r] = R[r][c]
        c = np.array([rt[0][0] * t[0] + rt[1][0] * t[1] + rt[2][0] * t[2] + rt[3][0],
                      rt[0][1] * t[0] + rt[1][1] * t[1] + rt[2][1] * t[2] + rt[3][1],
                      rt[0][2] * t[0] + rt[1][2] * t[1] + rt[2][2] * t[2] + rt[3][2]])
        out.append(dict(id=image_id, camera_id=camera_id, img_name=name, rotation=rot, position=-_f32(c)))
    return out


def loadColmapCamerasBin(data: bytes) -> list:
    """COLMAP ``cameras.bin`` (``utils/load-camera.ts:243-288``): models 0 (SIMPLE_PINHOLE) and 1 (PINHOLE)."""
    (n,) = struct.unpack_from("<Q", data, 0)
    off, out = 8, []
    for _ in range(n):
        camera_id, model_id = struct.unpack_from("<Ii", data, off); off += 8
        width, height = struct.unpack_from("<2Q", data, off); off += 16
        if model_id == 0:
            f, cx, cy = struct.unpack_from("<3d", data, off); off += 24
            fx = fy = f
        elif model_id == 1:
            fx, fy, cx, cy = struct.unpack_from("<4d", data, off); off += 32
        else:
            raise ValueError(f"Unsupported COLMAP camera model ID: {model_id}")
        out.append(dict(id=camera_id, camera_id=camera_id, width=width, height=height, fx=fx, fy=fy, cx=cx, cy=cy))
    return out


def mergeColmap(images: list, cameras: list) -> list:
    """The images.bin + cameras.bin merge of ``loadCamera`` (``utils/load-camera.ts:45-72``): intrinsics by camera_id, id from the image."""
    cmap = {c["id"]: c for c in cameras}
    out = []
    for img in images:
        if img.get("camera_id") in cmap:
            out.append({**img, **cmap[img["camera_id"]], "id": img["id"]})
        else:
            out.append(dict(img))
    return out


def cameraUniforms(cam: dict, width: Optional[int] = None, height: Optional[int] = None, znear: float = 0.01, zfar: float = 100.0) -> np.ndarray:
    """``Camera.set_preset`` + ``update_buffer`` (``camera/camera.ts:23-56, 138-205``): the 68-float block for a CameraData.

    ``width``/``height`` = canvas size (the trainer sets it to the image size, ``trainer.ts:583-584``).  Matrices are formed
    in binary64 from Float32Array operands and stored as f32, as wgpu-matrix does; the inverses are wgpu-matrix's cofactor formula
    (``synth.mat4_inverse``) applied to the stored float32 matrices, as ``update_buffer`` applies ``mat4.inverse`` to them."""
    w = int(width if width is not None else cam["width"])
    h = int(height if height is not None else cam["height"])
    fov_y = 45.0 / 180.0 * math.pi
    if cam.get("fx") and cam.get("fy") and cam.get("height"):
        fov_y = 2.0 * math.atan(cam["height"] / (2.0 * cam["fy"]))
    focal = 0.5 * h / math.tan(fov_y * 0.5)
    fov_x = 2.0 * math.atan(w / (2.0 * focal))
    rot = np.asarray(cam.get("rotation") if cam.get("rotation") is not None else np.eye(4).reshape(-1), np.float32).reshape(16)
    pos = _f32(cam.get("position") if cam.get("position") is not None else (0, 0, 5)).astype(np.float64)
    # mat4.translate(r, -t) as wgpu-matrix 3.2.0 evaluates it: the first three columns are copied as they are (a -0 stays a -0), the fourth is
    # m[c0] * v0 + m[c1] * v1 + m[c2] * v2 + m[c3] in binary64, stored as f32
    view_cm = rot.copy()
    r64 = rot.astype(np.float64)
    v0, v1, v2 = (float(x) for x in -pos)
    for r in range(4):
        view_cm[12 + r] = np.float32(r64[r] * v0 + r64[4 + r] * v1 + r64[8 + r] * v2 + r64[12 + r])
    tan_y, tan_x = math.tan(fov_y / 2.0), math.tan(fov_x / 2.0)
    top, right = tan_y * znear, tan_x * znear
    proj = np.zeros((4, 4))
    proj[0, 0] = 2.0 * znear / (2.0 * right)
    proj[1, 1] = -2.0 * znear / (2.0 * top)
    proj[2, 2] = zfar / (zfar - znear)
    proj[2, 3] = -(zfar * znear) / (zfar - znear)
    proj[3, 2] = 1.0
    proj = proj.astype(np.float32).astype(np.float64)
    out = np.zeros(68, np.float32)
    out[0:16] = view_cm
    out[32:48] = proj.T.reshape(-1)
    out[16:32] = synth.mat4_inverse(out[0:16])
    out[48:64] = synth.mat4_inverse(out[32:48])
    out[64:68] = (w, h, focal, focal)
    return out
