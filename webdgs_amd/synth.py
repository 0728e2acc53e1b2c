"""Deterministic synthetic scenes for parity tests and benchmarks.

The reference trains on COLMAP/PLY data loaded by ``src/utils/load-pointcloud.ts``; the on-disk
loaders are a "next" row.  The hot path only ever sees three byte-exact inputs, which
this module fabricates directly:

* ``gaussians``  uint32[N, 6]  -- 12 fp16: x y z opacity_raw | rot w x y z | log-sigma x y z, pad
                                  (``src/shaders/common.wgsl:20-24``, ``src/utils/load-pointcloud.ts:233-245``)
* ``sh``         uint32[N, 24] -- 48 fp16, coefficient-major ``[k][rgb]``, always 16 slots
                                  (``src/shaders/tiled-forward.wgsl:64-86``)
* ``camera``     float32[68]   -- view, view_inv, proj, proj_inv (column-major), viewport, focal
                                  (``src/camera/camera.ts:165-195``)

PRNG: splitmix64 seeded ``0x5EEDD650000 + config_id``; the state is a counter, so the whole stream is
vectorised.  Per-Gaussian draw order (part of the fixture contract): z, x, y, quat x4 (Box-Muller, two
uniforms each, cosine branch), opacity (Box-Muller), log-sigma x3, SH DC x3 (uniform), then bands
1..deg in ``[k][rgb]`` order (Box-Muller each).
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_SEED_BASE = 0x5EEDD650000


def _splitmix64(seed: int, count: int, offset: int = 0) -> np.ndarray:
    """``count`` consecutive outputs of splitmix64(seed), starting at output index ``offset``."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform01(seed: int, count: int, offset: int = 0) -> np.ndarray:
    return (_splitmix64(seed, count, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _box_muller(u1: np.ndarray, u2: np.ndarray) -> np.ndarray:
    return np.sqrt(-2.0 * np.log1p(-u1)) * np.cos(2.0 * math.pi * u2)


@dataclasses.dataclass(frozen=True)
class SceneConfig:
    """One BASELINE.json configuration (c1..c5) or a scaled-down variant of it."""

    config_id: int
    num_points: int
    width: int
    height: int
    sh_deg: int
    fy: float
    s0: float
    name: str = ""

    @property
    def tiles_x(self) -> int:
        return (self.width + 15) // 16

    @property
    def tiles_y(self) -> int:
        return (self.height + 15) // 16

    @property
    def total_tiles(self) -> int:
        return self.tiles_x * self.tiles_y


CONFIGS = {
    "c1": SceneConfig(1, 10_000, 256, 256, 0, 300.0, 0.005, "c1"),
    "c2": SceneConfig(2, 100_000, 640, 480, 1, 550.0, 0.003, "c2"),
    "c3": SceneConfig(3, 1_000_000, 1920, 1080, 3, 1200.0, 0.003, "c3-perf"),
    "c3-small": SceneConfig(3, 1_000_000, 1920, 1080, 3, 1200.0, 0.0005, "c3-small"),
    "c5": SceneConfig(5, 5_000_000, 3840, 2160, 3, 2400.0, 0.002, "c5"),
}


def f32_to_f16_bits(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even binary32 -> binary16 bit patterns (what ``Float16Array`` stores)."""
    return np.asarray(a, dtype=np.float32).astype(np.float16).view(np.uint16)


def make_gaussians(cfg: SceneConfig, num_points: int | None = None) -> tuple[np.ndarray, np.ndarray]:
    """Returns ``(gaussians uint32[N,6], sh uint32[N,24])`` for ``cfg`` (optionally the first ``num_points``)."""
    n = cfg.num_points if num_points is None else int(num_points)
    k_coef = (cfg.sh_deg + 1) ** 2
    draws = 3 + 8 + 2 + 3 + 3 + 2 * 3 * (k_coef - 1)
    u = _uniform01(_SEED_BASE + cfg.config_id, n * draws).reshape(n, draws)
    z = 2.0 + 8.0 * u[:, 0]
    half_w = z * cfg.width / (2.0 * cfg.fy)
    half_h = z * cfg.height / (2.0 * cfg.fy)
    x = (2.0 * u[:, 1] - 1.0) * half_w
    y = (2.0 * u[:, 2] - 1.0) * half_h
    q = np.stack([_box_muller(u[:, 3 + 2 * i], u[:, 4 + 2 * i]) for i in range(4)], axis=1)
    q = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
    opacity = _box_muller(u[:, 11], u[:, 12])
    log_s = math.log(cfg.s0) + u[:, 13:16] * math.log(10.0)
    half = np.zeros((n, 12), dtype=np.float32)
    half[:, 0], half[:, 1], half[:, 2], half[:, 3] = x, y, z, opacity
    half[:, 4:8] = q
    half[:, 8:11] = log_s
    gaussians = np.ascontiguousarray(f32_to_f16_bits(half)).view(np.uint32).reshape(n, 6)

    sh = np.zeros((n, 48), dtype=np.float32)
    sh[:, 0:3] = 2.0 * u[:, 16:19] - 1.0
    for j in range(3 * (k_coef - 1)):
        sh[:, 3 + j] = 0.1 * _box_muller(u[:, 19 + 2 * j], u[:, 20 + 2 * j])
    sh_words = np.ascontiguousarray(f32_to_f16_bits(sh)).view(np.uint32).reshape(n, 24)
    return gaussians, sh_words


def make_target_scene(gaussians: np.ndarray, sh: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """The ground-truth variant of a scene: ``opacity_raw + 1`` and ``DC + 0.2``, re-rounded to fp16."""
    g16 = gaussians.copy().view(np.uint16).reshape(-1, 12)
    g16[:, 3] = f32_to_f16_bits(g16[:, 3].view(np.float16).astype(np.float32) + 1.0)
    s16 = sh.copy().view(np.uint16).reshape(-1, 48)
    s16[:, 0:3] = f32_to_f16_bits(s16[:, 0:3].view(np.float16).astype(np.float32) + 0.2)
    return g16.view(np.uint32).reshape(-1, 6), s16.view(np.uint32).reshape(-1, 24)


def projection_matrix(width: int, height: int, fy: float, znear: float = 0.01, zfar: float = 100.0) -> np.ndarray:
    """``get_projection_matrix`` of ``src/camera/camera.ts:29-56`` (fx is ignored: SURVEY Q18). Column-major 4x4."""
    tan_y = (height * 0.5) / fy
    tan_x = (width * 0.5) / fy
    top = tan_y * znear
    right = tan_x * znear
    m = np.zeros(16, dtype=np.float64)
    m[0] = 2.0 * znear / (2.0 * right)
    m[5] = -2.0 * znear / (2.0 * top)
    m[10] = zfar / (zfar - znear)
    m[11] = 1.0
    m[14] = -(zfar * znear) / (zfar - znear)
    return m


def mat4_inverse(m_colmajor: np.ndarray) -> np.ndarray:
    """``mat4.inverse`` of wgpu-matrix 3.2.0 (the reference's dependency, ``package-lock.json:695-698``; not vendored in the
    reference tree, so its published cofactor algorithm is restated here), as called by ``Camera.update_buffer``
    (``src/camera/camera.ts:171,187``).  ``m_colmajor``: the 16 elements in storage order (column-major), evaluated in binary64
    exactly as JavaScript evaluates it on Float32Array operands; the caller's store to float32 is the final rounding."""
    m = [float(v) for v in np.asarray(m_colmajor, dtype=np.float64).reshape(16)]
    m00, m01, m02, m03, m10, m11, m12, m13, m20, m21, m22, m23, m30, m31, m32, m33 = m
    tmp0, tmp1, tmp2, tmp3 = m22 * m33, m32 * m23, m12 * m33, m32 * m13
    tmp4, tmp5, tmp6, tmp7 = m12 * m23, m22 * m13, m02 * m33, m32 * m03
    tmp8, tmp9, tmp10, tmp11 = m02 * m23, m22 * m03, m02 * m13, m12 * m03
    tmp12, tmp13, tmp14, tmp15 = m20 * m31, m30 * m21, m10 * m31, m30 * m11
    tmp16, tmp17, tmp18, tmp19 = m10 * m21, m20 * m11, m00 * m31, m30 * m01
    tmp20, tmp21, tmp22, tmp23 = m00 * m21, m20 * m01, m00 * m11, m10 * m01
    t0 = (tmp0 * m11 + tmp3 * m21 + tmp4 * m31) - (tmp1 * m11 + tmp2 * m21 + tmp5 * m31)
    t1 = (tmp1 * m01 + tmp6 * m21 + tmp9 * m31) - (tmp0 * m01 + tmp7 * m21 + tmp8 * m31)
    t2 = (tmp2 * m01 + tmp7 * m11 + tmp10 * m31) - (tmp3 * m01 + tmp6 * m11 + tmp11 * m31)
    t3 = (tmp5 * m01 + tmp8 * m11 + tmp11 * m21) - (tmp4 * m01 + tmp9 * m11 + tmp10 * m21)
    d = 1.0 / (m00 * t0 + m10 * t1 + m20 * t2 + m30 * t3)
    out = np.empty(16, dtype=np.float64)
    out[0], out[1], out[2], out[3] = d * t0, d * t1, d * t2, d * t3
    out[4] = d * ((tmp1 * m10 + tmp2 * m20 + tmp5 * m30) - (tmp0 * m10 + tmp3 * m20 + tmp4 * m30))
    out[5] = d * ((tmp0 * m00 + tmp7 * m20 + tmp8 * m30) - (tmp1 * m00 + tmp6 * m20 + tmp9 * m30))
    out[6] = d * ((tmp3 * m00 + tmp6 * m10 + tmp11 * m30) - (tmp2 * m00 + tmp7 * m10 + tmp10 * m30))
    out[7] = d * ((tmp4 * m00 + tmp9 * m10 + tmp10 * m20) - (tmp5 * m00 + tmp8 * m10 + tmp11 * m20))
    out[8] = d * ((tmp12 * m13 + tmp15 * m23 + tmp16 * m33) - (tmp13 * m13 + tmp14 * m23 + tmp17 * m33))
    out[9] = d * ((tmp13 * m03 + tmp18 * m23 + tmp21 * m33) - (tmp12 * m03 + tmp19 * m23 + tmp20 * m33))
    out[10] = d * ((tmp14 * m03 + tmp19 * m13 + tmp22 * m33) - (tmp15 * m03 + tmp18 * m13 + tmp23 * m33))
    out[11] = d * ((tmp17 * m03 + tmp20 * m13 + tmp23 * m23) - (tmp16 * m03 + tmp21 * m13 + tmp22 * m23))
    out[12] = d * ((tmp14 * m22 + tmp17 * m32 + tmp13 * m12) - (tmp16 * m32 + tmp12 * m12 + tmp15 * m22))
    out[13] = d * ((tmp20 * m32 + tmp12 * m02 + tmp19 * m22) - (tmp18 * m22 + tmp21 * m32 + tmp13 * m02))
    out[14] = d * ((tmp18 * m12 + tmp23 * m32 + tmp15 * m02) - (tmp22 * m32 + tmp14 * m02 + tmp19 * m12))
    out[15] = d * ((tmp22 * m22 + tmp16 * m02 + tmp21 * m12) - (tmp20 * m12 + tmp23 * m22 + tmp17 * m02))
    return out


def camera_block(view_rowmajor: np.ndarray, width: int, height: int, fy: float) -> np.ndarray:
    """Packs the 272-byte ``CameraUniforms`` block (``src/shaders/common.wgsl:1-8``) from a 4x4 world->view matrix."""
    view = np.asarray(view_rowmajor, dtype=np.float64).reshape(4, 4)
    proj_cm = projection_matrix(width, height, fy)
    out = np.zeros(68, dtype=np.float32)
    out[0:16] = view.T.reshape(-1)
    out[32:48] = proj_cm
    # the inverses are taken of the float32 matrices the uniform block stores, by wgpu-matrix's cofactor formula (camera.ts:171,187)
    out[16:32] = mat4_inverse(out[0:16])
    out[48:64] = mat4_inverse(out[32:48])
    out[64:66] = (width, height)
    out[66:68] = (fy, fy)
    return out


def identity_camera(cfg: SceneConfig) -> np.ndarray:
    return camera_block(np.eye(4), cfg.width, cfg.height, cfg.fy)


def circle_cameras(cfg: SceneConfig, count: int, radius: float = 1.0, target=(0.0, 0.0, 6.0)) -> np.ndarray:
    """``count`` cameras on a circle of ``radius`` around the origin looking at ``target`` (COLMAP axes: +x right, +y down, +z forward)."""
    cams = np.zeros((count, 68), dtype=np.float32)
    tgt = np.asarray(target, dtype=np.float64)
    for i in range(count):
        th = 2.0 * math.pi * i / count
        c = np.array([radius * math.cos(th), radius * math.sin(th), 0.0])
        f = tgt - c
        f /= np.linalg.norm(f)
        xr = np.cross(np.array([0.0, 1.0, 0.0]), f)
        xr /= np.linalg.norm(xr)
        yd = np.cross(f, xr)
        rot = np.stack([xr, yd, f], axis=0)
        view = np.eye(4)
        view[:3, :3] = rot
        view[:3, 3] = -rot @ c
        cams[i] = camera_block(view, cfg.width, cfg.height, cfg.fy)
    return cams


def render_settings(cfg: SceneConfig, width: int | None = None, height: int | None = None, gaussian_mode: float = 1.0) -> np.ndarray:
    """The 7-float ``RenderSettings`` block (``src/shaders/common.wgsl:10-18``, ``tiled-forward-pass.ts:174-182``)."""
    return np.array([1.0, cfg.sh_deg, width or cfg.width, height or cfg.height, 3.0, gaussian_mode, 128.0], dtype=np.float32)


def tile_info(width: int, height: int, max_tile_entries: int) -> np.ndarray:
    tx, ty = (width + 15) // 16, (height + 15) // 16
    return np.array([tx, ty, tx * ty, max_tile_entries], dtype=np.uint32)
