"""Float64 numpy model of the hot path's per-Gaussian mathematics, written from the TEXTBOOK formulas -- EWA splatting (Zwicker et
al.), the 3DGS projection (Kerbl et al.), real spherical harmonics from scipy's complex ones, Adam (Kingma & Ba) -- and from the
reference's documented conventions (data layouts, NDC / pixel mapping, rejection order).  It shares NO code with
``oracle/`` (the C++ restatement of the WGSL), ``webdgs_amd/csrc/wgslm.h`` or the kernels: no WGSL-shaped matrix type, no transliterated
statement order, float64 throughout, whole arrays at a time.  ``tests/test_oracle_independent.py`` uses it to check that the oracle --
against which every GPU result is compared bit for bit -- computes what the mathematics says.

Conventions taken from the reference's data contract (not from its arithmetic):
  * Gaussian row: 12 fp16 = x y z opacity_raw | quaternion w x y z | log-sigma x y z, pad     (common.wgsl:20-24)
  * SH row: 48 fp16, [k][rgb], k = l*l + l + m                                                   (tiled-forward.wgsl:64-86)
  * camera block: view, view_inv, proj, proj_inv as column-major 4x4, viewport, focal            (camera.ts:165-195)
  * pixel centre of an NDC point: ((0.5 ndc.x + 0.5) W, (-0.5 ndc.y + 0.5) H)                    (tiled-forward.wgsl:237)
"""
from __future__ import annotations

import numpy as np
import scipy.special


def f16_round(x):
    """IEEE round-to-nearest-even to binary16 and back (numpy's conversion)."""
    with np.errstate(over="ignore"):
        return np.asarray(x, np.float64).astype(np.float16).astype(np.float64)


def f16_bits(x) -> np.ndarray:
    with np.errstate(over="ignore"):
        return np.asarray(x, np.float64).astype(np.float16).view(np.uint16)


def f16_ulp_distance(bits_a, bits_b) -> np.ndarray:
    """Distance in units of the last place between two arrays of binary16 bit patterns (sign-magnitude -> ordered integers)."""
    def ordered(b):
        b = b.astype(np.int32)
        return np.where(b & 0x8000, -(b & 0x7FFF), b & 0x7FFF)
    return np.abs(ordered(np.asarray(bits_a)) - ordered(np.asarray(bits_b)))


def unpack_gaussians(g_u32: np.ndarray) -> dict:
    h = np.ascontiguousarray(g_u32).view(np.float16).reshape(-1, 12).astype(np.float64)
    return dict(pos=h[:, 0:3], opacity_raw=h[:, 3], quat=h[:, 4:8], log_scale=h[:, 8:11])


def unpack_sh(sh_u32: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(sh_u32).view(np.float16).reshape(-1, 16, 3).astype(np.float64)


def camera_parts(cam: np.ndarray) -> dict:
    c = np.asarray(cam, np.float64)
    view, proj = c[0:16].reshape(4, 4).T, c[32:48].reshape(4, 4).T   # column-major storage -> ordinary (row, column) matrices
    rot = view[:3, :3]
    return dict(view=view, proj=proj, rot=rot, cam_pos=-rot.T @ view[:3, 3], width=c[64], height=c[65], fx=c[66], fy=c[67])


def rotation_from_quaternion(q: np.ndarray) -> np.ndarray:
    """The usual rotation matrix of a quaternion (w, x, y, z), as every graphics text writes it for a unit quaternion; evaluated as that
    polynomial also when |q| != 1 (the renderer does not normalise, and the derivative with respect to q is taken of this expression)."""
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - w * z); R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y); R[..., 2, 1] = 2 * (y * z + w * x); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def real_sh_basis(dirs: np.ndarray, degree: int) -> np.ndarray:
    """Real spherical harmonics Y[k], k = l*l + l + m, l <= degree, of unit vectors, built from scipy's COMPLEX harmonics: sqrt(2) Re Y_l^m
    for m > 0, sqrt(2) Im Y_l^|m| for m < 0, Y_l^0 for m = 0, the Condon-Shortley phase of the complex functions kept (that is the sign
    convention of the 3DGS colour model: -y, +z, -x for l = 1)."""
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    theta, phi = np.arccos(np.clip(z, -1.0, 1.0)), np.arctan2(y, x)
    out = np.zeros((dirs.shape[0], (degree + 1) ** 2))
    for l in range(degree + 1):
        for m in range(-l, l + 1):
            Y = scipy.special.sph_harm_y(l, abs(m), theta, phi)
            out[:, l * l + l + m] = Y.real if m == 0 else (np.sqrt(2.0) * Y.real if m > 0 else np.sqrt(2.0) * Y.imag)
    return out


def project(gauss: dict, cam: dict, max_radius_px: float = 128.0) -> dict:
    """EWA projection of N Gaussians: view-space mean, NDC, pixel centre, 2D covariance (+0.3 px^2 low-pass), conic, opacity-aware
    axis-aligned extents sqrt(2 ln(128 sigma(o)) Sigma_ii) ("SnugBox"), everything in float64 and differentiable in the inputs."""
    pos, q, s = gauss["pos"], gauss["quat"], np.exp(gauss["log_scale"])
    t = pos @ cam["rot"].T + cam["view"][:3, 3]
    clip = np.concatenate([t, np.ones((t.shape[0], 1))], 1) @ cam["proj"].T
    with np.errstate(divide="ignore", invalid="ignore"):
        ndc = clip[:, :3] / clip[:, 3:4]
    W, H, fx, fy = cam["width"], cam["height"], cam["fx"], cam["fy"]
    px = np.stack([(0.5 * ndc[:, 0] + 0.5) * W, (-0.5 * ndc[:, 1] + 0.5) * H], 1)
    R = rotation_from_quaternion(q)
    cov3 = np.einsum("nij,nj,nkj->nik", R, s * s, R)                     # R diag(s^2) R^T
    limx, limy = 1.3 * (0.5 * W) / fx, 1.3 * (0.5 * H) / fy              # the 3DGS frustum clamp of the Jacobian's evaluation point
    tz = t[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        tx, ty = np.clip(t[:, 0] / tz, -limx, limx) * tz, np.clip(t[:, 1] / tz, -limy, limy) * tz
        J = np.zeros((t.shape[0], 2, 3))
        J[:, 0, 0], J[:, 0, 2] = fx / tz, -fx * tx / (tz * tz)
        J[:, 1, 1], J[:, 1, 2] = fy / tz, -fy * ty / (tz * tz)
    A = J @ cam["rot"]                                                   # d(pixel) / d(world)
    cov2 = A @ cov3 @ np.transpose(A, (0, 2, 1))
    a, b, c = cov2[:, 0, 0] + 0.3, cov2[:, 0, 1], cov2[:, 1, 1] + 0.3
    with np.errstate(divide="ignore", invalid="ignore"):
        det = a * c - b * b
        conic = np.stack([c / det, -b / det, a / det], 1)
        sig = sigmoid(gauss["opacity_raw"])
        tq = 2.0 * np.log(128.0 * sig)
        ext = np.sqrt(np.maximum(tq, 0.0)[:, None] * np.stack([a, c], 1))
    return dict(t=t, clip=clip, ndc=ndc, px=px, cov2=np.stack([a, b, c], 1), det=det, conic=conic, sigma=sig, tq=tq, extent=ext,
                extent_capped=np.minimum(ext, max_radius_px))


def colour(gauss: dict, sh: np.ndarray, cam: dict, degree: int) -> np.ndarray:
    d = gauss["pos"] - cam["cam_pos"]
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    Y = real_sh_basis(d, degree)
    return np.maximum(np.einsum("nk,nkc->nc", Y, sh[:, : (degree + 1) ** 2, :]) + 0.5, 0.0)


def tile_boxes(proj_out: dict, cam: dict) -> dict:
    """Visibility and the covered tile rectangle, following the reference's documented rejection order.  The box is
    taken from the fp16-rounded NDC centre and extents, because that is what the later stages see (A2)."""
    W, H = cam["width"], cam["height"]
    ntx, nty = int(np.ceil(W / 16)), int(np.ceil(H / 16))
    ndc, clip = proj_out["ndc"], proj_out["clip"]
    conic, det = proj_out["conic"], proj_out["det"]
    with np.errstate(invalid="ignore"):
        ok = clip[:, 3] != 0
        ok &= (np.abs(ndc[:, 0]) <= 1.2) & (np.abs(ndc[:, 1]) <= 1.2) & (ndc[:, 2] >= 0) & (ndc[:, 2] <= 1)
        ok &= det > 0
        ok &= (conic[:, 0] > 0) & (conic[:, 2] > 0) & (conic[:, 1] ** 2 - conic[:, 0] * conic[:, 2] < 0)
        ok &= proj_out["tq"] > 0
    n16 = f16_round(np.clip(np.nan_to_num(ndc[:, :2]), -60000, 60000))
    e16 = f16_round(np.nan_to_num(proj_out["extent_capped"]))
    pc = np.stack([(0.5 * n16[:, 0] + 0.5) * W, (-0.5 * n16[:, 1] + 0.5) * H], 1)
    lo, hi = pc - e16 - 2.0, pc + e16 + 2.0
    ok &= ~((hi[:, 0] < 0) | (hi[:, 1] < 0) | (lo[:, 0] >= W) | (lo[:, 1] >= H))
    bmin = np.maximum(lo, 0.0)
    bmax = np.minimum(hi, np.array([W - 1.0, H - 1.0]))
    ok &= ~((bmax[:, 0] < bmin[:, 0]) | (bmax[:, 1] < bmin[:, 1]))
    tmin = np.floor(np.maximum(bmin, 0)).astype(np.int64) // 16
    tmax = np.minimum(np.floor(np.maximum(bmax, 0)).astype(np.int64) // 16, np.array([ntx - 1, nty - 1]))
    count = np.maximum(tmax[:, 0] - tmin[:, 0] + 1, 0) * np.maximum(tmax[:, 1] - tmin[:, 1] + 1, 0)
    ok &= count <= 2048
    return dict(visible=ok, tile_min=tmin, tile_max=tmax, count=np.where(ok, count, 0), ndc16=n16, ext16=e16)


def depth_key16(view_z: np.ndarray) -> np.ndarray:
    """Top 16 bits of the order-preserving unsigned image of the float32 view depth."""
    bits = np.asarray(view_z, np.float32).view(np.uint32)
    ordered = np.where(bits & 0x80000000, ~bits, bits ^ np.uint32(0x80000000)).astype(np.uint32)
    return ordered >> 16


# ----------------------------------------------------------------------------- Adam (no bias correction: SURVEY Q14)
def adam_update(param, grad, m, v, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    m2 = beta1 * m + (1 - beta1) * grad
    v2 = beta2 * v + (1 - beta2) * grad * grad
    return param - lr * m2 / (np.sqrt(v2) + eps), m2, v2


# ----------------------------------------------------------------------------- densify helpers
def lowbias32(x: np.ndarray) -> np.ndarray:
    """Chris Wellons' lowbias32 integer hash (published constants), on uint32 arrays with wrapping arithmetic."""
    x = np.asarray(x, np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x.astype(np.uint32)
