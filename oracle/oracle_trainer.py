"""ORACLE (test infrastructure only; PARITY UNPINNED) -- CPU restatement of the reference's ``Trainer`` sequencing.

Follows ``/root/reference/src/trainer.ts``: ``step()`` 568-660 (one view: forward, rasterize, backward, Adam, re-pack; the densify
schedule 593-601 checked on ``iteration + 1``), ``runDensifyPruneMultiView()`` 373-497 (metric views at
``floor(size / metricDownscale)``, ``clear: false`` accumulation, integer division by the views used, decide / cap / total, the
``outN == 0 || outN == inN`` early-out, scatter, optimizer iteration carried over) and ``applyPointCloudSwap`` 201-237.  The metric
camera is ``Camera.set_preset`` + ``on_update_canvas`` + ``update_buffer`` (``src/camera/camera.ts:138-205``) applied to the view's
pose with the canvas resized to the metrics resolution.

Two deliberate differences from the reference, both host-side and both shared with the product:
* ``Math.random()`` cannot be reproduced, so the training and metric view indices are arguments;
* every metric view renders with its OWN camera (the reference uploads all of them before the one submit, so all its metric views
  render with the last camera: SURVEY Q12).

``step(view_ids)`` with more than one view is this repo's view-batched step (no counterpart in the reference): each
"rank" sums its views' unpacked gradients in fp32 in view order, the rank sums are added in rank order, one Adam.

Only ``tests/`` may import this module.
"""
from __future__ import annotations

import math

import numpy as np

from . import oracle as orc

DENSIFY_DEFAULTS = dict(schedule=dict(enabled=True, warmupIterations=500, interval=100, stopIterations=15_000), metricViews=10, metricDownscale=2,
                        metricThreshold=0.5, maxBufferBytes=128 * 1024 * 1024, maxNewPointsPerStep=5000, pruneOpacity=0.01, cloneThresholdCount=500,
                        splitScaleThreshold=1.0)  # trainer.ts:147-164


def _mat4_inverse(m):
    """wgpu-matrix 3.2.0 ``mat4.inverse`` (cofactor expansion, binary64 on the Float32Array's values); written out independently
    of the product's copy (webdgs_amd/synth.py) as plain 4x4 cofactors: inverse = adj(M) / det(M)."""
    a = np.asarray(m, np.float64).reshape(4, 4).T  # a[row][col]
    out = np.empty((4, 4), np.float64)
    m00, m01, m02, m03 = a[0][0], a[1][0], a[2][0], a[3][0]  # wgpu-matrix names m<col><row>
    m10, m11, m12, m13 = a[0][1], a[1][1], a[2][1], a[3][1]
    m20, m21, m22, m23 = a[0][2], a[1][2], a[2][2], a[3][2]
    m30, m31, m32, m33 = a[0][3], a[1][3], a[2][3], a[3][3]
    tmp = [m22 * m33, m32 * m23, m12 * m33, m32 * m13, m12 * m23, m22 * m13, m02 * m33, m32 * m03, m02 * m23, m22 * m03, m02 * m13, m12 * m03,
           m20 * m31, m30 * m21, m10 * m31, m30 * m11, m10 * m21, m20 * m11, m00 * m31, m30 * m01, m00 * m21, m20 * m01, m00 * m11, m10 * m01]
    t0 = (tmp[0] * m11 + tmp[3] * m21 + tmp[4] * m31) - (tmp[1] * m11 + tmp[2] * m21 + tmp[5] * m31)
    t1 = (tmp[1] * m01 + tmp[6] * m21 + tmp[9] * m31) - (tmp[0] * m01 + tmp[7] * m21 + tmp[8] * m31)
    t2 = (tmp[2] * m01 + tmp[7] * m11 + tmp[10] * m31) - (tmp[3] * m01 + tmp[6] * m11 + tmp[11] * m31)
    t3 = (tmp[5] * m01 + tmp[8] * m11 + tmp[11] * m21) - (tmp[4] * m01 + tmp[9] * m11 + tmp[10] * m21)
    d = 1.0 / (m00 * t0 + m10 * t1 + m20 * t2 + m30 * t3)
    flat = [d * t0, d * t1, d * t2, d * t3,
            d * ((tmp[1] * m10 + tmp[2] * m20 + tmp[5] * m30) - (tmp[0] * m10 + tmp[3] * m20 + tmp[4] * m30)),
            d * ((tmp[0] * m00 + tmp[7] * m20 + tmp[8] * m30) - (tmp[1] * m00 + tmp[6] * m20 + tmp[9] * m30)),
            d * ((tmp[3] * m00 + tmp[6] * m10 + tmp[11] * m30) - (tmp[2] * m00 + tmp[7] * m10 + tmp[10] * m30)),
            d * ((tmp[4] * m00 + tmp[9] * m10 + tmp[10] * m20) - (tmp[5] * m00 + tmp[8] * m10 + tmp[11] * m20)),
            d * ((tmp[12] * m13 + tmp[15] * m23 + tmp[16] * m33) - (tmp[13] * m13 + tmp[14] * m23 + tmp[17] * m33)),
            d * ((tmp[13] * m03 + tmp[18] * m23 + tmp[21] * m33) - (tmp[12] * m03 + tmp[19] * m23 + tmp[20] * m33)),
            d * ((tmp[14] * m03 + tmp[19] * m13 + tmp[22] * m33) - (tmp[15] * m03 + tmp[18] * m13 + tmp[23] * m33)),
            d * ((tmp[17] * m03 + tmp[20] * m13 + tmp[23] * m23) - (tmp[16] * m03 + tmp[21] * m13 + tmp[22] * m23)),
            d * ((tmp[14] * m22 + tmp[17] * m32 + tmp[13] * m12) - (tmp[16] * m32 + tmp[12] * m12 + tmp[15] * m22)),
            d * ((tmp[20] * m32 + tmp[12] * m02 + tmp[19] * m22) - (tmp[18] * m22 + tmp[21] * m32 + tmp[13] * m02)),
            d * ((tmp[18] * m12 + tmp[23] * m32 + tmp[15] * m02) - (tmp[22] * m32 + tmp[14] * m02 + tmp[19] * m12)),
            d * ((tmp[22] * m22 + tmp[16] * m02 + tmp[21] * m12) - (tmp[20] * m12 + tmp[23] * m22 + tmp[17] * m02))]
    del out
    return np.array(flat, np.float64)


def metrics_camera(camera_block: np.ndarray, width: int, height: int, znear: float = 0.01, zfar: float = 100.0) -> np.ndarray:
    """The 68-float block of the METRICS camera for a training view given as a camera block (view matrix, viewport, focal).

    ``set_preset`` (camera.ts:196-205): ``fovY = 2 atan(preset.height / (2 preset.fy))`` with the view's own height and fy;
    ``on_update_canvas`` (138-147): ``focal = 0.5 canvas.height / tan(fovY / 2)``, ``fovX = 2 atan(canvas.width / (2 focal))``,
    viewport = canvas size; ``update_buffer`` (165-195): view unchanged, projection of ``get_projection_matrix`` (29-56), both
    inverses by wgpu-matrix."""
    cam = np.asarray(camera_block, np.float32)
    preset_h, preset_fy = float(cam[65]), float(cam[67])
    fov_y = 2.0 * math.atan(preset_h / (2.0 * preset_fy))
    focal = 0.5 * height / math.tan(fov_y * 0.5)
    fov_x = 2.0 * math.atan(width / (2.0 * focal))
    tan_y, tan_x = math.tan(fov_y / 2.0), math.tan(fov_x / 2.0)
    top, right = tan_y * znear, tan_x * znear
    proj = np.zeros(16, np.float64)  # column-major, after the transpose of camera.ts:49
    proj[0] = 2.0 * znear / (2.0 * right)
    proj[5] = -2.0 * znear / (2.0 * top)
    proj[10] = zfar / (zfar - znear)
    proj[11] = 1.0
    proj[14] = -(zfar * znear) / (zfar - znear)
    out = np.zeros(68, np.float32)
    out[0:16] = cam[0:16]
    out[32:48] = proj
    out[16:32] = _mat4_inverse(out[0:16])
    out[48:64] = _mat4_inverse(out[32:48])
    out[64:68] = (width, height, focal, focal)
    return out


def _settings(sh_deg, w, h):
    return np.array([1.0, sh_deg, w, h, 3.0, 1.0, 128.0], np.float32)  # tiled-forward-pass.ts:174-182


def _tile_info(w, h):
    tx, ty = (w + 15) // 16, (h + 15) // 16
    return np.array([tx, ty, tx * ty, 0], np.uint32)


class OracleTrainer:
    def __init__(self, gaussians, sh, sh_deg, cameras, images, training_config=None, adam=None, densify=None):
        """``cameras``: 68-float blocks; ``images``: rgba8 arrays ``[H, W, 4]`` (``setDataset``, trainer.ts:240-247)."""
        self.g, self.sh, self.sh_deg = gaussians.copy(), sh.copy(), int(sh_deg)
        self.cameras = [np.asarray(c, np.float32) for c in cameras]
        self.images = [np.ascontiguousarray(i) for i in images]
        self.tcfg = orc.training_config(**(training_config or {}))
        self.acfg = orc.ADAM_DEFAULT.copy() if adam is None else np.asarray(adam, np.float32)
        d = {**DENSIFY_DEFAULTS, **(densify or {})}
        d["schedule"] = {**DENSIFY_DEFAULTS["schedule"], **((densify or {}).get("schedule") or {})}
        self.densify = d
        self.state = orc.unpack(self.g, self.sh)  # new Optimizer(...) -> initBuffers (optimizer.ts:145-253)
        self.iteration = 0            # Trainer.iteration
        self.optimizer_iteration = 0  # Optimizer.iteration, survives the swap (trainer.ts:492-495)
        self.last_densify_iteration = None
        self.last = None              # stages of the last view processed (for stage-by-stage comparisons)
        self.last_densify = None

    @property
    def num_points(self):
        return int(self.g.shape[0])

    def should_densify(self):  # trainer.ts:593-601
        s = self.densify["schedule"]
        nxt = self.iteration + 1
        warm, interval, stop = s["warmupIterations"], max(1, s["interval"]), s["stopIterations"]
        return bool(s["enabled"] and warm <= nxt <= stop and (nxt == warm or (nxt - warm) % interval == 0))

    def step(self, view_ids, world: int = 1, metric_view_ids=None):
        """One ``Trainer.step()``.  ``view_ids``: one index (the reference) or the global batch of the view-batched extension, dealt
        to ``world`` ranks round-robin.  ``metric_view_ids``: the views ``runDensifyPruneMultiView`` draws if this step densifies."""
        view_ids = [int(view_ids)] if np.isscalar(view_ids) else [int(v) for v in view_ids]
        densify_now = self.should_densify()
        h, w = self.images[view_ids[0]].shape[:2]
        st, ti = _settings(self.sh_deg, w, h), _tile_info(w, h)
        if len(view_ids) == 1:
            self.last = orc.train_step(self.g, self.sh, self.state, self.cameras[view_ids[0]], st, ti, self.images[view_ids[0]], self.tcfg, self.acfg)
        else:
            n = self.num_points
            total = np.zeros((n, 14), np.float32)
            visible = np.zeros(n, np.uint32)
            for r in range(world):
                acc = None
                for v in view_ids[r::world]:
                    fw = orc.view_gradients(self.g, self.sh, self.cameras[v], st, ti, self.images[v], self.tcfg)
                    vis = fw["tile_counts"] > 0
                    gf = orc.unpack_gradients_f32(fw["gradients"])
                    gf[~vis] = 0.0
                    if acc is None:
                        acc = gf  # wdgs_store_gradients: the first view overwrites
                    else:
                        acc[vis] = acc[vis] + gf[vis]  # wdgs_accumulate_gradients: fp32 adds in view order, visible Gaussians only
                    visible += vis.astype(np.uint32)
                    self.last = fw
                if acc is not None:
                    total = acc if r == 0 else total + acc  # the exchange: rank sums added in rank order
            orc.adam_f32(self.acfg, visible, np.ascontiguousarray(total), self.state)
            orc.repack(self.state, self.g, self.sh)
        self.optimizer_iteration += 1
        self.iteration += 1
        if densify_now:
            if metric_view_ids is None:
                raise ValueError(f"iteration {self.iteration} densifies: metric_view_ids required")
            self.run_densify_prune_multi_view(metric_view_ids, w, h)

    def run_densify_prune_multi_view(self, metric_view_ids, base_w, base_h):
        """trainer.ts:373-497 with the drawn view indices given (entries whose image size differs from the base are skipped as
        in lines 395-396 and do not count as used)."""
        d = self.densify
        down = max(1, int(d["metricDownscale"]))
        mw, mh = max(1, base_w // down), max(1, base_h // down)
        views_target = max(1, int(d["metricViews"]))
        n = self.num_points
        counts = np.zeros(n, np.uint32)  # encoder.clearBuffer(metricCounts)
        mst, mti = _settings(self.sh_deg, mw, mh), _tile_info(mw, mh)
        bst = mst.copy()
        bst[5] = 0.0
        used, attempts, per_view = 0, 0, []
        for idx in metric_view_ids:
            if attempts >= views_target * 4 or used >= views_target:
                break
            attempts += 1
            img = self.images[idx]
            if img.shape[1] != base_w or img.shape[0] != base_h:
                continue
            mcam = metrics_camera(self.cameras[idx], mw, mh)
            fw = orc.forward(self.g, self.sh, mcam, mst, mti)
            gt_small = orc.downsample_bilinear(img, mw, mh)
            err, mm, flags = orc.metric_map(fw["rgba8"], gt_small, float(d["metricThreshold"]))
            cap = max(int(fw["total_entries"]), 1)
            orc.metric_count(bst, fw["tile_ranges"], np.ascontiguousarray(fw["sorted_values"][:cap]), cap, fw["splats"], flags, fw["n_contrib"], counts)
            per_view.append(dict(view=idx, camera=mcam, rgba8=fw["rgba8"], gt_small=gt_small, flags=flags, minmax=mm, counts_after=counts.copy()))
            used += 1
        self.last_densify = dict(used_views=used, per_view=per_view, rebuilt=False)
        if used == 0:
            return
        raw = counts.copy()
        orc.metric_normalize(counts, used)
        # computeMaxOutPoints (densify-prune.ts:390-410): 24-byte Gaussians, 96-byte SH rows
        max_bytes = max(0, int(d["maxBufferBytes"]))
        max_out = min(max_bytes // 24, max_bytes // 96)
        max_new = max(0, int(d["maxNewPointsPerStep"]))
        if max_new > 0:
            max_out = min(max_out, max(1, n) + max_new)
        prep = orc.densify_prepare(self.g, counts, max_out, clone_threshold=int(d["cloneThresholdCount"]), prune_opacity=float(d["pruneOpacity"]),
                                   split_scale=float(d["splitScaleThreshold"]))
        out_n = min(prep["total"], max_out)
        self.last_densify.update(counts_raw=raw, counts=counts, prepared=prep, max_out=max_out, out_n=out_n)
        if out_n == 0 or out_n == n:
            return
        og, osh, ost = orc.densify_scatter(self.g, self.sh, self.state, prep, out_n, reset_new_state=True)
        self.g, self.sh, self.state = og, osh, ost  # requestPointCloudSwap + applyPointCloudSwap: the new Optimizer adopts the state
        self.last_densify_iteration = self.iteration
        self.last_densify["rebuilt"] = True
