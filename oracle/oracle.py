"""ORACLE (test infrastructure only; PARITY UNPINNED).

ctypes front-end of ``oracle/liboracle.so`` -- the CPU restatement of the reference's WGSL kernels.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product (``webdgs_amd``) never does.  Build with ``make -C oracle``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("oracle_forward.cpp", "oracle_backward.cpp", "oracle_densify.cpp", "wgsl_shim.hpp")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.orc_emit.restype = ctypes.c_uint32
        _lib.orc_densify_total.restype = ctypes.c_uint32
    return _lib


def use_native_build() -> bool:
    """Switches this process to ``liboracle_native.so`` (same sources, ``-march=native``), built on the spot: the build that
    bench.py times as ``cpu_baseline``.  Falls back to the portable build (returns False) if the compile fails."""
    global _lib
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle_native.so"], stdout=subprocess.DEVNULL)
        l = ctypes.CDLL(os.path.join(_HERE, "liboracle_native.so"))
        l.orc_emit.restype = ctypes.c_uint32
        l.orc_densify_total.restype = ctypes.c_uint32
        _lib = l
        return True
    except Exception:
        return False


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def _u32(v):
    return ctypes.c_uint32(int(v))


def _f32(v):
    return ctypes.c_float(float(v))


# ----------------------------------------------------------------------------- forward
def project_count(gaussians, sh, camera, settings, tinfo, splats=None, depths=None):
    n = gaussians.shape[0]
    splats = np.zeros((n, 6), np.uint32) if splats is None else splats
    depths = np.zeros(n, np.uint32) if depths is None else depths
    counts = np.zeros(n, np.uint32)
    stats = np.zeros(4, np.uint32)
    lib().orc_project_count(_u32(n), _p(gaussians), _p(sh), _p(camera), _p(settings), _p(tinfo), _p(splats), _p(depths), _p(counts), _p(stats))
    return splats, depths, counts, stats


def exclusive_scan(values):
    out = np.zeros_like(values)
    lib().orc_exclusive_scan(_u32(values.shape[0]), _p(values), _p(out))
    return out


def emit(splats, depths, counts, offsets, settings, tinfo, capacity):
    keys = np.zeros(capacity, np.uint32)
    vals = np.zeros(capacity, np.uint32)
    dropped = lib().orc_emit(_u32(splats.shape[0]), _p(splats), _p(depths), _p(counts), _p(offsets), _p(settings), _p(tinfo), _p(keys), _p(vals), _u32(capacity))
    return keys, vals, int(dropped)


def sort_pairs(keys, vals, count):
    k = keys.copy()
    v = vals.copy()
    lib().orc_sort_pairs(_u32(count), _p(k), _p(v))
    return k, v


def tile_ranges(sorted_keys, total_entries, total_tiles):
    r = np.zeros(total_tiles + 1, np.uint32)
    lib().orc_tile_ranges(_u32(total_entries), _p(sorted_keys), _u32(total_tiles), _p(r))
    return r


def rasterize(settings, tinfo, splats, ranges, sorted_keys, sorted_vals, total_entries, max_batches=0):
    w, h = int(settings[2]), int(settings[3])
    rgba = np.zeros((h, w, 4), np.uint8)
    alpha = np.zeros((h, w), np.float32)
    ncontrib = np.zeros((h, w), np.uint32)
    lib().orc_rasterize(_p(settings), _p(tinfo), _p(splats), _u32(splats.shape[0]), _p(ranges), _p(sorted_keys), _p(sorted_vals),
                        _u32(total_entries), _u32(max_batches), _p(rgba), _p(alpha), _p(ncontrib))
    return rgba, alpha, ncontrib


def forward(gaussians, sh, camera, settings, tinfo, capacity=None, max_batches=0):
    """K1..K14 in the order of ``TiledForwardPass.encode`` + ``TiledRasterizer.encode``; returns every stage."""
    splats, depths, counts, stats = project_count(gaussians, sh, camera, settings, tinfo)
    offsets = exclusive_scan(counts)
    n = counts.shape[0]
    total = int(offsets[-1]) + int(counts[-1]) if n else 0
    stats[0] = total
    cap = total if capacity is None else capacity
    keys, vals, dropped = emit(splats, depths, counts, offsets, settings, tinfo, max(cap, 1))
    e = min(total, cap)
    skeys, svals = sort_pairs(keys, vals, e)
    ranges = tile_ranges(skeys, e, int(tinfo[2]))
    rgba, alpha, ncontrib = rasterize(settings, tinfo, splats, ranges, skeys, svals, e, max_batches)
    return dict(splats=splats, depths=depths, tile_counts=counts, tile_offsets=offsets, stats=stats, keys=keys, values=vals,
                sorted_keys=skeys, sorted_values=svals, total_entries=e, dropped=dropped, tile_ranges=ranges, rgba8=rgba,
                final_T=alpha, n_contrib=ncontrib)


# ----------------------------------------------------------------------------- backward
def training_config(lambda_l1=0.8, lambda_l2=0.0, lambda_dssim=0.2, c1=0.01 * 0.01, c2=0.03 * 0.03):
    return np.array([lambda_l1, lambda_l2, lambda_dssim, c1, c2], np.float32)


def loss_grad(pred, targ, cfg):
    h, w = pred.shape[:2]
    out = np.zeros((h, w, 4), np.float32)
    lib().orc_loss_grad(_u32(w), _u32(h), _p(pred), _p(targ), _p(cfg), _p(out))
    return out


def backward_rasterize(settings, n, ranges, sorted_vals, splats, final_t, ncontrib, lossgrad):
    gm = np.zeros(2 * n, np.int32)
    gc = np.zeros(4 * n, np.int32)
    go = np.zeros(n, np.int32)
    gcol = np.zeros(3 * n, np.int32)
    lib().orc_backward_rasterize(_p(settings), _p(ranges), _p(sorted_vals), _p(splats), _p(final_t), _p(ncontrib), _p(lossgrad),
                                 _p(gm), _p(gc), _p(go), _p(gcol))
    return gm, gc, go, gcol


def geometry_backward(camera, settings, gaussians, gm, gc, go, gcol):
    n = gaussians.shape[0]
    grads = np.zeros((n, 8), np.uint32)
    lib().orc_geometry_backward(_u32(n), _p(camera), _p(settings), _p(gaussians), _p(gm), _p(gc), _p(go), _p(gcol), _p(grads))
    return grads


ADAM_DEFAULT = np.array([0.00016, 0.0025, 0.05, 0.005, 0.001, 0.9, 0.999, 1e-8], np.float32)  # adam-config.ts:12-21


def new_optimizer_state(n):
    return dict(opt_pos=np.zeros((n, 12), np.float32), opt_rot=np.zeros((n, 12), np.float32), opt_scale=np.zeros((n, 12), np.float32),
                opt_opacity=np.zeros((n, 3), np.float32), param_sh=np.zeros((n, 48), np.float32), state_sh=np.zeros((n, 96), np.float32))


def unpack(gaussians, sh):
    n = gaussians.shape[0]
    st = new_optimizer_state(n)
    lib().orc_unpack(_u32(n), _p(gaussians), _p(sh), _p(st["opt_pos"]), _p(st["opt_rot"]), _p(st["opt_scale"]), _p(st["opt_opacity"]), _p(st["param_sh"]))
    return st


def adam(cfg, tile_counts, grads, st):
    n = tile_counts.shape[0]
    lib().orc_adam(_u32(n), _p(cfg), _p(tile_counts), _p(grads), _p(st["opt_pos"]), _p(st["opt_rot"]), _p(st["opt_scale"]),
                   _p(st["opt_opacity"]), _p(st["param_sh"]), _p(st["state_sh"]))


def adam_f32(cfg, visible_counts, grad_f32, st):
    """The view-batched Adam this repo adds: fp32 gradient sums [N,14] + u32 visibility counts."""
    n = visible_counts.shape[0]
    lib().orc_adam_f32(_u32(n), _p(cfg), _p(visible_counts), _p(grad_f32), _p(st["opt_pos"]), _p(st["opt_rot"]), _p(st["opt_scale"]),
                       _p(st["opt_opacity"]), _p(st["param_sh"]), _p(st["state_sh"]))


def set_literal_order(on: bool) -> None:
    """Evaluate q / power / C += c*alpha*vis as the WGSL source is parenthesised (no FMA) instead of the pinned FMA order."""
    lib().orc_set_literal_order(ctypes.c_int(1 if on else 0))


def set_k17_fix(flags: int) -> None:
    """TEST-ONLY: undo SURVEY Q10 (bit 0) and / or Q11 (bit 1) inside the oracle's K17 (oracle_backward.cpp); 0 restores the reference's K17."""
    lib().orc_set_k17_fix(ctypes.c_int(int(flags)))


def unpack_gradients_f32(grads):
    """GaussianGradient[N] (8 u32 of fp16 pairs) -> f32[N,14] in component order pos3, opacity, rot4, log-sigma3, rgb3 (exact)."""
    h = np.ascontiguousarray(grads).view(np.float16).reshape(-1, 16).astype(np.float32)
    return np.ascontiguousarray(h[:, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 14]])


def repack(st, gaussians, sh):
    n = gaussians.shape[0]
    lib().orc_repack(_u32(n), _p(st["opt_pos"]), _p(st["opt_rot"]), _p(st["opt_scale"]), _p(st["opt_opacity"]), _p(st["param_sh"]), _p(gaussians), _p(sh))


def view_gradients(gaussians, sh, camera, settings, tinfo, target_rgba8, tcfg=None, max_batches=0):
    """K1..K17 of one view (trainer.ts:606-628): every forward stage plus the packed GaussianGradients; nothing is updated."""
    tcfg = training_config() if tcfg is None else tcfg
    fw = forward(gaussians, sh, camera, settings, tinfo, max_batches=max_batches)
    lg = loss_grad(fw["rgba8"], target_rgba8, tcfg)
    bsettings = settings.copy()
    bsettings[5] = 0.0
    n = gaussians.shape[0]
    gm, gc, go, gcol = backward_rasterize(bsettings, n, fw["tile_ranges"], fw["sorted_values"], fw["splats"], fw["final_T"], fw["n_contrib"], lg)
    grads = geometry_backward(camera, bsettings, gaussians, gm, gc, go, gcol)
    fw.update(loss_grad=lg, grad_means=gm, grad_conics=gc, grad_opacity=go, grad_colors=gcol, gradients=grads)
    return fw


def train_step(gaussians, sh, st, camera, settings, tinfo, target_rgba8, tcfg=None, acfg=None, max_batches=0):
    """One ``Trainer.step()`` (trainer.ts:568-660): forward, rasterize, loss, backward, Adam, re-pack (in place)."""
    tcfg = training_config() if tcfg is None else tcfg
    acfg = ADAM_DEFAULT if acfg is None else acfg
    fw = forward(gaussians, sh, camera, settings, tinfo, max_batches=max_batches)
    lg = loss_grad(fw["rgba8"], target_rgba8, tcfg)
    bsettings = settings.copy()
    bsettings[5] = 0.0  # tiled-backward-pass.ts:150-159 builds its own settings block with gaussian_mode = 0
    n = gaussians.shape[0]
    gm, gc, go, gcol = backward_rasterize(bsettings, n, fw["tile_ranges"], fw["sorted_values"], fw["splats"], fw["final_T"], fw["n_contrib"], lg)
    grads = geometry_backward(camera, bsettings, gaussians, gm, gc, go, gcol)
    adam(acfg, fw["tile_counts"], grads, st)
    repack(st, gaussians, sh)
    fw.update(loss_grad=lg, grad_means=gm, grad_conics=gc, grad_opacity=go, grad_colors=gcol, gradients=grads)
    return fw


# ----------------------------------------------------------------------------- densify
def downsample_bilinear(src, dst_w, dst_h):
    sh_, sw_ = src.shape[:2]
    dst = np.zeros((dst_h, dst_w, 4), np.uint8)
    lib().orc_downsample_bilinear(_u32(sw_), _u32(sh_), _p(src), _u32(dst_w), _u32(dst_h), _p(dst))
    return dst


def metric_map(pred, targ, threshold, err_scale=1_000_000.0):
    h, w = pred.shape[:2]
    err = np.zeros((h, w), np.uint32)
    mm = np.zeros(2, np.uint32)
    flags = np.zeros((h, w), np.uint32)
    lib().orc_metric_map(_u32(w), _u32(h), _p(pred), _p(targ), _f32(err_scale), _f32(threshold), _p(err), _p(mm), _p(flags))
    return err, mm, flags


def metric_count(settings, ranges, sorted_vals, num_instances, splats, flags, ncontrib, counts):
    lib().orc_metric_count(_p(settings), _p(ranges), _p(sorted_vals), _u32(num_instances), _p(splats), _u32(splats.shape[0]), _p(flags),
                           _p(ncontrib), _p(counts), _u32(counts.shape[0]))


def metric_normalize(counts, divisor):
    lib().orc_metric_normalize(_u32(counts.shape[0]), _u32(divisor), _p(counts))


def densify_prepare(gaussians, metric_counts, max_out, clone_threshold=500, prune_opacity=0.01, split_scale=1.0):
    """``DensifyPrunePass.encodePrepare`` (densify-prune.ts:458-468): decide, scan, cap, scan, total."""
    n = gaussians.shape[0]
    counts = np.zeros(n, np.uint32)
    actions = np.zeros(n, np.uint32)
    lib().orc_densify_decide(_u32(n), _p(gaussians), _p(metric_counts), _u32(clone_threshold), _f32(prune_opacity), _f32(split_scale), _p(counts), _p(actions))
    pre = exclusive_scan(counts)
    lib().orc_densify_cap(_u32(n), _u32(max_out), _p(pre), _p(counts), _p(actions))
    offsets = exclusive_scan(counts)
    total = int(lib().orc_densify_total(_u32(n), _p(offsets), _p(counts)))
    return dict(actions=actions, counts=counts, offsets=offsets, total=total)


def densify_scatter(gaussians, sh, st, prep, out_n, reset_new_state=True):
    n = gaussians.shape[0]
    og = np.zeros((out_n, 6), np.uint32)
    osh = np.zeros((out_n, 24), np.uint32)
    lib().orc_scatter_gaussians(_u32(n), _u32(out_n), _p(gaussians), _p(sh), _p(prep["offsets"]), _p(prep["counts"]), _p(prep["actions"]), _p(og), _p(osh))
    ost = new_optimizer_state(out_n)
    lib().orc_scatter_optimizer(_u32(n), _u32(out_n), _u32(1 if reset_new_state else 0), _p(prep["offsets"]), _p(prep["counts"]), _p(prep["actions"]),
                                _p(st["opt_pos"]), _p(st["opt_rot"]), _p(st["opt_scale"]), _p(st["opt_opacity"]), _p(st["param_sh"]), _p(st["state_sh"]),
                                _p(ost["opt_pos"]), _p(ost["opt_rot"]), _p(ost["opt_scale"]), _p(ost["opt_opacity"]), _p(ost["param_sh"]), _p(ost["state_sh"]))
    return og, osh, ost
