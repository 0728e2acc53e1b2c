"""CPU: implementation-independent evidence for the oracle.

The reference ships no test vector, so the C++ oracle -- against which every GPU result is compared bit for bit -- is
"parity unpinned".  What CAN be pinned is that the oracle computes what the mathematics says.  ``tests/indep64.py`` restates the
per-Gaussian mathematics from the textbook formulas in float64 numpy (no code shared with ``oracle/``, ``wgslm.h`` or the kernels);
this file compares:

  (a) K1  projection, conic, SnugBox extents, SH colour (degrees 0..3), depth key, tile counts           -- test_k1_*
  (b) K17 the chain rule to position / log-scale / quaternion / raw opacity, against central differences of the float64 forward;
          the reference's K17 deviates from the true gradient in exactly the places named in the test                -- test_k17_*
  (c) K18 Adam without bias correction, quaternion renormalisation, visibility skip                                 -- test_adam_*
  (d) K26-K30 densify decisions, capacity rule, offsets, clone / split children                                     -- test_densify_*

K14 / K16 have their float64 checks in tests/test_oracle_golden.py (re-composite, finite differences of the composite)."""
import numpy as np
import pytest

from webdgs_amd import synth

import harness
import indep64 as ind


def _scene(base="c1", n=None, w=None, h=None, sh_deg=None, s0=None, view=None):
    cfg = harness.small_config(base, num_points=n, width=w, height=h, sh_deg=sh_deg, s0=s0)
    g, sh = synth.make_gaussians(cfg)
    cam = synth.identity_camera(cfg) if view is None else synth.circle_cameras(cfg, 8)[view]
    return cfg, g, sh, cam


# ----------------------------------------------------------------------------------------------------------------- (a) K1
@pytest.mark.parametrize("case", [dict(base="c1"), dict(base="c1", view=3), dict(base="c2", n=6000, sh_deg=1, view=5), dict(base="c2", n=6000, w=320, h=240, sh_deg=2, view=1),
                                  dict(base="c3", n=8000, w=480, h=272, sh_deg=3, view=6, s0=0.01)],
                         ids=["c1-identity", "c1-rotated", "sh1", "sh2", "sh3"])
def test_k1_projection_extents_colour_and_tile_counts(orc, case):
    cfg, g, sh, cam = _scene(**case)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    splats, depths, counts, _ = orc.project_count(g, sh, cam, st, ti)
    got = splats.view(np.float16).reshape(-1, 12)           # ndc.xy | extent.xy | conic.xy | conic.z, 0 | r, g | b, sigma(o)
    gs, cp = ind.unpack_gaussians(g), ind.camera_parts(cam)
    pr = ind.project(gs, cp)
    tb = ind.tile_boxes(pr, cp)
    col = ind.colour(gs, ind.unpack_sh(sh), cp, cfg.sh_deg)

    # visibility and tile counts: identical except where a float32 and a float64 evaluation of the same number fall on different sides
    # of an fp16 rounding boundary or of a tile edge (documented ties): a handful in 10 k at most, each off by one tile row / column
    vis_o = counts > 0
    assert (vis_o != tb["visible"]).sum() <= max(2, vis_o.size // 3000), ((vis_o != tb["visible"]).sum(), vis_o.size)
    both = vis_o & tb["visible"]
    assert both.sum() > 0.3 * vis_o.size
    differ = np.flatnonzero(both & (counts != tb["count"]))
    assert differ.size <= max(3, both.sum() // 1500), (differ.size, both.sum())
    for i in differ:  # each explained by the stored fp16 centre / extent being one ulp away from the float64 value's rounding
        ulps = max(ind.f16_ulp_distance(got[i, 0:4].view(np.uint16), ind.f16_bits(np.concatenate([tb["ndc16"][i], tb["ext16"][i]]))).max(), 0)
        assert ulps <= 1 or abs(int(counts[i]) - int(tb["count"][i])) <= max(tb["tile_max"][i] - tb["tile_min"][i] + 1), (i, ulps, counts[i], tb["count"][i])

    # stored fp16 fields vs the float64 value rounded to fp16: within one unit in the last place, and mostly identical
    want = np.concatenate([np.clip(pr["ndc"][:, :2], -60000, 60000), pr["extent_capped"], pr["conic"], np.zeros((g.shape[0], 1)),
                           np.clip(col, 0, 1), np.clip(pr["sigma"], 0, 1)[:, None]], 1)
    d = ind.f16_ulp_distance(got[both].view(np.uint16), ind.f16_bits(want[both]))
    names = ["ndc.x", "ndc.y", "extent.x", "extent.y", "conic.x", "conic.y", "conic.z", "pad", "r", "g", "b", "sigma"]
    # (the off-diagonal conic term is a difference of products: where it all but cancels, one fp16 ulp of a number 1e-5 of the diagonal
    # says nothing, and the float32 result is judged against the diagonal's scale instead)
    abs_err = np.abs(got[both].astype(np.float64) - want[both])
    slack = np.zeros_like(abs_err)
    slack[:, 5] = 1e-5 * np.sqrt(np.abs(want[both][:, 4] * want[both][:, 6]))
    for j, name in enumerate(names):
        bad = (d[:, j] > 1) & (abs_err[:, j] > slack[:, j])
        assert not bad.any(), (name, int(d[:, j].max()), int(np.argmax(bad)))
        assert (d[:, j] == 0).mean() > 0.97, (name, float((d[:, j] == 0).mean()))

    # depth key: top 16 bits of the ordered view depth (7 mantissa bits: a float32 / float64 disagreement is a 1-in-10^5 event)
    key = ind.depth_key16(pr["t"][:, 2])
    assert ((depths[both] >> 16) != key[both]).sum() <= 1


def test_k1_real_sh_basis_is_the_scipy_one():
    """The colour model's basis functions are the real spherical harmonics (Condon-Shortley phase kept), orthonormal on the sphere:
    checked by quadrature, so the SH constants the oracle multiplies by are pinned by scipy, not by a transcription."""
    rng = np.random.default_rng(1)
    d = rng.normal(size=(200000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    Y = ind.real_sh_basis(d, 3)
    gram = 4 * np.pi * (Y.T @ Y) / d.shape[0]           # Monte-Carlo <Y_j, Y_k> over the sphere
    assert np.abs(gram - np.eye(16)).max() < 0.03
    # l = 1 in the colour model's order and sign: -y, +z, -x (times sqrt(3 / 4 pi))
    c1 = np.sqrt(3 / (4 * np.pi))
    assert np.allclose(Y[:, 1:4], np.stack([-c1 * d[:, 1], c1 * d[:, 2], -c1 * d[:, 0]], 1), atol=1e-12)


# ----------------------------------------------------------------------------------------------------------------- (b) K17
def _k17_case(orc, view, n=400, seed=0):
    """Visible, well-conditioned Gaussians under a rotated camera, random cotangents for (pixel centre, conic, sigma(opacity)) given to
    K17 as exact fixed-point accumulators, and F(params) = <cotangent, float64 forward(params)> whose gradient K17 claims to be."""
    cfg = harness.small_config("c2", num_points=4000, width=640, height=480, sh_deg=0, s0=0.02)
    g, sh = synth.make_gaussians(cfg)
    cam = synth.circle_cameras(cfg, 8, radius=2.5)[view]           # a clearly rotated view (Q10 needs view3x3 != its transpose)
    gs, cp = ind.unpack_gaussians(g), ind.camera_parts(cam)
    pr = ind.project(gs, cp)
    tb = ind.tile_boxes(pr, cp)
    lim = 1.2 * np.array([0.5 * cp["width"] / cp["fx"], 0.5 * cp["height"] / cp["fy"]])
    ok = tb["visible"] & (np.abs(pr["t"][:, 0] / pr["t"][:, 2]) < lim[0]) & (np.abs(pr["t"][:, 1] / pr["t"][:, 2]) < lim[1]) & (pr["extent"].max(1) < 100.0)
    pick = np.flatnonzero(ok)[:n]
    assert pick.size >= 200
    g = np.ascontiguousarray(g[pick])
    rng = np.random.default_rng(seed)
    m = pick.size
    gm = rng.integers(-3_000_000, 3_000_000, (m, 2)).astype(np.int32)       # dF / d(pixel centre), x 1e6
    gc = np.zeros((m, 4), np.int32)
    gc[:, [0, 1, 3]] = rng.integers(-40_000_000, 40_000_000, (m, 3))         # dF / d(conic.x, conic.y, conic.z), x 1e6 (slot 2 unused)
    go = rng.integers(-2_000_000, 2_000_000, m).astype(np.int32)            # dF / d(sigma(opacity_raw)), x 1e6
    gcol = rng.integers(-1_000_000, 1_000_000, (m, 3)).astype(np.int32)
    st = synth.render_settings(cfg)
    st[5] = 0.0
    return cfg, g, cam, st, gm, gc, go, gcol


def _k17_fd(g, cam, gm, gc, go):
    """Central differences of F in float64, one parameter of every Gaussian at a time (F_i depends on Gaussian i only)."""
    cp = ind.camera_parts(cam)
    base = ind.unpack_gaussians(g)
    wm, wc, wo = gm / 1e6, gc[:, [0, 1, 3]] / 1e6, go / 1e6

    def F(gs):
        pr = ind.project(gs, cp)
        return (wm * pr["px"]).sum(1) + (wc * pr["conic"]).sum(1) + wo * pr["sigma"]

    out = {}
    for key, width in (("pos", 3), ("opacity_raw", 1), ("quat", 4), ("log_scale", 3)):
        cols = []
        for j in range(width):
            eps = 1e-5
            hi, lo = {k: v.copy() for k, v in base.items()}, {k: v.copy() for k, v in base.items()}
            if width == 1:
                hi[key] = hi[key] + eps; lo[key] = lo[key] - eps
            else:
                hi[key][:, j] += eps; lo[key][:, j] -= eps
            cols.append((F(hi) - F(lo)) / (2 * eps))
        out[key] = np.stack(cols, 1)
    return out


def _k17_oracle(orc, g, cam, st, gm, gc, go, gcol, flags):
    orc.set_k17_fix(flags)
    try:
        grads = orc.geometry_backward(cam, st, g, np.ascontiguousarray(gm.reshape(-1)), np.ascontiguousarray(gc.reshape(-1)), go, np.ascontiguousarray(gcol.reshape(-1)))
    finally:
        orc.set_k17_fix(0)
    h = grads.view(np.float16).reshape(-1, 16).astype(np.float64)
    return dict(pos=h[:, 0:3], opacity_raw=h[:, 3:4], quat=h[:, 4:8], log_scale=h[:, 8:11], colour=h[:, 12:15])


def _rel(got, ref):
    """Worst per-Gaussian error relative to that Gaussian's gradient norm (floored at 1 % of the median norm)."""
    nrm = np.linalg.norm(ref, axis=1)
    return float((np.linalg.norm(got - ref, axis=1) / np.maximum(nrm, 0.01 * np.median(nrm))).max())


@pytest.mark.parametrize("view", [1, 4, 6])
def test_k17_with_its_three_deviations_undone_is_the_true_gradient(orc, view):
    """The reference's K17, as restated by the oracle, with Q10, Q11 and Q23 switched off (oracle_backward.cpp: orc_set_k17_fix) equals
    the float64 finite-difference gradient of the float64 forward to fp16 output precision -- for position, raw opacity, quaternion and
    log-scale.  So these three are the ONLY places where the restated chain rule departs from the true gradient; everything else in
    K17 (Jacobian, clamp masks, quaternion and scale chain, sigmoid) is read and evaluated correctly."""
    cfg, g, cam, st, gm, gc, go, gcol = _k17_case(orc, view)
    fd = _k17_fd(g, cam, gm, gc, go)
    got = _k17_oracle(orc, g, cam, st, gm, gc, go, gcol, flags=7)
    for key in ("pos", "opacity_raw", "quat", "log_scale"):
        err = _rel(got[key], fd[key])
        assert err < 3e-3, (key, err)          # fp16 outputs: 2^-11 = 4.9e-4 per component, fp32 arithmetic inside
    # colour passes through K17 unchanged (tiled-backward.wgsl:293-297): the fixed-point sums, rounded to fp16
    assert np.array_equal(got["colour"], ind.f16_round(gcol / 1e6))


@pytest.mark.parametrize("flags,broken", [(0, ("pos", "quat", "log_scale")), (6, ("pos", "quat", "log_scale")), (5, ("pos",)), (3, ("pos", "quat", "log_scale"))],
                         ids=["reference", "only-Q10-left", "only-Q11-left", "only-Q23-left"])
def test_k17_each_deviation_is_real(orc, flags, broken):
    """Leaving any one of the three in place breaks the agreement (so each switch undoes a real deviation, and the reference's K17 --
    flags 0, what the product reproduces bit for bit -- is NOT the gradient of the reference's forward):
      Q10 (W = view3x3 instead of its transpose) corrupts the covariance path: position, quaternion and log-scale;
      Q11 (+0.5 instead of -0.5 viewport.y) flips the y part of the position gradient that comes through the projected centre;
      Q23 (dL/dconic.y counted twice) corrupts everything that depends on the 2D covariance."""
    cfg, g, cam, st, gm, gc, go, gcol = _k17_case(orc, 4)
    fd = _k17_fd(g, cam, gm, gc, go)
    got = _k17_oracle(orc, g, cam, st, gm, gc, go, gcol, flags=flags)
    for key in ("pos", "quat", "log_scale"):
        err = _rel(got[key], fd[key])
        if key in broken:
            assert err > 0.05, (key, err)
        else:
            assert err < 3e-3, (key, err)
    assert _rel(got["opacity_raw"], fd["opacity_raw"]) < 3e-3    # the opacity path has no deviation
