"""CPU: the "stated fp32 tolerance" of BASELINE.json's north star, measured.

WGSL leaves FMA contraction to the implementation, so the reference's raster expressions have more than one legal value.  The
oracle pins one (DESIGN.md "dmath": q = fma(fma(cx,dx,(2cy)dy),dx,(cz dy)dy), C = fma(c, alpha*vis, C)) and the HIP kernels
reproduce that one bit for bit.  This test evaluates the SAME pipeline with the source read literally -- left to right, one rounding
per operator, no FMA (tiled-rasterizer.wgsl:228-238, tiled-backward-rasterize.wgsl:108-110) -- on BASELINE configs c1 and c2 at
full size and bounds how far the two legal evaluations are apart.  The bounds asserted here are the tolerance DESIGN.md section 2 states.
"""
import numpy as np
import pytest

from webdgs_amd import synth

CASES = [("c1", {}), ("c2", {})]  # 10 k / 256x256 / SH0 and 100 k / 640x480 / SH1, both at their BASELINE sizes

# stated tolerance (max over c1 and c2, with head-room); the measured values are printed by the test and quoted in DESIGN.md
TOL = dict(rgba8_lsb=1, rgba8_differ_frac=1e-4, final_T_abs=5e-7, n_contrib_mismatch_frac=1e-5, acc_abs_fixed=1000, acc_rel=2e-3,
           grad_f16_p999_ulp=1, grad_f16_mismatch_frac=0.005)


def _f16_ulp_distance(a_bits, b_bits):
    """Distance in fp16 representable values between two arrays of fp16 bit patterns (sign-magnitude -> monotone integer)."""
    def key(b):
        b = b.astype(np.int32)
        return np.where(b & 0x8000, -(b & 0x7FFF), b & 0x7FFF)
    return np.abs(key(a_bits) - key(b_bits))


@pytest.mark.parametrize("base,kw", CASES)
def test_literal_order_vs_pinned_order(orc, base, kw):
    import harness
    cfg = harness.small_config(base, **kw)
    g, sh, cam = harness.scene(cfg)
    st, ti = synth.render_settings(cfg), synth.tile_info(cfg.width, cfg.height, 0)
    tg, tsh = synth.make_target_scene(g, sh)
    target = orc.forward(tg, tsh, cam, st, ti)["rgba8"]
    try:
        orc.set_literal_order(False)
        pin = orc.view_gradients(g, sh, cam, st, ti, target)
        orc.set_literal_order(True)
        lit = orc.view_gradients(g, sh, cam, st, ti, target)
    finally:
        orc.set_literal_order(False)
    # ---- everything upstream of the raster is untouched by the switch: bit-identical, as north_star demands for tile/sort indices
    for k in ("tile_counts", "tile_offsets", "sorted_keys", "sorted_values", "tile_ranges", "splats"):
        assert np.array_equal(pin[k], lit[k]), k
    rgba = np.abs(pin["rgba8"].astype(np.int32) - lit["rgba8"].astype(np.int32))
    t_abs = np.abs(pin["final_T"].astype(np.float64) - lit["final_T"].astype(np.float64))
    nc_bad = float((pin["n_contrib"] != lit["n_contrib"]).mean())
    print(f"[{base}] rgba8: max {rgba.max()} LSB, {100.0 * (rgba > 0).mean():.4f} % of channel values differ; final_T max abs {t_abs.max():.3e}; "
          f"n_contrib differs at {100.0 * nc_bad:.5f} % of pixels")
    assert rgba.max() <= TOL["rgba8_lsb"] and (rgba > 0).mean() <= TOL["rgba8_differ_frac"]
    assert t_abs.max() <= TOL["final_T_abs"]
    assert nc_bad <= TOL["n_contrib_mismatch_frac"]
    # ---- backward: the x1e6 fixed-point accumulators and the packed fp16 gradients
    worst_abs, worst_rel = 0, 0.0
    for k in ("grad_means", "grad_conics", "grad_opacity", "grad_colors"):
        a, b = pin[k].astype(np.int64), lit[k].astype(np.int64)
        d = np.abs(a - b)
        scale = np.maximum(np.abs(a), np.abs(b))
        rel = float((d / np.maximum(scale, 1)).max(where=scale > 1000, initial=0.0))
        worst_abs, worst_rel = max(worst_abs, int(d.max())), max(worst_rel, rel)
    # (a pixel whose alpha sits on the 1/255 threshold may contribute in one order and not in the other, Q7: that is the tail)
    assert worst_abs <= TOL["acc_abs_fixed"] and worst_rel <= TOL["acc_rel"]
    pg = np.ascontiguousarray(pin["gradients"]).view(np.uint16)
    lg = np.ascontiguousarray(lit["gradients"]).view(np.uint16)
    fin = np.isfinite(pg.view(np.float16).astype(np.float32)) & np.isfinite(lg.view(np.float16).astype(np.float32))
    ulp = _f16_ulp_distance(pg, lg)[fin]
    frac = float((ulp > 0).mean())
    p999 = int(np.percentile(ulp, 99.9)) if ulp.size else 0
    print(f"[{base}] accumulators: max |diff| {worst_abs} fixed-point units (1e-6), max rel {worst_rel:.2e} (|v| > 1e-3); "
          f"fp16 gradients: {100.0 * frac:.3f} % differ, 99.9th percentile {p999} ulp, max {int(ulp.max()) if ulp.size else 0} ulp")
    # (the maximum is not bounded: K17 differences of nearly equal accumulator terms amplify one fixed-point unit into many fp16 ulps)
    assert p999 <= TOL["grad_f16_p999_ulp"]
    assert frac <= TOL["grad_f16_mismatch_frac"]
