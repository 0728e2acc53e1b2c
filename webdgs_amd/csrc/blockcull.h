// Conservative block-level alpha test shared by the kernels that walk a tile list per 8x8 pixel block (backward_raster.hip, densify.hip).
#pragma once
#include "dmath.h"

// Can any pixel centre of the block [x0, x1] x [y0, y1] (offsets from the splat centre) reach alpha >= 1/255?  alpha = min(0.99, o G)
// with G = exp(-q/2), q = A dx^2 + 2 B dx dy + C dy^2: the reference skips a pixel-splat pair below that alpha before it touches any
// state (tiled-backward-rasterize.wgsl:116-118), so a splat whose LARGEST alpha over the block is below it changes nothing for
// this wave.  For a positive definite conic the minimum of q over the rectangle is 0 if the rectangle holds the centre, else the
// smallest of the four edge minima (a convex function takes its minimum over a convex set that excludes the unconstrained minimiser on
// the boundary).  The test is CONSERVATIVE, never exact: it uses approximate reciprocals / logarithm and is padded by a margin that
// covers their error and the rounding of q (terms of size M, a few ulp each) a hundred times over; anything it is unsure about --
// an indefinite conic after fp16 rounding, a NaN -- is kept, and the per-pixel test below still decides.  Outputs do not depend
// on it; only the number of (wave, splat) iterations does (about a third of the box-overlapping ones were empty at BASELINE c3).
WD_DEV bool block_reaches_min_alpha(float A, float B, float C, float opacity, float x0, float x1, float y0, float y1) {
    if ((x0 <= 0.0f) & (x1 >= 0.0f) & (y0 <= 0.0f) & (y1 >= 0.0f)) return true;  // the centre is inside: q = 0 there
    const float det = A * C - B * B;
    if (!((A > 0.0f) & (C > 0.0f) & (det > 0.0f))) return true;  // not provably convex (also NaN): keep
    const float rA = __builtin_amdgcn_rcpf(A), rC = __builtin_amdgcn_rcpf(C);
    auto q_at = [&](float dx, float dy) { return (A * dx) * dx + ((B + B) * dx) * dy + (C * dy) * dy; };
    auto clampf = [](float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); };
    const float q1 = q_at(x0, clampf(-(B * x0) * rC, y0, y1));
    const float q2 = q_at(x1, clampf(-(B * x1) * rC, y0, y1));
    const float q3 = q_at(clampf(-(B * y0) * rA, x0, x1), y0);
    const float q4 = q_at(clampf(-(B * y1) * rA, x0, x1), y1);
    const float qmin = fminf(fminf(q1, q2), fminf(q3, q4));
    const float mx = fmaxf(fabsf(x0), fabsf(x1)), my = fmaxf(fabsf(y0), fabsf(y1));
    const float M = (A * mx) * mx + (2.0f * fabsf(B) * mx) * my + (C * my) * my;
    if (!(M <= 3.0e38f)) return true;   // an infinite or NaN conic: q may be a NaN at single pixels (inf * 0), whose alpha the per-pixel test then takes as 0.99
    const float thr = 2.0f * __logf(255.0f * opacity);  // alpha >= 1/255  <=>  q <= 2 ln(255 o)
    return !(qmin > thr + (1e-3f + 1e-5f * M));         // NaN anywhere -> true
}
