// Backward rasterization (K16): dL/d{mean2D, conic, opacity, colour} per splat; one WAVE (64-thread workgroup) per 8x8
// pixel block, four per 16x16 tile.
//
// Replaces backward_rasterize_main (src/shaders/tiled-backward-rasterize.wgsl:34-172): per pixel, back to front over its
// first n_contrib tile entries, with every lane re-loading the splat from global memory and issuing 9 global fixed-point
// atomics per contributing (pixel, splat) pair -- the reference's dominant cost.  Here:
//   * a wave walks the entries [0, max n_contrib of its 64 pixels) back to front in chunks of 64 (lane = entry); the
//     Gaussian index two chunks ahead and the Splat one chunk ahead are fetched into registers (latency overlaps compute);
//   * the chunk is compacted (ballot + prefix popcount, order preserving) to the splats whose extent box overlaps the block,
//     into 3 KB of wave-private LDS; no workgroup barrier exists, so a block with few contributors never waits for a
//     neighbour with many;
//   * eight of the 9 per-pixel contributions are summed across the wave through a transposition in wave-private LDS (every lane stores a
//     column of an [8][64] array, lane (q, part) adds eight consecutive entries of row q, three DPP steps finish the sum) and blue in
//     registers; the sums land in twelve lanes (eight slots, and blue as one partial sum per 16-lane row) that issue ONE atomic
//     instruction per (wave, splat) on twelve consecutive words -- instead of 9 per (pixel, splat).  (Round 2 summed in registers with a
//     halving butterfly -- v_permlane32_swap, v_permlane16_swap, DPP -- which costs few instructions but much issue time on gfx950:
//     DESIGN.md section 4, "What a VALU instruction costs".  That form is kept behind WDGS_BWR_SUMS=butterfly for same-box comparisons.)
//   * one wave per workgroup; a tile's four blocks are numbered so that they are dispatched back to back on one XCD and share its L2
//     lines of the entry list (WDGS_BWR_WPW=4: workgroup = tile).
// The contributions keep the reference's semantics exactly: each is truncated to i32 at x1e6 per pixel
// (common.wgsl:113-116) and integer addition is order-free, so the result is bit-reproducible and equal to the oracle's.
// Bound: fp32 VALU issue -- about 113 wave-instructions per (wave, splat) with a contributing pixel: the pinned exp (12), one division (8),
// nine fixed-point conversions (18), the per-pixel gradient arithmetic (about 45), the reduction (17) and the tests and bookkeeping
// around them, at 0.96-0.99 of the rate this chip sustains for a pure FMA stream; DESIGN.md section 4 has the counters.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "dmath.h"
#include "blockcull.h"
#include "longlist.h"

namespace {

constexpr u32 ACC_STRIDE = 12;  // i32 per Gaussian: mean.xy, conic.xyz, opacity, r, g, and blue as four partial sums (words 8..11)

WD_DEV int dpp_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }       // quad_perm [1,0,3,2]
WD_DEV int dpp_xor2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true); }       // quad_perm [2,3,0,1]
WD_DEV int dpp_half_mirror(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true); }
WD_DEV int dpp_mirror(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true); }

WD_DEV int dpp_ror4(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, true); }  // row_ror:4: a rotation by one quad within the row
WD_DEV int dpp_ror8(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true); }  // row_ror:8
// every lane of each quad ends up with the quad's sum
WD_DEV int quad_sum(int v0) {
    unsigned v = (unsigned)v0;  // wrapping two's-complement sums, like atomicAdd on atomic<i32>
    v += (unsigned)dpp_xor1((int)v);
    v += (unsigned)dpp_xor2((int)v);
    return (int)v;
}
// every lane of each 16-lane row ends up with the row's sum
WD_DEV int row_sum(int v0) {
    unsigned v = (unsigned)v0;  // wrapping two's-complement sums, like atomicAdd on atomic<i32>
    v += (unsigned)dpp_xor1((int)v);
    v += (unsigned)dpp_xor2((int)v);
    v += (unsigned)dpp_half_mirror((int)v);
    v += (unsigned)dpp_mirror((int)v);
    return (int)v;
}
// lanes 0..31: a[l] + a[l+32];  lanes 32..63: b[l-32] + b[l]   (v_permlane32_swap: measured semantics on gfx950)
WD_DEV int fold32(int a, int b) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    return (int)(r[0] + r[1]);
}
// rows (16 lanes) 0..3: a.r0+a.r1 | b.r0+b.r1 | a.r2+a.r3 | b.r2+b.r3   (v_permlane16_swap)
WD_DEV int fold16(int a, int b) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
    return (int)(r[0] + r[1]);
}

typedef wd_pair f2;  // component-wise scalar arithmetic (dmath.h)

constexpr float FIXED_SCALE = 1000000.0f;  // common.wgsl:113-116
WD_DEV int cvt_fixed(float scaled) {
    // i32(v * 1e6) after the multiply: truncate toward zero, saturate, NaN -> 0 -- exactly v_cvt_i32_f32.
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(scaled));
    return r;
}

// WPW = waves per workgroup: 1 (default: workgroup = one 8x8 block; the four blocks of a tile are numbered so that they are dispatched back
// to back on one XCD and share its L2 lines of the entry list) or 4 (workgroup = tile, round 2's form).
// LDS_SUMS: the wave sums of eight of the nine contributions go through a transposition in wave-private LDS (below) instead of the
// register butterfly.
// TIMELINE (measurement tool, WDGS_BWR_TIMELINE=<file>; one-wave workgroups only): every wave leaves {start, end} of the 100 MHz wall clock, where it
// ran (XCC, SE, CU, SIMD from the hardware id registers) and how many splats it iterated over -- scripts/bwr_timeline.py reads the file.
// PRIO: a launch whose waves are all resident from the start (c2: 4 800 waves, 8 192 slots) takes as long as its LONGEST wave -- the waves of a
// SIMD start together and leave one by one, profiles/r05r_bwr_timeline_c2.txt -- and a wave iterates faster the fewer waves compete for its
// SIMD's issue slots (446 ns per iteration alone, 617 ns among four: profiles/r05v_bwr_wave_rate.txt).  The wave therefore sets its issue
// priority from the entries it still has to walk, once per chunk: longest remaining chain first.  c2 187.0 -> 176.9 us per step, c3 unchanged
// (profiles/r05x_bwr_issue_priority_sweep.txt; thresholds swept there).  Arbitration only: results cannot depend on it.
template <u32 WPW, bool LDS_SUMS, bool TIMELINE, bool PRIO>
__device__ __attribute__((always_inline)) void backward_rasterize_body(const RenderSettings& settings, u32 num_tiles_x, const u32* __restrict__ ranges,
                                                                 const u32* __restrict__ instances, const u32* __restrict__ splats,
                                                                 const float* __restrict__ final_T, const u32* __restrict__ n_contrib,
                                                                 const float4* __restrict__ loss_grad, int* __restrict__ acc,
                                                                 unsigned long long* __restrict__ timeline, u32 tile_id, u32 sub, u32 range_start /* ranges[tile_id] */,
                                                                 u32 range_end /* ranges[tile_id + 1] */, u32 n_val /* n_contrib of the lane's pixel, 0 outside the image */,
                                                                 float4* s_geo, float4* s_con, float4* s_col, int* s_sum) {
    const unsigned long long t_start = TIMELINE ? wall_clock64() : 0ull;
    u32 iterations = 0u;
    auto leave_timeline = [&]() {
        if (TIMELINE && (threadIdx.x & 63u) == 0u) {
            u32 hw_id, xcc_id;
            asm volatile("s_getreg_b32 %0, hwreg(4, 0, 32)" : "=s"(hw_id));    // HW_REG_HW_ID: wave, SIMD, CU, SH, SE
            asm volatile("s_getreg_b32 %0, hwreg(20, 0, 32)" : "=s"(xcc_id));  // HW_REG_XCC_ID
            unsigned long long* const rec = timeline + (size_t)blockIdx.x * 4u;
            rec[0] = t_start; rec[1] = wall_clock64(); rec[2] = ((unsigned long long)xcc_id << 32) | hw_id; rec[3] = iterations;
        }
    };
    const u32 tile_x = tile_id % num_tiles_x, tile_y = tile_id / num_tiles_x;
    const u32 lane = threadIdx.x & 63u;
    const u32 bx = tile_x * 16u + (sub & 1u) * 8u, by = tile_y * 16u + (sub >> 1) * 8u;
    const u32 pixel_x = bx + (lane & 7u), pixel_y = by + (lane >> 3);
    const float vx = settings.viewport_x, vy = settings.viewport_y;
    const u32 W = wd_to_u32(vx);
    const size_t p = (size_t)pixel_y * W + pixel_x;   // (inside the image wherever n_val != 0)
    const float cap = (settings.max_splat_radius_px > 0.0f) ? settings.max_splat_radius_px : 1e9f;
    const float blk_x0 = (float)bx + 0.5f, blk_x1 = (float)bx + 7.5f, blk_y0 = (float)by + 0.5f, blk_y1 = (float)by + 7.5f;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const u32 quad_lane = lane & 3u;
    // LDS_SUMS: lane (q, part) = (lane >> 3, lane & 7) adds up the eight contributions [q][8 part .. 8 part + 7] -- two 16-byte reads, the
    // halves taken in opposite order by odd q, which makes both reads conflict-free in ds_read_b128's lane groups
    const u32 sum_q = lane >> 3, sum_first = (sum_q & 1u) * 4u;
    const int4* const sum_rd0 = reinterpret_cast<const int4*>(s_sum + (LDS_SUMS ? sum_q * 64u + (lane & 7u) * 8u + sum_first : 0u));
    const int4* const sum_rd1 = reinterpret_cast<const int4*>(s_sum + (LDS_SUMS ? sum_q * 64u + (lane & 7u) * 8u + (4u - sum_first) : 0u));
    const bool sum_lane = LDS_SUMS ? ((lane & 7u) == 0u) : false;
    const bool atomic_lane = LDS_SUMS ? ((lane & 7u) == 0u || (lane & 15u) == 1u) : ((lane & 15u) < 3u);
    const u32 atomic_slot = LDS_SUMS ? (sum_lane ? sum_q : 8u + (lane >> 4)) : ((quad_lane == 2u) ? 8u + (lane >> 4) : 2u * (lane >> 4) + quad_lane);

    const u32 tile_entries = (range_end > range_start) ? range_end - range_start : 0u;
    float T = 0.0f;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (min(n_val, tile_entries) > 0u) { T = final_T[p]; g = loss_grad[p]; }
    // A pixel whose final T is a NaN -- the forward pass met a Splat with a NaN alpha inside its extent box: a non-finite Gaussian -- contributes
    // nothing: T stays a NaN through every division, each of the nine contributions has T or dL/dalpha = (..) * T as a factor, and the fixed-point
    // conversion of a NaN is 0 (common.wgsl:113-116 as the parity oracle pins it; v_cvt_i32_f32).  Such a pixel leaves the walk here.  Late in a run of the
    // reference's schedule tile 0 holds thousands of such Gaussians and every one of its pixels is of this kind.
    const u32 pix_n = (T == T) ? min(n_val, tile_entries) : 0u;

    u32 wmax = pix_n;  // wave maximum (uniform)
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) wmax = max(wmax, (u32)__shfl_xor((int)wmax, (int)d, 64));
    if (wmax == 0u) { leave_timeline(); return; }
    const f2 pxy = f2{(float)pixel_x + 0.5f, (float)pixel_y + 0.5f};
    const f2 g_rg = f2{g.x, g.y};
    const float g_b = g.z;
    f2 ar_rg = f2{0.f, 0.f};  // accum_rec: the colour accumulated behind the current splat
    float ar_b = 0.f;

    // chunk [lo, hi) of the tile list, lane j <-> entry lo + j; software pipeline: index two chunks ahead, Splat one ahead
    auto chunk_lo = [&](u32 hi_) { return (hi_ > 64u) ? hi_ - 64u : 0u; };
    auto fetch_idx = [&](u32 hi_) -> u32 {  // hi_ == 0: no such chunk
        const u32 lo_ = chunk_lo(hi_);
        return (hi_ > 0u && lane < hi_ - lo_) ? instances[range_start + lo_ + lane] : 0xFFFFFFFFu;
    };
    u32 gidx_c = fetch_idx(wmax);
    u32 gidx_n = fetch_idx(chunk_lo(wmax));
    uint2 w01 = make_uint2(0u, 0u), w23 = w01, w45 = w01;
    if (gidx_c != 0xFFFFFFFFu) {
        const uint2* sp = reinterpret_cast<const uint2*>(splats + (size_t)gidx_c * 6);
        w01 = sp[0]; w23 = sp[1]; w45 = sp[2];
    }
    for (u32 hi = wmax; hi > 0u;) {
        const u32 lo = chunk_lo(hi);
        if (PRIO) {
            // (hi is the same in every lane, which the compiler cannot see: as a scalar, or the four s_setprio end up in exec-masked regions
            // that are not branched around and the last one always wins)
            const u32 left = (u32)__builtin_amdgcn_readfirstlane((int)hi);
            if (left >= 128u) __builtin_amdgcn_s_setprio(3);
            else if (left >= 64u) __builtin_amdgcn_s_setprio(2);
            else if (left >= 32u) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        // ---- this lane's entry: overlap test against the wave's block, order-preserving compaction into LDS
        const bool have = gidx_c != 0xFFFFFFFFu;
        const float cx = (wd_unpack_lo(w01.x) * 0.5f + 0.5f) * vx;
        const float cy = (wd_unpack_hi(w01.x) * -0.5f + 0.5f) * vy;
        // (WGSL's min, as the parity oracle evaluates it: a NaN extent stays a NaN and then passes every "outside" test; fminf would make it the cap)
        const float ex = wd_min(wd_unpack_lo(w01.y), cap), ey = wd_min(wd_unpack_hi(w01.y), cap);
        const bool in_box = have && !((blk_x0 - cx) > ex || (cx - blk_x1) > ex || (blk_y0 - cy) > ey || (cy - blk_y1) > ey);
        const bool ok = in_box && block_reaches_min_alpha(wd_unpack_lo(w23.x), wd_unpack_hi(w23.x), wd_unpack_lo(w23.y), wd_unpack_hi(w45.y), blk_x0 - cx,
                                                          blk_x1 - cx, blk_y0 - cy, blk_y1 - cy);
        const unsigned long long m = __ballot(ok);
        const u32 n_list = (u32)__popcll(m);
        if (ok) {
            const u32 slot = (u32)__popcll(m & lt_mask);
            s_geo[slot] = make_float4(cx, cy, ex, ey);
            s_con[slot] = make_float4(-0.5f * wd_unpack_lo(w23.x), -0.5f * wd_unpack_hi(w23.x), -0.5f * wd_unpack_lo(w23.y), wd_unpack_hi(w45.y));  // s = -conic / 2
            s_col[slot] = make_float4(wd_unpack_lo(w45.x), wd_unpack_hi(w45.x), wd_unpack_lo(w45.y), __uint_as_float(gidx_c));
        }
        // A pixel composited the entries at positions below its n_contrib: of this chunk, lanes j < pix_n - lo, i.e. -- the compaction keeps
        // the order -- the first `mine` records of the list.  (Counting them here, once per chunk, replaces a fourth record array with the
        // entry's position and its read in every iteration: 3 KB of records + 2 KB of sums = 5 KB per wave, 32 resident waves per CU.)
        const u32 rel = (pix_n > lo) ? min(pix_n - lo, 64u) : 0u;
        const u32 mine = (u32)__popcll(m & ((rel >= 64u) ? ~0ull : ((1ull << rel) - 1ull)));
        // next chunk's Splat gather and the index fetch of the chunk after it: in flight while this chunk is processed
        gidx_c = gidx_n;
        if (gidx_c != 0xFFFFFFFFu) {
            const uint2* sp = reinterpret_cast<const uint2*>(splats + (size_t)gidx_c * 6);
            w01 = sp[0]; w23 = sp[1]; w45 = sp[2];
        }
        gidx_n = fetch_idx(chunk_lo(lo));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS records written above are read below by other lanes
        __builtin_amdgcn_wave_barrier();

        for (u32 i = n_list; i-- > 0u;) {  // back to front
            // all three records at once: one LDS round trip per iteration (after the block-level alpha test nearly every iteration
            // has a contributing pixel, so the early exits the staged reads used to serve are rare)
            const float4 geo = s_geo[i];
            const float4 con = s_con[i];
            const float4 col = s_col[i];
            const f2 d = pxy - f2{geo.x, geo.y};
            // (bitwise, not short-circuit: no branches for the three tests)
            const bool cand = ((int)(i < mine) & (int)!(fabsf(d.x) > geo.z) & (int)!(fabsf(d.y) > geo.w)) != 0;
            // The reference's factors -0.5 (exponent) and 2 (off-diagonal conic term, dpow: tiled-backward-rasterize.wgsl:96-99, 141-142)
            // are powers of two: they commute with every rounding of the products and sums they pass through.  The record therefore holds
            // s = -0.5 * conic, which gives the exponent's argument directly, and the derivative terms further down carry the factor
            // that is left into the fixed-point scale.  Same bits, no multiplication by 2 or -0.5 per (pixel, splat); no operand here is
            // small enough for a product to be subnormal (conics are fp16 values).
            const float syd = con.y * d.y;
            const float t1 = __builtin_fmaf(con.x, d.x, syd + syd);
            // exp with its range handling hoisted out of the common case (dmath.h wd_exp_inrange): an argument above 87 or a NaN -- an
            // indefinite conic after fp16 rounding -- sends the whole wave through the full form; arguments below -80 are clamped, which
            // only changes G on lanes whose alpha is far below 1/255 either way, and G is not used on those.
            const float xe = __builtin_fmaf(t1, d.x, (con.z * d.y) * d.y);  // -0.5 * power
            float xc;  // max(xe, -80): one v_max_f32 (fmaxf would first quiet a NaN, which the test below sends to the full form)
            asm("v_max_f32 %0, 0xc2a00000, %1" : "=v"(xc) : "v"(xe));  // 0xc2a00000 = -80.0f
            float G = wd_exp_inrange(xc);
            if (__builtin_expect(__any(!(xe <= 87.0f)), 0)) G = wd_exp(xe);
            const float og = con.w * G;
            const float alpha = (og < 0.99f) ? og : 0.99f;  // WGSL min(0.99, opacity*G)
            const bool act = cand && !(alpha < (1.0f / 255.0f));
            if (!__any(act)) continue;
            if (TIMELINE) iterations++;
            // No branch around the per-pixel arithmetic: a pixel that does not contribute runs it with alpha = 0 and dL/dalpha = 0,
            // which leaves its state exactly as it was (T / 1 = T, accum_rec = 0 * colour + 1 * accum_rec) and makes every one of its
            // nine fixed-point contributions 0 (products with a zero factor; an inf * 0 = NaN converts to 0 as well) -- two selects
            // instead of nine zero-initialisations and an exec-mask round trip.  Contributing pixels see the reference's own products
            // and sums, in its order (tiled-backward-rasterize.wgsl:108-160); pairs (f2) are component-wise scalar arithmetic.
            const float alpha_m = act ? alpha : 0.0f;
            // T in [1e-4, 1] (the forward pass stops before T falls below 1e-4), 1 - alpha in [0.01, 1]: the ordinary-operand division
            const float oma = 1.0f - alpha_m;
            T = wd_div_inrange(T, oma);
            // (multiply-adds below are FMAs where the parity oracle pins them: accum_rec, dL_dalpha, dpow)
            const float aT = alpha_m * T;
            const f2 frg = (aT * g_rg) * FIXED_SCALE;
            const int f_r = cvt_fixed(frg.x);
            const int f_g = cvt_fixed(frg.y);
            const int f_b = cvt_fixed((aT * g_b) * FIXED_SCALE);
            const f2 col_rg = f2{col.x, col.y};
            const f2 dc_rg = col_rg - ar_rg;
            // (the reference starts this sum from 0.0; that only decides the sign of an all-zero sum, which the fixed-point
            // conversion of every product it feeds maps to 0 either way)
            const float dL_dalpha_all = __builtin_fmaf(col.z - ar_b, g_b, __builtin_fmaf(dc_rg.y, g_rg.y, dc_rg.x * g_rg.x)) * T;
            const float dL_dalpha = act ? dL_dalpha_all : 0.0f;
            // accum_rec = last_alpha * last_color + (1 - last_alpha) * accum_rec (tiled-backward-rasterize.wgsl:120-123) is evaluated by
            // the reference at the START of the next contributing iteration from the values kept in last_alpha / last_color; the same
            // operation on the same operands is done here, while alpha and the colour are still in registers (the first iteration's
            // update, from last_alpha = 0 and accum_rec = 0, is +0 either way).
            ar_rg = f2{__builtin_fmaf(alpha_m, col_rg.x, oma * ar_rg.x), __builtin_fmaf(alpha_m, col_rg.y, oma * ar_rg.y)};
            ar_b = __builtin_fmaf(alpha_m, col.z, oma * ar_b);
            const float dL_dG = con.w * dL_dalpha;
            const int f_op = cvt_fixed((G * dL_dalpha) * FIXED_SCALE);
            // with q = (fma(s.x, dx, s.y dy), fma(s.z, dy, s.y dx)) = -dpow / 4:  -dG = -(mhG * dpow) = -((-0.5 G) * (-4 q)) = -2 (G q),
            // and the -2 rides on the scale (-2e6 is exact)
            const f2 qpow = f2{__builtin_fmaf(con.x, d.x, syd), __builtin_fmaf(con.z, d.y, con.y * d.x)};
            const float mhG = -0.5f * G;
            const f2 fm = (dL_dG * (G * qpow)) * (-2.0f * FIXED_SCALE);
            const int f_mx = cvt_fixed(fm.x);
            const int f_my = cvt_fixed(fm.y);
            const f2 fc = (dL_dG * ((mhG * d) * d)) * FIXED_SCALE;  // conic.x and conic.z terms
            const int f_cx = cvt_fixed(fc.x);
            const int f_cz = cvt_fixed(fc.y);
            // conic.y term: the reference's ((mhG * 2) * dx) * dy, scaled by dL_dG and 1e6.  A factor 2 commutes with every rounding, so it
            // is taken out of the chain -- which then starts from the mhG * dx of the conic.x term -- and folded into the scale (2e6 is exact).
            const int f_cy = cvt_fixed((dL_dG * ((mhG * d.x) * d.y)) * (2.0f * FIXED_SCALE));
            int m;
            if (LDS_SUMS) {
                // ---- nine wave sums.  A cross-lane VALU operation is expensive on gfx950 when it is counted in issue time rather than in
                // instructions (scripts/microbench/valu_issue.hip, profiles/r03s_valu_issue_w7.txt: v_permlane*_swap 10-13 cycles and a DPP
                // add 5-8 among ordinary arithmetic, which costs 2.4), and the halving butterfly needs six swaps and nine DPP steps.  Eight
                // of the sums therefore go through LDS, whose pipe this kernel leaves idle: every pixel lane stores its eight contributions
                // as a column of an [8][64] array, lane (q, part) reads back eight consecutive entries of row q and adds them with plain
                // integer adds, and three DPP steps inside the eight lanes of a row's group finish the sum.  LDS operations of one wave
                // execute in order, so the stores, the transposed reads and the next iteration's stores need no barrier between them.
                s_sum[0u * 64u + lane] = f_mx; s_sum[1u * 64u + lane] = f_my; s_sum[2u * 64u + lane] = f_cx; s_sum[3u * 64u + lane] = f_cy;
                s_sum[4u * 64u + lane] = f_cz; s_sum[5u * 64u + lane] = f_op; s_sum[6u * 64u + lane] = f_r; s_sum[7u * 64u + lane] = f_g;
                const int4 h0 = *sum_rd0, h1 = *sum_rd1;
                unsigned x = ((unsigned)h0.x + (unsigned)h0.y) + ((unsigned)h0.z + (unsigned)h0.w) + (((unsigned)h1.x + (unsigned)h1.y) + ((unsigned)h1.z + (unsigned)h1.w));
                x += (unsigned)dpp_xor1((int)x);
                x += (unsigned)dpp_xor2((int)x);
                x += (unsigned)dpp_half_mirror((int)x);   // the other quad of the eight-lane group
                // blue stays in registers: quad sums, then the four quads of a 16-lane row; row r adds its partial sum to slot 8 + r
                unsigned b = (unsigned)quad_sum(f_b);
                b += (unsigned)dpp_ror4((int)b);
                b += (unsigned)dpp_ror8((int)b);
                m = sum_lane ? (int)x : (int)b;
            } else {
            // ---- nine wave sums by a halving butterfly.  Accumulator slots: 0 mx 1 my 2 cx 3 cy 4 cz 5 op 6 r 7 g 8 b.
            // fold32 pairs slot j with slot j+4: lanes < 32 then carry slot j, lanes >= 32 slot j+4.
            const int w0 = fold32(f_mx, f_cz), w1 = fold32(f_my, f_op), w2 = fold32(f_cx, f_r), w3 = fold32(f_cy, f_g);
            // fold16: rows 0..3 of u0 carry slots 0,2,4,6; rows of u1 carry slots 1,3,5,7 (per column)
            const int u0 = quad_sum(fold16(w0, w2)), u1 = quad_sum(fold16(w1, w3)), q8 = quad_sum(f_b);
            // Every lane of a quad now holds its quad's sum of each of the three registers.  Lane j of the quad keeps one of them
            // (j = 0: u0, 1: u1, 2: blue), so the remaining row reduction -- quads 0..3 of a 16-lane row -- is done ONCE on the merged
            // register (two rotations by whole quads) instead of once per register.  Blue is not folded across the four rows: row r
            // adds its partial sum to slot 8 + r (the accumulator row has twelve words; geometry_backward adds the four up), so the
            // twelve atomic lanes of a splat hit twelve different consecutive words.
            m = (quad_lane == 0u) ? u0 : (quad_lane == 1u ? u1 : q8);
            m = (int)((unsigned)m + (unsigned)dpp_ror4(m));
            m = (int)((unsigned)m + (unsigned)dpp_ror8(m));
            }
            if (atomic_lane) {
                const u32 slot = atomic_slot;
                const u32 gidx = __float_as_uint(col.w);
                // (no test for m == 0: the atomics cost this kernel nothing -- built without them it took the same time, round 3 -- while the
                // test is a VALU compare in every iteration)
                atomicAdd(&acc[(size_t)gidx * ACC_STRIDE + slot], m);
            }
        }
        __builtin_amdgcn_wave_barrier();  // all lanes are done reading the records before the next chunk overwrites them
        hi = lo;
    }
    leave_timeline();
}

// ---- long tile lists (longlist.h): the backward walk of a block whose pixels the forward pass composited through per-pixel lists.  Lane = pixel: it
// walks ITS list back to front from its last contributor, fetches each record it meets (the forward tasks left the chunk's unpacked entries), evaluates
// what backward_rasterize_body evaluates for a (pixel, splat) pair -- the same operations on the same operands, in the parity oracle's forms (full exp,
// IEEE division) -- and adds its nine fixed-point contributions itself: integer sums do not care who adds them.  As many trips as the longest
// per-pixel list has elements in front of its last contributor -- not as the tile list has entries.
__device__ __attribute__((always_inline)) void long_backward_help(const RenderSettings& settings, u32 num_tiles_x, const float* __restrict__ final_T,
                                                                  const u32* __restrict__ n_contrib, const float4* __restrict__ loss_grad, int* __restrict__ acc, const LongWork lw) {
    // (the lane number derived again, not taken from the walk above: a vector register kept alive across that walk for this function's sake costs it --
    // raster.hip's kernel says how much)
    const u32 lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const u32 n_blocks = lw.hdr[LL_BLOCKS];   // (ll_frame_on: all of them exist)
    const u32 W = wd_to_u32(settings.viewport_x), H = wd_to_u32(settings.viewport_y);
    __builtin_amdgcn_s_setprio(3);
    // block records are dealt to the launch's waves in turn (wave w takes records w, w + waves, ...: the first waves of the grid start at once; no
    // queue, so the pass may be encoded any number of times against one forward pass)
    for (u32 lb = blockIdx.x; lb < n_blocks; lb += gridDim.x) {
        const LongBlock blk = lw.blocks[lb];
        const LongSync sy = lw.sync[lb];
        if (blk.chunks == 0u || sy.walked == 0u) continue;   // (no room, or walked the plain way: the block's main wave does the backward walk too)
        const u32 tile_x = blk.tile % num_tiles_x, tile_y = blk.tile / num_tiles_x;
        const u32 pixel_x = tile_x * 16u + (blk.sub & 1u) * 8u + (lane & 7u), pixel_y = tile_y * 16u + (blk.sub >> 1) * 8u + (lane >> 3);
        const bool in_bounds = pixel_x < W && pixel_y < H;
        const size_t p = (size_t)pixel_y * W + pixel_x;
        u32 left = lw.jlast[(size_t)lb * 64u + lane];   // elements [0, left) of the list lie at positions below the pixel's n_contrib
        float T = 0.0f;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in_bounds && left > 0u) { T = final_T[p]; g = loss_grad[p]; }
        if (!(T == T)) left = 0u;   // (a NaN final T: nothing to add, backward_rasterize_body)
        const f2 pxy = f2{(float)pixel_x + 0.5f, (float)pixel_y + 0.5f};
        const f2 g_rg = f2{g.x, g.y};
        const float g_b = g.z;
        f2 ar_rg = f2{0.f, 0.f};
        float ar_b = 0.f;
        const u32* const rows = lw.rows + ((size_t)sy.row_base * 64u + lane) * 4u + 3u;   // (the element's fourth word: the position)
        const float4* const recs = lw.records + (size_t)blk.first_item * 192u;
        u32 trips = left;
#pragma unroll
        for (u32 d = 32; d >= 1; d >>= 1) trips = max(trips, (u32)__shfl_xor((int)trips, (int)d, 64));
        // the position of trip tr + 2 and the record of trip tr + 1 are requested while trip tr computes (two dependent fetches per element otherwise)
        auto position = [&](u32 tr) -> u32 { return (tr < left) ? rows[(size_t)(left - 1u - tr) * 256u] - 1u : 0u; };
        u32 pos_next = position(0u);
        float4 geo_n = make_float4(0.f, 0.f, 0.f, 0.f), con_n = geo_n, col_n = geo_n;
        if (0u < left) { const float4* const r = recs + (size_t)pos_next * 3u; geo_n = r[0]; con_n = r[1]; col_n = r[2]; }
        pos_next = position(1u);
        for (u32 tr = 0; tr < trips; tr++) {
            const float4 geo = geo_n, con = con_n, col = col_n;
            if (tr + 1u < left) { const float4* const r = recs + (size_t)pos_next * 3u; geo_n = r[0]; con_n = r[1]; col_n = r[2]; }   // (chunk pos >> 6, entry pos & 63 of the block's items: 64 x 3 per item)
            pos_next = position(tr + 2u);
            if (tr >= left) continue;   // (this lane's list is done; others go on)
            const f2 d = pxy - f2{geo.x, geo.y};
            const float syd = con.y * d.y;
            const float t1 = __builtin_fmaf(con.x, d.x, syd + syd);
            const float xe = __builtin_fmaf(t1, d.x, (con.z * d.y) * d.y);
            const float G = wd_exp(xe);
            const float og = con.w * G;
            const float alpha = (og < 0.99f) ? og : 0.99f;
            if (alpha < (1.0f / 255.0f)) continue;   // (tiled-backward-rasterize.wgsl:116-118)
            const float oma = 1.0f - alpha;
            T = wd_div(T, oma);
            const float aT = alpha * T;
            const f2 frg = (aT * g_rg) * FIXED_SCALE;
            const int f_r = cvt_fixed(frg.x);
            const int f_g = cvt_fixed(frg.y);
            const int f_b = cvt_fixed((aT * g_b) * FIXED_SCALE);
            const f2 col_rg = f2{col.x, col.y};
            const f2 dc_rg = col_rg - ar_rg;
            const float dL_dalpha = __builtin_fmaf(col.z - ar_b, g_b, __builtin_fmaf(dc_rg.y, g_rg.y, dc_rg.x * g_rg.x)) * T;
            ar_rg = f2{__builtin_fmaf(alpha, col_rg.x, oma * ar_rg.x), __builtin_fmaf(alpha, col_rg.y, oma * ar_rg.y)};
            ar_b = __builtin_fmaf(alpha, col.z, oma * ar_b);
            const float dL_dG = con.w * dL_dalpha;
            const int f_op = cvt_fixed((G * dL_dalpha) * FIXED_SCALE);
            const f2 qpow = f2{__builtin_fmaf(con.x, d.x, syd), __builtin_fmaf(con.z, d.y, con.y * d.x)};
            const float mhG = -0.5f * G;
            const f2 fm = (dL_dG * (G * qpow)) * (-2.0f * FIXED_SCALE);
            const f2 fc = (dL_dG * ((mhG * d) * d)) * FIXED_SCALE;
            const int f_cy = cvt_fixed((dL_dG * ((mhG * d.x) * d.y)) * (2.0f * FIXED_SCALE));
            int* const a = acc + (size_t)__float_as_uint(col.w) * ACC_STRIDE;
            // accumulator slots: 0 mean.x 1 mean.y 2 conic.x 3 conic.y 4 conic.z 5 opacity 6 r 7 g 8.. b (four partial sums; geometry_backward adds them up)
            const int v[9] = {cvt_fixed(fm.x), cvt_fixed(fm.y), cvt_fixed(fc.x), f_cy, cvt_fixed(fc.y), f_op, f_r, f_g, f_b};
#pragma unroll
            for (u32 s = 0; s < 9u; s++)
                if (v[s] != 0) atomicAdd(a + s, v[s]);
        }
    }
    __builtin_amdgcn_s_setprio(0);
}

// The kernel.  HELP (one-wave workgroups, LDS sums): blocks whose pixels the forward pass composited through per-pixel lists (lw.flags bit 4 + block,
// longlist.h) are left to long_backward_help, which every wave runs once its own block is done.
template <u32 WPW, bool LDS_SUMS, bool TIMELINE = false, bool PRIO = true, bool HELP = false>
__global__ __launch_bounds__(64 * WPW, 8) void backward_rasterize_kernel(RenderSettings settings, u32 num_tiles_x, u32 num_tiles, const u32* __restrict__ ranges,
                                                                 const u32* __restrict__ instances, const u32* __restrict__ splats,
                                                                 const float* __restrict__ final_T, const u32* __restrict__ n_contrib,
                                                                 const float4* __restrict__ loss_grad, int* __restrict__ acc, u32* __restrict__ acc_dirty,
                                                                 unsigned long long* __restrict__ timeline, LongWork lw) {
    // the accumulators hold sums from here on (acc_clear_if_dirty below, and the consuming forms of geometry_backward, backward.hip)
    if (blockIdx.x == 0u && threadIdx.x == 0u) *acc_dirty = 1u;
    __shared__ float4 s_geo_all[WPW][64];  // centre.x, centre.y, extent.x, extent.y
    __shared__ float4 s_con_all[WPW][64];  // conic.x, conic.y, conic.z, opacity
    __shared__ float4 s_col_all[WPW][64];  // r, g, b, gaussian index (bits)
    __shared__ int s_sum_all[LDS_SUMS ? WPW : 1u][LDS_SUMS ? 8u * 64u : 1u];  // [slot 0..7][pixel lane]: one iteration's contributions
    const u32 blocks_wanted = HELP ? lw.hdr[LL_BLOCKS] : 0u, items_wanted = HELP ? lw.hdr[LL_ITEMS] : 0u;   // (requested now, looked at below)
    // four independent waves per workgroup (one tile): no barrier is ever taken, the grouping only keeps the tile's waves on one
    // CU (shared L1/L2 lines for the entry list) and the workgroup count within the per-CU slot limit.
    u32 tile_id, sub;
    bool mine = true;
    if (WPW == 4u) {
        tile_id = blockIdx.x; sub = threadIdx.x >> 6;
    } else {
        // launch slots b, b + 8, ... share an XCD: slot j of XCD k is block (j & 3) of the XCD's tile number j >> 2; XCD k owns the tiles
        // k, k + 8, k + 16, ... (any tile count: the grid is rounded up and surplus slots leave)
        const u32 k = blockIdx.x & 7u, j = blockIdx.x >> 3;
        tile_id = k + 8u * (j >> 2);
        sub = j & 3u;
        mine = tile_id < num_tiles;
    }
    // (the walk's first two words, requested together with the words that decide whose walk it is: one round trip, not two in a row)
    const u32 range_start = mine ? ranges[tile_id] : 0u, range_end = mine ? ranges[tile_id + 1u] : 0u;
    u32 n_val = 0u;
    if (mine) {   // (the pixel of backward_rasterize_body)
        const u32 lane = threadIdx.x & 63u, W = wd_to_u32(settings.viewport_x), H = wd_to_u32(settings.viewport_y);
        const u32 pixel_x = (tile_id % num_tiles_x) * 16u + (sub & 1u) * 8u + (lane & 7u), pixel_y = (tile_id / num_tiles_x) * 16u + (sub >> 1) * 8u + (lane >> 3);
        if (pixel_x < W && pixel_y < H) n_val = n_contrib[(size_t)pixel_y * W + pixel_x];
    }
    const bool long_on = HELP && ll_frame_on(lw, blocks_wanted, items_wanted);
    if (long_on && mine && ((lw.flags[tile_id] >> (4u + sub)) & 1u)) mine = false;
    const u32 slot = (WPW == 4u) ? sub : 0u;
    // HELP: the walk's pointer arguments are made to sit in scalar registers HERE.  With the help's 38 more argument words in the kernel the compiler fetches
    // arguments where they are first used, some inside the walk behind branches -- and a scalar load that MAY be in flight (they return out of order)
    // turns every partial wait of the walk's inner loop (three record reads, "wait for the first") into a full one: +1.2 % on the kernel, same box
    // (profiles/r08z_kernels_same_box.txt; the loop's code is otherwise identical).
    if (HELP) asm volatile("" ::"s"(ranges), "s"(instances), "s"(splats), "s"(final_T), "s"(n_contrib), "s"(loss_grad), "s"(acc));
    if (mine)
        backward_rasterize_body<WPW, LDS_SUMS, TIMELINE, PRIO>(settings, num_tiles_x, ranges, instances, splats, final_T, n_contrib, loss_grad, acc, timeline, tile_id, sub, range_start, range_end, n_val, s_geo_all[slot],
                                                               s_con_all[slot], s_col_all[slot], s_sum_all[LDS_SUMS ? slot : 0u]);
    if (long_on) long_backward_help(settings, num_tiles_x, final_T, n_contrib, loss_grad, acc, lw);
}

// clearBuffer x4 (tiled-backward-pass.ts:624-627) as a kernel that first looks at the accumulators' state word: the Trainer's forms of
// K17 (geometry_backward_accumulate / geometry_backward_adam) put every row they have read back to zero -- only the ~18 % of rows a view
// touches are non-zero -- and mark the buffer clean, so the next view's clear finds nothing to do (it used to be a 48 MB memset per view at
// c3: 8-19 us).  After the plain K17 (TiledBackwardPass.encode, which leaves the sums readable) or a resize the word says dirty and the
// whole buffer is cleared here.  The word is device state, so a recorded command buffer takes the right branch at every replay.
__global__ __launch_bounds__(256) void acc_clear_if_dirty_kernel(int4* __restrict__ acc, u32 quads, const u32* __restrict__ acc_dirty) {
    WD_STREAM_PRIO();
    if (*acc_dirty == 0u) return;
    const int4 z = make_int4(0, 0, 0, 0);
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += gridDim.x * blockDim.x) acc[i] = z;
}


}  // namespace

int launch_acc_clear_if_dirty(wdgs_device* dev, void* acc, u32 n, void* acc_dirty) {
    const u32 quads = std::max(n, 1u) * (ACC_STRIDE / 4u);
    WDGS_LAUNCH(dev, "acc_clear", acc_clear_if_dirty_kernel, dim3(std::min(ceil_div(quads, 256u * 8u), 2048u)), dim3(256), 0, (int4*)acc, quads, (const u32*)acc_dirty);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_backward_rasterize(wdgs_device* dev, const RenderSettings& st, u32 num_tiles_x, u32 num_tiles_y, const void* ranges, const void* instances,
                              const void* splats, const void* final_t, const void* n_contrib, const void* loss_grad, void* acc, void* acc_dirty, const LongWork* long_work) {
    const u32 tiles = num_tiles_x * num_tiles_y;
    if (tiles == 0) return WDGS_OK;
    // one 8x8 block (one wave) per workgroup: nothing is shared inside a tile's workgroup but cache lines, and single-wave workgroups are
    // placed as soon as ONE wave slot is free -- a shorter tail: 303 -> 295.5 us at c3, same box (r03l).  WDGS_BWR_WPW=4: workgroup = tile.
    static const bool one_wave = !(std::getenv("WDGS_BWR_WPW") && std::getenv("WDGS_BWR_WPW")[0] == '4');
    // WDGS_BWR_SUMS=butterfly: round 2's register-only reduction (same-box A/B; the LDS form is 6 KB of LDS per wave instead of 4)
    static const bool lds_sums = !(std::getenv("WDGS_BWR_SUMS") && std::getenv("WDGS_BWR_SUMS")[0] == 'b');
    // WDGS_BWR_PAD_LDS=<bytes>: unused dynamic LDS per workgroup -- an occupancy experiment (fewer resident waves of this kernel per CU)
    static const u32 pad_lds = std::getenv("WDGS_BWR_PAD_LDS") ? (u32)std::atoi(std::getenv("WDGS_BWR_PAD_LDS")) : 0u;
    const LongWork lw = long_work ? *long_work : LongWork{};
#define WDGS_BWR_ARGS st, num_tiles_x, tiles, (const u32*)ranges, (const u32*)instances, (const u32*)splats, (const float*)final_t, (const u32*)n_contrib, (const float4*)loss_grad, \
                      (int*)acc, (u32*)acc_dirty
    if (one_wave) {
        const u32 slots = ceil_div(tiles, 8u) * 8u * 4u;   // 4 blocks per tile, tiles rounded up to a multiple of the 8 XCDs
        // WDGS_BWR_PRIO=0: the form without issue priorities (same-box A/B)
        static const bool no_prio = std::getenv("WDGS_BWR_PRIO") && std::getenv("WDGS_BWR_PRIO")[0] == '0';
        const bool prio = !no_prio && slots <= 8192u;  // (a launch that does not fit the chip's 8 192 wave slots gains nothing: c3 +-0)
        // WDGS_BWR_TIMELINE=<file> (measurement tool; eager launches only): per-wave records of this launch are appended to the file
        static const char* const timeline_file = std::getenv("WDGS_BWR_TIMELINE");
        if (timeline_file && lds_sums && !dev->capturing) {
            unsigned long long* tl = nullptr;
            const size_t bytes = (size_t)slots * 4u * sizeof(unsigned long long);
            WDGS_CHECK_HIP(hipMalloc((void**)&tl, bytes));
            WDGS_CHECK_HIP(hipMemsetAsync(tl, 0, bytes, dev->stream));
            if (prio) hipLaunchKernelGGL((backward_rasterize_kernel<1u, true, true, true>), dim3(slots), dim3(64), pad_lds, dev->stream, WDGS_BWR_ARGS, tl, LongWork{});
            else hipLaunchKernelGGL((backward_rasterize_kernel<1u, true, true, false>), dim3(slots), dim3(64), pad_lds, dev->stream, WDGS_BWR_ARGS, tl, LongWork{});
            std::vector<unsigned long long> host((size_t)slots * 4u);
            WDGS_CHECK_HIP(hipMemcpyAsync(host.data(), tl, bytes, hipMemcpyDeviceToHost, dev->stream));
            WDGS_CHECK_HIP(hipStreamSynchronize(dev->stream));
            (void)hipFree(tl);
            if (FILE* f = std::fopen(timeline_file, "ab")) { const u32 head[2] = {slots, tiles}; std::fwrite(head, 4, 2, f); std::fwrite(host.data(), 8, host.size(), f); std::fclose(f); }
            WDGS_CHECK_HIP(hipGetLastError());
            return WDGS_OK;
        }
        unsigned long long* const no_tl = nullptr;
        if (lds_sums && lw.hdr && lw.threshold) {   // (long tile lists, longlist.h)
            if (prio) WDGS_LAUNCH(dev, "backward_rasterize", (backward_rasterize_kernel<1u, true, false, true, true>), dim3(slots), dim3(64), pad_lds, WDGS_BWR_ARGS, no_tl, lw);
            else WDGS_LAUNCH(dev, "backward_rasterize", (backward_rasterize_kernel<1u, true, false, false, true>), dim3(slots), dim3(64), pad_lds, WDGS_BWR_ARGS, no_tl, lw);
        } else if (lds_sums && !prio) {
            WDGS_LAUNCH(dev, "backward_rasterize", (backward_rasterize_kernel<1u, true, false, false>), dim3(slots), dim3(64), pad_lds, WDGS_BWR_ARGS, no_tl, lw);
        } else if (lds_sums) {
            WDGS_LAUNCH(dev, "backward_rasterize", (backward_rasterize_kernel<1u, true>), dim3(slots), dim3(64), pad_lds, WDGS_BWR_ARGS, no_tl, lw);
        } else {
            WDGS_LAUNCH(dev, "backward_rasterize", (backward_rasterize_kernel<1u, false>), dim3(slots), dim3(64), pad_lds, WDGS_BWR_ARGS, no_tl, lw);
        }
    } else {
        unsigned long long* const no_tl = nullptr;
        if (lds_sums) WDGS_LAUNCH(dev, "backward_rasterize", (backward_rasterize_kernel<4u, true>), dim3(tiles), dim3(256), pad_lds, WDGS_BWR_ARGS, no_tl, lw);
        else WDGS_LAUNCH(dev, "backward_rasterize", (backward_rasterize_kernel<4u, false>), dim3(tiles), dim3(256), pad_lds, WDGS_BWR_ARGS, no_tl, lw);
    }
#undef WDGS_BWR_ARGS
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}
