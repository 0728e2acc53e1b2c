// Front-to-back alpha compositing (K14), one WAVE (64-thread workgroup) per 8x8 pixel block, four per 16x16 tile.
//
// Replaces tiled_rasterize (src/shaders/tiled-rasterizer.wgsl:82-273): a fixed 32 x 256 batch loop with three
// barriers per batch even when empty, a 48-byte AoS LDS record and no early-out.  Here:
//   * a wave walks its tile's sorted entries in chunks of 64 (lane = entry): (key, index) two chunks ahead and the 24-byte
//     Splat one chunk ahead are fetched into registers, so both global round trips overlap with compositing;
//   * the chunk is compacted to the splats whose extent box overlaps the wave's 8x8 block (ballot + prefix popcount, order
//     preserving) into a 3 KB wave-private LDS record set; the per-pixel extent test of the reference still decides;
//   * the inner loop reads the compacted records by broadcast ds_read_b128, one iteration ahead of their use (two iterations per loop
//     trip, the two register sets swapping roles);
//   * no workgroup barrier exists: a wave stops as soon as ITS 64 pixels are saturated (A > 0.99) or the tile's entries
//     end -- after saturation the reference's loop `continue`s without touching C, A or last_contributor (lines 224-226),
//     so stopping cannot change an output.  (A 256-thread version spent half its wave-cycles waiting at barriers for the
//     slowest of its four waves: profiles/r01a_pmc_summary.txt.)
//   * the four waves of a tile share a workgroup (hence a CU and an XCD), so the tile's entries and splats are fetched once and
//     served to the other three waves by L1/L2; tiles are dealt to the 8 XCDs round-robin (blockIdx order) -- giving each XCD one
//     contiguous band of the image instead was measured 15-30 % slower (the bands' loads differ, the XCDs finish apart).
// Bound: fp32 VALU issue -- 33 wave-instructions per (wave, splat) iteration incl. the deterministic exp (12), at 0.86 of the rate this chip
// sustains for a pure FMA stream -- not HBM.
// Arithmetic is the pinned contraction of DESIGN.md "raster math", bit-identical to the parity oracle.
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dmath.h"
#include "longlist.h"

namespace {

// WPW = waves per workgroup: 1 (one 8x8 block per workgroup, a tile's four blocks dispatched back to back on one XCD) or 4 (workgroup = tile);
// see backward_raster.hip.
// TIMELINE (measurement tool, WDGS_FWR_TIMELINE=<file>, eager launches): every wave leaves {start, end} of the 100 MHz wall clock, where it ran and how
// many records it composited -- the format of backward_raster.hip's, read by scripts/bwr_timeline.py.
// EXACT: the tile holds a Splat with a NaN or an infinity among its fp16 fields (project.hip marks such tiles).  The fast forms below assume
// ordinary operands -- hardware min / max / med3 return the operand that is not a NaN, the in-range exp never sees one -- while the parity oracle
// evaluates WGSL's own formulas (min(e1, e2) = e2 < e1 ? e2 : e1, clamp = min(max(e, lo), hi): a NaN stays a NaN) with the full exp.  EXACT takes
// every such operation in the oracle's form, so that a tile of non-finite Splats -- what a long run of the reference's schedule collects in tile 0 --
// composites to the same bits: a NaN alpha makes the pixel's sums NaN for good (it never saturates, n_contrib keeps following the finite alphas).
template <bool GAUSSIAN_MODE, u32 WPW, bool TIMELINE, bool EXACT>
__device__ __attribute__((always_inline)) void rasterize_body(const RenderSettings& settings, const TileInfo& ti, const u32* __restrict__ splats, u32 num_splats,
                           const u32* __restrict__ ranges, const u32* __restrict__ sorted_keys,
                           const u32* __restrict__ sorted_vals, const u32* __restrict__ count_ptr, u32 max_entries,
                           u32* __restrict__ out_rgba8, float* __restrict__ out_alpha, u32* __restrict__ out_ncontrib, u32 issue_priority,
                           unsigned long long* __restrict__ timeline, u32 tile_id, u32 sub, u32 lane, u32 total /* *count_ptr */, u32 start /* ranges[tile_id] */,
                           float4* s_geo, float4* s_con, float4* s_col) {
    const unsigned long long t_start = TIMELINE ? wall_clock64() : 0ull;
    u32 iterations = 0u;
    const u32 tile_x = tile_id % ti.num_tiles_x, tile_y = tile_id / ti.num_tiles_x;
    const u32 bx = tile_x * 16u + (sub & 1u) * 8u, by = tile_y * 16u + (sub >> 1) * 8u;  // block origin
    const u32 pixel_x = bx + (lane & 7u), pixel_y = by + (lane >> 3);
    const float vx = settings.viewport_x, vy = settings.viewport_y;
    const u32 W = wd_to_u32(vx), H = wd_to_u32(vy);
    const bool in_bounds = pixel_x < W && pixel_y < H;
    const float px = (float)pixel_x + 0.5f, py = (float)pixel_y + 0.5f;
    const float blk_x0 = (float)bx + 0.5f, blk_x1 = (float)bx + 7.5f, blk_y0 = (float)by + 0.5f, blk_y1 = (float)by + 7.5f;
    const float cap = (settings.max_splat_radius_px > 0.0f) ? settings.max_splat_radius_px : 1e9f;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    float cr = 0.0f, cg = 0.0f, cb = 0.0f, A = 0.0f;
    u32 last_contributor = 0u;

    if (__any(in_bounds) && start < total) {  // 0xFFFFFFFF (empty tile) fails the second test too
        const u32 want_key = tile_id + 1u;
        auto fetch_kv = [&](u32 c, u32& key, u32& val) {
            const u32 pos = c * 64u + lane;  // position in the tile's list
            const u32 entry = start + pos;
            const bool in_range = entry < total && (max_entries == 0u || pos < max_entries);
            key = in_range ? sorted_keys[entry] : 0u;
            val = in_range ? sorted_vals[entry] : 0xFFFFFFFFu;
        };
        u32 key_c, val_c, key_n, val_n;
        fetch_kv(0u, key_c, val_c);
        fetch_kv(1u, key_n, val_n);
        bool valid = (key_c >> 16u) == want_key && val_c < num_splats;
        uint2 w01 = make_uint2(0u, 0u), w23 = w01, w45 = w01;
        if (valid) {
            const uint2* sp = reinterpret_cast<const uint2*>(splats + (size_t)val_c * 6);
            w01 = sp[0]; w23 = sp[1]; w45 = sp[2];
        }
        // issue priority from the entries the wave may still have to walk: longest remaining list first (backward_raster.hip, PRIO).  The
        // next tile's start bounds this tile's list (the end of all entries if that tile is empty); arbitration only, results cannot depend on it.
        const u32 next_start = ranges[tile_id + 1u];
        const u32 list_len = ((next_start > start && next_start < total) ? next_start : total) - start;
        bool dead = false;   // (EXACT; uniform) no pixel of the block can still change its sums
        for (u32 chunk = 0;; chunk++) {
            // entries of a tile are contiguous, so the valid lanes are a prefix of the chunk
            const unsigned long long vmask = __ballot(valid);
            if (vmask == 0ull) break;
            if (issue_priority) {
                // (as a scalar: exec-masked s_setprio would all execute, backward_raster.hip)
                const u32 done = chunk * 64u, left = (u32)__builtin_amdgcn_readfirstlane((int)((list_len > done) ? list_len - done : 0u));
                if (left >= 128u) __builtin_amdgcn_s_setprio(3);
                else if (left >= 64u) __builtin_amdgcn_s_setprio(2);
                else if (left >= 32u) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
            // ---- this lane's entry: overlap test against the wave's block (conservative and exact per axis: a splat is dropped
            //      only if the nearest block pixel already fails the per-pixel test |p - c| > extent, which is monotone in p)
            const float cx = (wd_unpack_lo(w01.x) * 0.5f + 0.5f) * vx;
            const float cy = (wd_unpack_hi(w01.x) * -0.5f + 0.5f) * vy;
            // (EXACT: WGSL's min keeps a NaN extent, which then passes every "outside" test -- fminf would turn it into the cap)
            const float ex = EXACT ? wd_min(wd_unpack_lo(w01.y), cap) : fminf(wd_unpack_lo(w01.y), cap);
            const float ey = EXACT ? wd_min(wd_unpack_hi(w01.y), cap) : fminf(wd_unpack_hi(w01.y), cap);
            bool ok = valid && !((blk_x0 - cx) > ex || (cx - blk_x1) > ex || (blk_y0 - cy) > ey || (cy - blk_y1) > ey);
            if (EXACT && dead) {
                // Every pixel of the block is saturated or holds NaN sums (below).  A saturated pixel skips every record; a NaN pixel can only still
                // change its n_contrib, and only at a record whose alpha is a number >= 1/255.  A record with a NaN centre, conic or opacity has a NaN
                // alpha at every pixel (the NaN reaches the exponent's argument through either FMA, or the product with the opacity) and leaves such a
                // pixel exactly as it is: it is dropped here.  Late in a run of the reference's schedule tile 0 holds ~11 000 such records behind its
                // ~60 real ones (a NaN depth sorts last): 2.1 ms of walking them one by one become ~175 chunk headers.
                const bool nan_rec = __builtin_isunordered(cx, cy) | __builtin_isunordered(wd_unpack_lo(w23.x), wd_unpack_hi(w23.x)) |
                                     __builtin_isunordered(wd_unpack_lo(w23.y), wd_unpack_hi(w45.y));
                ok = ok && !nan_rec;
            }
            const unsigned long long m = __ballot(ok);
            const u32 cnt = (u32)__popcll(m);
            if (TIMELINE) iterations += cnt;
            if (ok) {
                const u32 slot = (u32)__popcll(m & lt_mask);
                s_geo[slot] = make_float4(cx, cy, ex, ey);
                // the exponent's argument is -0.5 * (c.x dx^2 + 2 c.y dx dy + c.z dy^2) (tiled-rasterizer.wgsl:229-231): the factors -0.5 and 2
                // are powers of two, which commute with every rounding of the products and sums they pass through, so they are applied to
                // the conic once per record instead of to the power once per (pixel, splat) -- same bits, one multiplication less
                s_con[slot] = make_float4(-0.5f * wd_unpack_lo(w23.x), -wd_unpack_hi(w23.x), -0.5f * wd_unpack_lo(w23.y), wd_unpack_hi(w45.y));
                s_col[slot] = make_float4(wd_unpack_lo(w45.x), wd_unpack_hi(w45.x), wd_unpack_lo(w45.y), __uint_as_float(chunk * 64u + lane + 1u));
            }
            // issue the next chunk's gather and the (key, index) loads of the chunk after it; they land while this chunk composites
            valid = (key_n >> 16u) == want_key && val_n < num_splats;
            if (valid) {
                const uint2* sp = reinterpret_cast<const uint2*>(splats + (size_t)val_n * 6);
                w01 = sp[0]; w23 = sp[1]; w45 = sp[2];
            }
            fetch_kv(chunk + 2u, key_n, val_n);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS records written above are read below by other lanes
            __builtin_amdgcn_wave_barrier();

            if (GAUSSIAN_MODE) {
                auto composite = [&](const float4 geo, const float4 con, const float4 col) {
                    const float dx = px - geo.x, dy = py - geo.y;
                    // (bitwise, not short-circuit: no branches for the tests)
                    const bool active = ((int)in_bounds & (int)!(fabsf(dx) > geo.z) & (int)!(fabsf(dy) > geo.w) & (int)!(A > 0.99f)) != 0;
                    // (no wave-wide "nobody is active" early-out: after the compaction nearly every splat has an active pixel, and the
                    // test cost two VALU operations per iteration; an iteration without one falls through the empty exec mask below)
                    if (active) {
                        const float t1 = __builtin_fmaf(con.x, dx, con.y * dy);
                        const float xe = __builtin_fmaf(t1, dx, (con.z * dy) * dy);  // = -0.5 * power (the record holds the scaled conic)
                        // exp with its range handling taken out (dmath.h wd_exp_inrange) by ONE clamp of the argument to [-86, 87]
                        // (v_med3_f32), exact where it matters:
                        //  * above 87 -- an indefinite conic after fp16 rounding -- exp is at least 6e37 either way and alpha = min(.., 0.99) = 0.99;
                        //  * below -86, where wd_exp returns 0, alpha is at most exp(-86) = 4.5e-38 rather than 0: a weight that every colour
                        //    sum and the alpha sum absorb without a trace (the smallest weight that matters is above 1e-10 of the sums;
                        //    alone, it quantises to 0 and leaves T = 1 - A = 1); the 1/255 test for n_contrib is far away;
                        //  * a NaN argument (an infinite fp16 conic) leaves v_med3_f32 as -86, i.e. as the same vanishing weight, where
                        //    WGSL's clamp of a NaN alpha would give 0.
                        // G*opacity >= +0 and never NaN here (opacity in (1/128, 1], G finite), so the hardware min equals WGSL's
                        // select-based clamp, and the lower clamp is the identity and is left out.
                        float alpha;
                        if (EXACT) {
                            alpha = wd_clamp(wd_exp(xe) * con.w, 0.0f, 0.99f);   // tiled-rasterizer.wgsl:228-233 as the parity oracle evaluates it
                        } else {
                            const float xc = __builtin_amdgcn_fmed3f(xe, -86.0f, 87.0f);
                            alpha = fminf(wd_exp_inrange(xc) * con.w, 0.99f);
                        }
                        const float w = alpha * (1.0f - A);
                        cr = __builtin_fmaf(col.x, w, cr);
                        cg = __builtin_fmaf(col.y, w, cg);
                        cb = __builtin_fmaf(col.z, w, cb);
                        A = A + w;
                        last_contributor = (alpha >= (1.0f / 255.0f)) ? __float_as_uint(col.w) : last_contributor;
                    }
                };
                // The records of the next iteration are fetched while this one composites: an LDS round trip (~100 cycles) is as long as
                // an iteration's arithmetic, and read at the top of the iteration that uses them it cost a third of the wave's time in
                // waiting.  Two iterations per trip through the loop, so that the two record sets swap roles without register copies.
                if (EXACT) {   // (rare path: the plain loop, fewer live registers)
#pragma unroll 1
                    for (u32 i = 0; i < cnt; i++) composite(s_geo[i], s_con[i], s_col[i]);
                } else {
                float4 geo_a = s_geo[0], con_a = s_con[0], col_a = s_col[0];  // (cnt == 0: a stale record, never used)
                for (u32 i = 0; i < cnt; i += 2u) {
                    const float4 geo_b = s_geo[i + 1u], con_b = s_con[i + 1u], col_b = s_col[i + 1u];  // (i + 1 <= 64: the spare record)
                    composite(geo_a, con_a, col_a);
                    if (i + 1u >= cnt) break;
                    geo_a = s_geo[i + 2u]; con_a = s_con[i + 2u]; col_a = s_col[i + 2u];   // (i + 2 <= 64)
                    composite(geo_b, con_b, col_b);
                }
                }
            } else {
                for (u32 i = 0; i < cnt; i++) {
                    // point-cloud preview (tiled-rasterizer.wgsl:212-222): paints yellow discs, no saturation test
                    const float4 geo = s_geo[i];
                    const float dx = px - geo.x, dy = py - geo.y;
                    const bool inside = ((int)in_bounds & (int)!(fabsf(dx) > geo.z) & (int)!(fabsf(dy) > geo.w)) != 0;
                    if (inside) {
                        const float dist_sq = dx * dx + dy * dy;
                        const float limit = fminf(settings.point_size_px, cap);
                        if (dist_sq <= limit * limit) {
                            cr = 1.0f; cg = 1.0f; cb = 0.0f; A = 1.0f;
                            last_contributor = __float_as_uint(s_col[i].w);
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();  // all lanes are done reading the records before the next chunk overwrites them
            // every pixel of this wave saturated -> nothing later can change an output of this wave
            if (GAUSSIAN_MODE && !__any(in_bounds && !(A > 0.99f))) break;
            if (EXACT && GAUSSIAN_MODE) dead = !__any(in_bounds && (A <= 0.99f));   // (false for a saturated and for a NaN sum)
            if (vmask != ~0ull) break;  // the tile's list ended inside this chunk
        }
    }

    if (in_bounds) {
        const size_t p = (size_t)pixel_y * W + pixel_x;
        const u32 r8 = wd_to_u32(fminf(fmaxf(cr, 0.0f), 1.0f) * 255.0f + 0.5f);
        const u32 g8 = wd_to_u32(fminf(fmaxf(cg, 0.0f), 1.0f) * 255.0f + 0.5f);
        const u32 b8 = wd_to_u32(fminf(fmaxf(cb, 0.0f), 1.0f) * 255.0f + 0.5f);
        out_rgba8[p] = r8 | (g8 << 8) | (b8 << 16) | 0xFF000000u;
        out_alpha[p] = 1.0f - A;
        out_ncontrib[p] = last_contributor;
    }
    if (TIMELINE && lane == 0u) {
        u32 hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(4, 0, 32)" : "=s"(hw_id));    // HW_REG_HW_ID
        asm volatile("s_getreg_b32 %0, hwreg(20, 0, 32)" : "=s"(xcc_id));  // HW_REG_XCC_ID
        unsigned long long* const rec = timeline + ((size_t)blockIdx.x * WPW + (threadIdx.x >> 6)) * 4u;
        rec[0] = t_start; rec[1] = wall_clock64(); rec[2] = ((unsigned long long)xcc_id << 32) | hw_id; rec[3] = iterations;
    }
}


// ================================================================================================ long tile lists (longlist.h): the forward tasks
// Shared by the count and the fill task of an item (block, chunk): the chunk's 64 entries, unpacked, and the records that overlap the block compacted
// into the wave's record set -- exactly what the wave-per-block walk builds (EXACT forms: the tasks serve non-finite tiles too).  Returns the number
// of records kept; *have / *g_idx: this lane's own entry.
struct LongCtx {
    RenderSettings settings; TileInfo ti;
    const u32 *splats, *ranges, *sorted_keys, *sorted_vals, *count_ptr;
    u32 num_splats;
    u32 *out_rgba8; float* out_alpha; u32* out_ncontrib;
    const u32 *nf_stamp, *nf_frame;
};
WD_DEV u32 long_chunk_records(const LongCtx& c, const LongBlock& blk, u32 chunk, u32 lane, float4* s_geo, float4* s_con, float4* s_col, float4* rec_out /*nullable: [64][3]*/) {
    const float vx = c.settings.viewport_x, vy = c.settings.viewport_y;
    const float cap = (c.settings.max_splat_radius_px > 0.0f) ? c.settings.max_splat_radius_px : 1e9f;
    const u32 tile_x = blk.tile % c.ti.num_tiles_x, tile_y = blk.tile / c.ti.num_tiles_x;
    const u32 bx = tile_x * 16u + (blk.sub & 1u) * 8u, by = tile_y * 16u + (blk.sub >> 1) * 8u;
    const float blk_x0 = (float)bx + 0.5f, blk_x1 = (float)bx + 7.5f, blk_y0 = (float)by + 0.5f, blk_y1 = (float)by + 7.5f;
    const u32 total = *c.count_ptr, start = c.ranges[blk.tile];
    const u32 pos = chunk * 64u + lane, entry = start + pos;
    const bool in_range = entry < total;
    const u32 key = in_range ? c.sorted_keys[entry] : 0u;
    const u32 val = in_range ? c.sorted_vals[entry] : 0xFFFFFFFFu;
    const bool valid = (key >> 16u) == blk.tile + 1u && val < c.num_splats;
    uint2 w01 = make_uint2(0u, 0u), w23 = w01, w45 = w01;
    if (valid) {
        const uint2* sp = reinterpret_cast<const uint2*>(c.splats + (size_t)val * 6);
        w01 = sp[0]; w23 = sp[1]; w45 = sp[2];
    }
    const float cx = (wd_unpack_lo(w01.x) * 0.5f + 0.5f) * vx;
    const float cy = (wd_unpack_hi(w01.x) * -0.5f + 0.5f) * vy;
    const float ex = wd_min(wd_unpack_lo(w01.y), cap), ey = wd_min(wd_unpack_hi(w01.y), cap);
    bool ok = valid && !((blk_x0 - cx) > ex || (cx - blk_x1) > ex || (blk_y0 - cy) > ey || (cy - blk_y1) > ey);
    // nan_rec: a NaN centre, conic or opacity -- the record's alpha is a NaN at EVERY pixel (the NaN reaches the exponent's argument through either FMA,
    // or the product with the opacity).  A pixel that has met such a record inside its box holds NaN sums from then on, and a further one leaves it
    // exactly as it is (w = NaN again; n_contrib follows numbers only).  If such a record's centre or extent is a NaN, EVERY pixel of the block is inside
    // its box (the box tests are comparisons): behind the first such record of the chunk every nan_rec of the chunk is dropped here -- the late
    // regime's tile 0, thousands of non-finite Gaussians behind ~60 real ones, keeps one record per chunk.
    const bool nan_rec = ok && (__builtin_isunordered(cx, cy) | __builtin_isunordered(wd_unpack_lo(w23.x), wd_unpack_hi(w23.x)) | __builtin_isunordered(wd_unpack_lo(w23.y), wd_unpack_hi(w45.y)));
    const unsigned long long universal = __ballot(nan_rec && (__builtin_isunordered(cx, cy) | __builtin_isunordered(ex, ey)));   // (NaN alpha AND inside at every pixel)
    if (universal != 0ull) ok = ok && !(nan_rec && lane > (u32)__builtin_ctzll(universal));
    const unsigned long long m = __ballot(ok);
    if (ok) {
        const u32 slot = (u32)__popcll(m & ((1ull << lane) - 1ull));
        s_geo[slot] = make_float4(cx, cy, ex, ey);
        s_con[slot] = make_float4(-0.5f * wd_unpack_lo(w23.x), -wd_unpack_hi(w23.x), -0.5f * wd_unpack_lo(w23.y), wd_unpack_hi(w45.y));
        // r | g and b as the Splat's fp16 bits; z: 1 = nan_rec
        s_col[slot] = make_float4(__uint_as_float(w45.x), __uint_as_float(w45.y & 0xFFFFu), nan_rec ? 1.0f : 0.0f, __uint_as_float(pos + 1u));
    }
    if (rec_out && valid) {   // the entry as the backward walk wants it (backward_raster.hip: -conic / 2 throughout, the Gaussian's index)
        float4* r = rec_out + (size_t)lane * 3u;
        r[0] = make_float4(cx, cy, ex, ey);
        r[1] = make_float4(-0.5f * wd_unpack_lo(w23.x), -0.5f * wd_unpack_hi(w23.x), -0.5f * wd_unpack_lo(w23.y), wd_unpack_hi(w45.y));
        r[2] = make_float4(wd_unpack_lo(w45.x), wd_unpack_hi(w45.x), wd_unpack_lo(w45.y), __uint_as_float(val));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return (u32)__popcll(m);
}

// ---- the four kinds of task.  Everything a task decides on is uniform across the wave; no lane-number branches (longlist.h).
// count / fill of item (block, chunk): the chunk's records; per pixel the records whose box holds it -- counted, or (fill) appended to the pixel's list
WD_DEV void long_task_item(const LongCtx& c, const LongWork& lw, u32 item, bool fill, u32 lane, float4* s_geo, float4* s_con, float4* s_col) {
    const u32 lb = lw.item_block[item];
    if (lb == 0xFFFFFFFFu) return;   // (a slot of a tile that found no room)
    const LongBlock blk = lw.blocks[lb];
    LongSync* const sy = lw.sync + lb;
    u32 row_base = 0u;
    if (fill) {
        const bool ready = ll_wait(&sy->scanned, 1u, lw.hdr, 0x201u);
        row_base = ready ? (u32)__builtin_amdgcn_readfirstlane((int)ll_ld(&sy->row_base)) : LL_NO_ROWS;
    }
    if (row_base != LL_NO_ROWS) {
        const u32 chunk = item - blk.first_item;
        const u32 cnt = long_chunk_records(c, blk, chunk, lane, s_geo, s_con, s_col, fill ? nullptr : lw.records + (size_t)item * 192u);
        const u32 W = wd_to_u32(c.settings.viewport_x), H = wd_to_u32(c.settings.viewport_y);
        const u32 tile_x = blk.tile % c.ti.num_tiles_x, tile_y = blk.tile / c.ti.num_tiles_x;
        const u32 pixel_x = tile_x * 16u + (blk.sub & 1u) * 8u + (lane & 7u), pixel_y = tile_y * 16u + (blk.sub >> 1) * 8u + (lane >> 3);
        const bool in_bounds = pixel_x < W && pixel_y < H;
        const float px = (float)pixel_x + 0.5f, py = (float)pixel_y + 0.5f;
        u32 k = fill ? ll_ld(&lw.off[(size_t)item * 64u + lane]) : 0u;
        // A pixel that has met a record with a NaN alpha holds NaN sums from then on: a further such record leaves it exactly as it is (w = NaN again,
        // n_contrib follows numbers only).  Of the NaN records of a chunk a pixel's list therefore keeps the first one only -- which is what keeps the lists
        // of the late regime's tile 0 short: ~60 real records and one element per chunk of the thousands of non-finite Gaussians piled up behind them.
        bool poisoned = false;
        for (u32 i = 0; i < cnt; i++) {
            const float4 geo = s_geo[i];
            const float4 col = s_col[i];
            const float dx = px - geo.x, dy = py - geo.y;
            const bool in_box = ((int)in_bounds & (int)!(fabsf(dx) > geo.z) & (int)!(fabsf(dy) > geo.w)) != 0;   // (tiled-rasterizer.wgsl:205-207)
            const bool nan_rec = col.z != 0.0f;
            const bool inside = in_box && !(nan_rec && poisoned);
            poisoned = poisoned || (in_box && nan_rec);
            if (inside && fill) {
                const float4 con = s_con[i];
                const float t1 = __builtin_fmaf(con.x, dx, con.y * dy);
                const float xe = __builtin_fmaf(t1, dx, (con.z * dy) * dy);
                const float alpha = wd_clamp(wd_exp(xe) * con.w, 0.0f, 0.99f);
                u32* const e = lw.rows + (((size_t)row_base + k) * 64u + lane) * 4u;
                ll_st(e, __float_as_uint(alpha)); ll_st(e + 1, __float_as_uint(col.x)); ll_st(e + 2, __float_as_uint(col.y)); ll_st(e + 3, __float_as_uint(col.w));
            }
            k += inside ? 1u : 0u;
        }
        __builtin_amdgcn_wave_barrier();   // (the record set is rewritten by the wave's next task)
        if (!fill) ll_st(&lw.cnt[(size_t)item * 64u + lane], k);
    }
    ll_signal(fill ? &sy->filled : &sy->counted, lane);
}

// scan of block lb: where each chunk's part of every pixel's list starts; the block's rows
WD_DEV void long_task_scan(const LongWork& lw, u32 lb, u32 lane) {
    const LongBlock blk = lw.blocks[lb];
    LongSync* const sy = lw.sync + lb;
    if (blk.chunks == 0u) return;   // (a block record of a tile that found no room)
    u32 base = LL_NO_ROWS, rows = 0u;
    if (ll_wait(&sy->counted, blk.chunks, lw.hdr, 0x101u)) {
        // (the counts were stored through to memory, ll_st; ONE invalidation of this XCD's caches here lets the loop read them with ordinary loads,
        // eight in flight -- read one by one with coherent loads the 177 chunks of the late regime's tile cost 177 memory round trips)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        u32 run = 0u;
        const u32* const cnt0 = lw.cnt + (size_t)blk.first_item * 64u + lane;
        u32* const off0 = lw.off + (size_t)blk.first_item * 64u + lane;
        for (u32 ch = 0; ch < blk.chunks; ch += 8u) {
            u32 v[8];
#pragma unroll
            for (u32 q = 0; q < 8u; q++) v[q] = (ch + q < blk.chunks) ? cnt0[(size_t)(ch + q) * 64u] : 0u;
#pragma unroll
            for (u32 q = 0; q < 8u; q++) {
                if (ch + q < blk.chunks) ll_st(off0 + (size_t)(ch + q) * 64u, run);
                run += v[q];
            }
        }
        ll_st(&lw.total[(size_t)lb * 64u + lane], run);
        rows = run;
#pragma unroll
        for (u32 d = 32; d >= 1; d >>= 1) rows = max(rows, (u32)__shfl_xor((int)rows, (int)d, 64));
        rows = (u32)__builtin_amdgcn_readfirstlane((int)rows);
        // (per-pixel lists pay when they are much shorter than the tile's: a trip of the per-pixel backward walk fetches its record from memory and
        // adds its contributions itself -- several times the cost of a record of the wave-per-block walk)
        const bool worth = rows * 6u <= blk.chunks * 64u;
        const u32 want = (worth && lane == 0u) ? rows : 0u;   // (one lane asks for the rows, the others for none)
        atomicAdd(&lw.hdr[LL_ROWS_WANTED], want);
        const u32 b0 = (u32)__builtin_amdgcn_readfirstlane((int)atomicAdd(&lw.hdr[LL_ROWS], want));
        if (worth && b0 + rows <= lw.max_rows && b0 + rows >= b0) base = b0;
    }
    ll_st(&sy->row_base, base);   // (uniform values, stored by every lane)
    ll_st(&sy->rows, rows);
    ll_signal(&sy->scanned, lane);
}

// walk of block lb: lane = pixel, trip j = element j of every pixel's list
WD_DEV void long_task_walk(const LongCtx& c, const LongWork& lw, u32 lb, u32 lane, float4* s_geo, float4* s_con, float4* s_col) {
    const LongBlock blk = lw.blocks[lb];
    LongSync* const sy = lw.sync + lb;
    if (blk.chunks == 0u) return;
    const bool filled = ll_wait(&sy->filled, blk.chunks, lw.hdr, 0x301u);
    const u32 row_base = filled ? (u32)__builtin_amdgcn_readfirstlane((int)ll_ld(&sy->row_base)) : LL_NO_ROWS;
    if (row_base == LL_NO_ROWS) {   // no rows: the wave-per-block walk, here (the block's main wave has left it alone)
        if (c.nf_stamp == nullptr || c.nf_stamp[blk.tile] == *c.nf_frame)
            rasterize_body<true, 4u, false, true>(c.settings, c.ti, c.splats, c.num_splats, c.ranges, c.sorted_keys, c.sorted_vals, c.count_ptr, 0u, c.out_rgba8, c.out_alpha,
                                                  c.out_ncontrib, 0u, nullptr, blk.tile, blk.sub, lane, *c.count_ptr, c.ranges[blk.tile], s_geo, s_con, s_col);
        else
            rasterize_body<true, 4u, false, false>(c.settings, c.ti, c.splats, c.num_splats, c.ranges, c.sorted_keys, c.sorted_vals, c.count_ptr, 0u, c.out_rgba8, c.out_alpha,
                                                   c.out_ncontrib, 0u, nullptr, blk.tile, blk.sub, lane, *c.count_ptr, c.ranges[blk.tile], s_geo, s_con, s_col);
        return;
    }
    const u32 W = wd_to_u32(c.settings.viewport_x), H = wd_to_u32(c.settings.viewport_y);
    const u32 tile_x = blk.tile % c.ti.num_tiles_x, tile_y = blk.tile / c.ti.num_tiles_x;
    const u32 pixel_x = tile_x * 16u + (blk.sub & 1u) * 8u + (lane & 7u), pixel_y = tile_y * 16u + (blk.sub >> 1) * 8u + (lane >> 3);
    const bool in_bounds = pixel_x < W && pixel_y < H;
    const u32 tot = ll_ld(&lw.total[(size_t)lb * 64u + lane]);
    const u32 trips = min((u32)__builtin_amdgcn_readfirstlane((int)ll_ld(&sy->rows)), blk.chunks * 64u);
    float cr = 0.0f, cg = 0.0f, cb = 0.0f, A = 0.0f;
    u32 last_contributor = 0u, jl = 0u;
    // (the rows were stored through to memory by the fill tasks, ll_st: one invalidation of this XCD's caches, then ordinary 16-byte loads, eight
    // trips ahead of their use -- what is serial per pixel is A -> w -> A and the three colour FMAs)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const float4* const rows = reinterpret_cast<const float4*>(lw.rows) + (size_t)row_base * 64u + lane;
    auto row = [&](u32 j) -> float4 { return (j < tot) ? rows[(size_t)j * 64u] : make_float4(0.f, 0.f, 0.f, 0.f); };
    float4 ring[8];
#pragma unroll
    for (u32 q = 0; q < 8u; q++) ring[q] = row(q);
    bool done = false;
    for (u32 j0 = 0; j0 < trips && !done; j0 += 8u) {
#pragma unroll
        for (u32 q = 0; q < 8u; q++) {
            const u32 j = j0 + q;
            const float4 e = ring[q];
            ring[q] = row(j + 8u);
            const bool active = j < tot && !(A > 0.99f);   // (tiled-rasterizer.wgsl:224-226: a saturated pixel skips the record)
            if (active) {
                const float alpha = e.x;
                const float w = alpha * (1.0f - A);
                cr = __builtin_fmaf(wd_unpack_lo(__float_as_uint(e.y)), w, cr);
                cg = __builtin_fmaf(wd_unpack_hi(__float_as_uint(e.y)), w, cg);
                cb = __builtin_fmaf(wd_unpack_lo(__float_as_uint(e.z)), w, cb);
                A = A + w;
                const bool contributes = alpha >= (1.0f / 255.0f);
                last_contributor = contributes ? __float_as_uint(e.w) : last_contributor;
                jl = contributes ? j + 1u : jl;
            }
        }
        done = !__any(j0 + 8u < tot && !(A > 0.99f));   // nothing left that could change a sum
    }
    if (in_bounds) {
        const size_t p = (size_t)pixel_y * W + pixel_x;
        const u32 r8 = wd_to_u32(fminf(fmaxf(cr, 0.0f), 1.0f) * 255.0f + 0.5f);
        const u32 g8 = wd_to_u32(fminf(fmaxf(cg, 0.0f), 1.0f) * 255.0f + 0.5f);
        const u32 b8 = wd_to_u32(fminf(fmaxf(cb, 0.0f), 1.0f) * 255.0f + 0.5f);
        c.out_rgba8[p] = r8 | (g8 << 8) | (b8 << 16) | 0xFF000000u;
        c.out_alpha[p] = 1.0f - A;
        c.out_ncontrib[p] = last_contributor;
    }
    lw.jlast[(size_t)lb * 64u + lane] = jl;
    sy->walked = 1u;   // (uniform; the backward kernel -- a later launch -- reads it and the tile's mark)
    atomicOr(&lw.flags[blk.tile], lane == 0u ? (16u << blk.sub) : 0u);
}

// One wave works the forward queue off: [count tasks][scan tasks][fill tasks][walk tasks] (longlist.h).  s_geo / s_con / s_col: the wave's own record set.
__device__ __attribute__((always_inline)) void long_forward_help(const LongCtx c, const LongWork lw, u32 lane, float4* s_geo, float4* s_con, float4* s_col) {
    const u32 n_items = lw.hdr[LL_ITEMS], n_blocks = lw.hdr[LL_BLOCKS];   // (ll_frame_on: all of them exist)
    const u32 n_tasks = 2u * (n_items + n_blocks), batch = ll_batch(n_tasks);
    __builtin_amdgcn_s_setprio(3);   // the long lists are the frame's longest chains: first in line for the issue slots
    for (u32 t0 = ll_pull(&lw.hdr[LL_FWD_HEAD], n_tasks, batch, lane); t0 != 0xFFFFFFFFu; t0 = ll_pull(&lw.hdr[LL_FWD_HEAD], n_tasks, batch, lane)) {
        const u32 t1 = min(t0 + batch, n_tasks);
        for (u32 t = t0; t < t1; t++) {
            if (t < n_items) long_task_item(c, lw, t, false, lane, s_geo, s_con, s_col);
            else if (t < n_items + n_blocks) long_task_scan(lw, t - n_items, lane);
            else if (t < 2u * n_items + n_blocks) long_task_item(c, lw, t - n_items - n_blocks, true, lane, s_geo, s_con, s_col);
            else long_task_walk(c, lw, t - 2u * n_items - n_blocks, lane, s_geo, s_con, s_col);
        }
    }
    __builtin_amdgcn_s_setprio(0);
}

// The kernel: one wave per 8x8 block; a tile marked as holding a non-finite Splat (nf_stamp[tile] == *nf_frame: project.hip stamps the tiles of such
// Splats with the number the frame's scan kernel then gives the frame) takes the EXACT body, every other tile the fast one.  nf_stamp == nullptr:
// nothing is known about the Splats -- every tile takes the EXACT body.  HELP: the blocks of long tiles (lw.flags) are left to the tasks of longlist.h,
// which every wave helps to work off once its own block is done.
template <bool GAUSSIAN_MODE, u32 WPW, bool TIMELINE = false, bool HELP = false>
__global__ __launch_bounds__(64 * WPW, 8) void rasterize_kernel(RenderSettings settings, TileInfo ti, const u32* __restrict__ splats, u32 num_splats,
                                                        const u32* __restrict__ ranges, const u32* __restrict__ sorted_keys,
                                                        const u32* __restrict__ sorted_vals, const u32* __restrict__ count_ptr, u32 max_entries,
                                                        u32* __restrict__ out_rgba8, float* __restrict__ out_alpha, u32* __restrict__ out_ncontrib, u32 issue_priority,
                                                        unsigned long long* __restrict__ timeline, const u32* __restrict__ nf_stamp, const u32* __restrict__ nf_frame, LongWork lw) {
    // (one record more than a chunk holds: the loop reads one record ahead)
    __shared__ float4 s_geo_all[WPW][65];  // centre.x, centre.y, extent.x, extent.y   (pixels)
    __shared__ float4 s_con_all[WPW][65];  // -0.5*conic.x, -conic.y, -0.5*conic.z, opacity (Gaussian mode: see the record build)
    __shared__ float4 s_col_all[WPW][65];  // r, g, b, position in the tile list + 1 (bits)
    const u32 slot = (WPW == 4u) ? (threadIdx.x >> 6) : 0u;
    float4* const s_geo = s_geo_all[slot];  // wave-private record sets
    float4* const s_con = s_con_all[slot];
    float4* const s_col = s_col_all[slot];
    const u32 blocks_wanted = HELP ? lw.hdr[LL_BLOCKS] : 0u, items_wanted = HELP ? lw.hdr[LL_ITEMS] : 0u;   // (requested now, looked at below)
    // (HELP: the wave's number in a scalar register for the help below, which derives everything it needs from it -- a vector register kept alive across
    // the walk for the help's sake costs the walk a spill per chunk: +3.5 us at c3, profiles/r08w_kernels_same_box.txt)
    const u32 wave_s = (HELP && WPW == 4u) ? (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0u;
    // independent waves (no barrier is ever taken): one per 8x8 block
    u32 tile_id, sub;
    bool mine = true;
    if (WPW == 4u) {
        tile_id = blockIdx.x; sub = threadIdx.x >> 6;
    } else {
        // launch slots b, b + 8, ... share an XCD: slot j of XCD k is block (j & 3) of the XCD's tile number j >> 2; XCD k owns tiles k, k + 8, ...
        const u32 k = blockIdx.x & 7u, j = blockIdx.x >> 3;
        tile_id = k + 8u * (j >> 2);
        sub = j & 3u;
        mine = tile_id < ti.total_tiles;
    }
    // (the walk's first two words, requested together with the words that decide which walk it is: one round trip, not two in a row)
    const u32 total = *count_ptr;
    const u32 start = mine ? ranges[tile_id] : 0xFFFFFFFFu;
    const bool exact = nf_stamp == nullptr || (mine && nf_stamp[tile_id] == *nf_frame);   // (uniform per workgroup)
    const bool long_on = HELP && ll_frame_on(lw, blocks_wanted, items_wanted);
    if (long_on && mine && ((lw.flags[tile_id] >> sub) & 1u)) mine = false;   // (a long list: the tasks composite and write this block)
    if (mine) {
        if (exact)
            rasterize_body<GAUSSIAN_MODE, WPW, TIMELINE, true>(settings, ti, splats, num_splats, ranges, sorted_keys, sorted_vals, count_ptr, max_entries, out_rgba8, out_alpha,
                                                               out_ncontrib, issue_priority, timeline, tile_id, sub, threadIdx.x & 63u, total, start, s_geo, s_con, s_col);
        else
            rasterize_body<GAUSSIAN_MODE, WPW, TIMELINE, false>(settings, ti, splats, num_splats, ranges, sorted_keys, sorted_vals, count_ptr, max_entries, out_rgba8, out_alpha,
                                                                out_ncontrib, issue_priority, timeline, tile_id, sub, threadIdx.x & 63u, total, start, s_geo, s_con, s_col);
    }
    if (long_on)
        long_forward_help(LongCtx{settings, ti, splats, ranges, sorted_keys, sorted_vals, count_ptr, num_splats, out_rgba8, out_alpha, out_ncontrib, nf_stamp, nf_frame}, lw,
                          __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), s_geo_all[wave_s], s_con_all[wave_s], s_col_all[wave_s]);
}

}  // namespace

int launch_rasterize(wdgs_device* dev, const RenderSettings& st, const TileInfo& ti, const void* splats, u32 num_splats, const void* ranges,
                     const void* sorted_keys, const void* sorted_vals, const void* count_ptr, u32 max_batches, void* out_rgba8, void* out_alpha,
                     void* out_ncontrib, const void* nf_stamp, const void* nf_frame, const LongWork* long_work) {
    if (ti.total_tiles == 0) return WDGS_OK;
    const u32 max_entries = max_batches * 256u;  // compat cap: 32 batches x 256 splats per tile; 0 = unlimited
    // (one-wave workgroups help backward_rasterize -- 303 -> 295.5 us -- but not this kernel: 119.2 vs 119.8 us, r03m; workgroup = tile stays)
    static const bool one_wave = std::getenv("WDGS_RASTER_WPW") && std::getenv("WDGS_RASTER_WPW")[0] == '1';
    const u32 slots = ceil_div(ti.total_tiles, 8u) * 8u * 4u;   // 4 blocks per tile, tiles rounded up to a multiple of the 8 XCDs
    // WDGS_FWR_PRIO=0: no issue priorities (same-box A/B)
    static const u32 issue_priority = (std::getenv("WDGS_FWR_PRIO") && std::getenv("WDGS_FWR_PRIO")[0] == '0') ? 0u : 1u;
    const u32 issue_priority_now = (issue_priority && slots <= 8192u) ? 1u : 0u;  // launches whose waves (4 per tile in either workgroup shape) are all resident from the start
    const LongWork lw = long_work ? *long_work : LongWork{};
#define RASTER_ARGS st, ti, (const u32*)splats, num_splats, (const u32*)ranges, (const u32*)sorted_keys, (const u32*)sorted_vals, (const u32*)count_ptr, max_entries, \
                    (u32*)out_rgba8, (float*)out_alpha, (u32*)out_ncontrib, issue_priority_now, (unsigned long long*)nullptr, (const u32*)nf_stamp, (const u32*)nf_frame, lw
    // WDGS_FWR_TIMELINE=<file> (measurement tool; eager launches of the Gaussian mode in its default workgroup shape): per-wave records appended to the file
    static const char* const timeline_file = std::getenv("WDGS_FWR_TIMELINE");
    if (timeline_file && st.gaussian_mode >= 0.5f && !one_wave && !dev->capturing) {
        unsigned long long* tl = nullptr;
        const size_t bytes = (size_t)ti.total_tiles * 4u * 4u * sizeof(unsigned long long);
        WDGS_CHECK_HIP(hipMalloc((void**)&tl, bytes));
        WDGS_CHECK_HIP(hipMemsetAsync(tl, 0, bytes, dev->stream));
        hipLaunchKernelGGL((rasterize_kernel<true, 4u, true>), dim3(ti.total_tiles), dim3(256), 0, dev->stream, st, ti, (const u32*)splats, num_splats, (const u32*)ranges,
                           (const u32*)sorted_keys, (const u32*)sorted_vals, (const u32*)count_ptr, max_entries, (u32*)out_rgba8, (float*)out_alpha, (u32*)out_ncontrib,
                           issue_priority_now, tl, (const u32*)nf_stamp, (const u32*)nf_frame, LongWork{});
        std::vector<unsigned long long> host((size_t)ti.total_tiles * 16u);
        WDGS_CHECK_HIP(hipMemcpyAsync(host.data(), tl, bytes, hipMemcpyDeviceToHost, dev->stream));
        WDGS_CHECK_HIP(hipStreamSynchronize(dev->stream));
        (void)hipFree(tl);
        if (FILE* f = std::fopen(timeline_file, "ab")) { const u32 head[2] = {ti.total_tiles * 4u, ti.total_tiles}; std::fwrite(head, 4, 2, f); std::fwrite(host.data(), 8, host.size(), f); std::fclose(f); }
        WDGS_CHECK_HIP(hipGetLastError());
        return WDGS_OK;
    }
    if (st.gaussian_mode >= 0.5f) {
        // (long tile lists, longlist.h: Gaussian mode, uncapped lists, the default workgroup shape)
        if (one_wave) WDGS_LAUNCH(dev, "rasterize", (rasterize_kernel<true, 1u>), dim3(slots), dim3(64), 0, RASTER_ARGS);
        else if (lw.hdr && lw.threshold && max_entries == 0u) WDGS_LAUNCH(dev, "rasterize", (rasterize_kernel<true, 4u, false, true>), dim3(ti.total_tiles), dim3(256), 0, RASTER_ARGS);
        else WDGS_LAUNCH(dev, "rasterize", (rasterize_kernel<true, 4u>), dim3(ti.total_tiles), dim3(256), 0, RASTER_ARGS);
    } else {
        if (one_wave) WDGS_LAUNCH(dev, "rasterize_points", (rasterize_kernel<false, 1u>), dim3(slots), dim3(64), 0, RASTER_ARGS);
        else WDGS_LAUNCH(dev, "rasterize_points", (rasterize_kernel<false, 4u>), dim3(ti.total_tiles), dim3(256), 0, RASTER_ARGS);
    }
#undef RASTER_ARGS
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}
