// Front-to-back alpha compositing, one 256-thread workgroup per 16x16 tile (K14).
//
// Replaces tiled_rasterize (src/shaders/tiled-rasterizer.wgsl:82-273): a fixed 32 x 256 batch loop with three
// barriers per batch even when empty, a 48-byte AoS LDS record and no early-out.  Here:
//   * splats are staged once per 256-entry batch as three float4 LDS planes (centre+extent | conic+opacity | colour);
//   * each wave owns an 8x8 pixel block and first compacts the batch to the splats whose extent box overlaps that
//     block (one lane per splat, ballot + prefix popcount, order preserving), so its inner loop only visits splats that
//     at least one of its pixels accepts -- the per-pixel extent test of the reference is kept and is what decides;
//   * the inner loop walks the compacted list through v_readlane (index in an SGPR, LDS planes read by broadcast);
//   * the walk ends when the tile's entries end or when every pixel of the tile is saturated (A > 0.99), which
//     cannot change any output: after saturation the reference's loop `continue`s without touching C, A or
//     last_contributor (lines 224-226).
// Bound: fp32 VALU issue (about 40 lane-ops per accepted pixel-splat pair incl. the deterministic exp), not HBM: the
// tile's splat list is read once (4 B key + 4 B index + 24 B Splat per entry) and 12 B/pixel are written.
// Arithmetic is the pinned contraction of DESIGN.md "raster math", bit-identical to the parity oracle.
#include "common.h"
#include "dmath.h"

namespace {

constexpr u32 BATCH = 256;

template <bool GAUSSIAN_MODE>
__global__ __launch_bounds__(256) void rasterize_kernel(RenderSettings settings, TileInfo ti, const u32* __restrict__ splats, u32 num_splats,
                                                         const u32* __restrict__ ranges, const u32* __restrict__ sorted_keys,
                                                         const u32* __restrict__ sorted_vals, const u32* __restrict__ count_ptr, u32 max_batches,
                                                         u32* __restrict__ out_rgba8, float* __restrict__ out_alpha, u32* __restrict__ out_ncontrib) {
    __shared__ float4 s_geo[BATCH];  // centre.x, centre.y, extent.x, extent.y   (pixels)
    __shared__ float4 s_con[BATCH];  // conic.x, 2*conic.y, conic.z, opacity
    __shared__ float4 s_col[BATCH];  // r, g, b, -
    __shared__ unsigned char s_list[4][BATCH];  // per wave: staged indices that overlap the wave's 8x8 block, in order

    const u32 tile_id = blockIdx.x;
    const u32 tile_x = tile_id % ti.num_tiles_x, tile_y = tile_id / ti.num_tiles_x;
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u32 bx = tile_x * 16u + (wave & 1u) * 8u, by = tile_y * 16u + (wave >> 1) * 8u;  // block origin
    const u32 pixel_x = bx + (lane & 7u), pixel_y = by + (lane >> 3);
    const float vx = settings.viewport_x, vy = settings.viewport_y;
    const u32 W = wd_to_u32(vx), H = wd_to_u32(vy);
    const bool in_bounds = pixel_x < W && pixel_y < H;
    const float px = (float)pixel_x + 0.5f, py = (float)pixel_y + 0.5f;
    const float blk_x0 = (float)bx + 0.5f, blk_x1 = (float)bx + 7.5f, blk_y0 = (float)by + 0.5f, blk_y1 = (float)by + 7.5f;
    const float cap = (settings.max_splat_radius_px > 0.0f) ? settings.max_splat_radius_px : 1e9f;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    const u32 total = *count_ptr;
    const u32 start = ranges[tile_id];
    const bool has_data = start < total;  // 0xFFFFFFFF (empty tile) fails this too

    float cr = 0.0f, cg = 0.0f, cb = 0.0f, A = 0.0f;
    u32 last_contributor = 0u;

    if (has_data) {
        for (u32 batch = 0; max_batches == 0u || batch < max_batches; batch++) {
            const u32 entry = start + batch * BATCH + threadIdx.x;
            bool valid = false;
            if (entry < total) {
                const u32 key = sorted_keys[entry];
                if ((key >> 16u) == tile_id + 1u) {
                    const u32 g = sorted_vals[entry];
                    if (g < num_splats) {
                        const uint2* sp = reinterpret_cast<const uint2*>(splats + (size_t)g * 6);
                        const uint2 w01 = sp[0], w23 = sp[1], w45 = sp[2];
                        const float cx = (wd_unpack_lo(w01.x) * 0.5f + 0.5f) * vx;
                        const float cy = (wd_unpack_hi(w01.x) * -0.5f + 0.5f) * vy;
                        s_geo[threadIdx.x] = make_float4(cx, cy, fminf(wd_unpack_lo(w01.y), cap), fminf(wd_unpack_hi(w01.y), cap));
                        s_con[threadIdx.x] = make_float4(wd_unpack_lo(w23.x), 2.0f * wd_unpack_hi(w23.x), wd_unpack_lo(w23.y), wd_unpack_hi(w45.y));
                        s_col[threadIdx.x] = make_float4(wd_unpack_lo(w45.x), wd_unpack_hi(w45.x), wd_unpack_lo(w45.y), 0.0f);
                        valid = true;
                    }
                }
            }
            // Entries of a tile are contiguous, so the valid lanes are a prefix of the batch.
            const u32 n_valid = (u32)__syncthreads_count(valid);
            if (n_valid == 0u) break;

            const bool lane_live = in_bounds && (!GAUSSIAN_MODE || !(A > 0.99f));
            if (__any(lane_live)) {
                // --- compact the batch to this wave's block (conservative and exact per axis: a splat is dropped only if
                //     the nearest block pixel already fails the per-pixel test |p - c| > extent, which is monotone in p)
                u32 cnt = 0;
                for (u32 r = 0; r * 64u < n_valid; r++) {
                    const u32 j = r * 64u + lane;
                    bool ok = false;
                    if (j < n_valid) {
                        const float4 geo = s_geo[j];
                        ok = !((blk_x0 - geo.x) > geo.z || (geo.x - blk_x1) > geo.z || (blk_y0 - geo.y) > geo.w || (geo.y - blk_y1) > geo.w);
                    }
                    const unsigned long long m = __ballot(ok);
                    if (ok) s_list[wave][cnt + (u32)__popcll(m & lt_mask)] = (unsigned char)j;
                    cnt += (u32)__popcll(m);
                }
                const u32 processed_base = batch * BATCH;
                for (u32 c0 = 0; c0 < cnt; c0 += 64u) {
                    // wave-private LDS list: this read is ordered after the writes above within the wave
                    const u32 mine = (c0 + lane < cnt) ? (u32)s_list[wave][c0 + lane] : 0u;
                    const u32 chunk = min(64u, cnt - c0);
                    for (u32 kk = 0; kk < chunk; kk++) {
                        const u32 i = (u32)__builtin_amdgcn_readlane((int)mine, (int)kk);
                        const float4 geo = s_geo[i];
                        const float dx = px - geo.x, dy = py - geo.y;
                        const bool inside = in_bounds && !(fabsf(dx) > geo.z || fabsf(dy) > geo.w);
                        if (GAUSSIAN_MODE) {
                            const bool active = inside && !(A > 0.99f);
                            if (!__any(active)) continue;
                            if (active) {
                                const float4 con = s_con[i];
                                const float4 col = s_col[i];
                                const float t1 = __builtin_fmaf(con.x, dx, con.y * dy);
                                const float q = __builtin_fmaf(t1, dx, (con.z * dy) * dy);
                                const float G = wd_exp(-0.5f * q);
                                // G*opacity >= +0 and never NaN here (opacity in (1/128, 1], G in [0, inf]), so the hardware
                                // min/max equal WGSL's select-based clamp.
                                const float alpha = fminf(fmaxf(G * con.w, 0.0f), 0.99f);
                                const float w = alpha * (1.0f - A);
                                cr = __builtin_fmaf(col.x, w, cr);
                                cg = __builtin_fmaf(col.y, w, cg);
                                cb = __builtin_fmaf(col.z, w, cb);
                                A = A + w;
                                if (alpha >= (1.0f / 255.0f)) last_contributor = processed_base + i + 1u;
                            }
                        } else {
                            // point-cloud preview (tiled-rasterizer.wgsl:212-222): paints yellow discs, no saturation test
                            if (inside) {
                                const float dist_sq = dx * dx + dy * dy;
                                const float limit = fminf(settings.point_size_px, cap);
                                if (dist_sq <= limit * limit) {
                                    cr = 1.0f; cg = 1.0f; cb = 0.0f; A = 1.0f;
                                    last_contributor = processed_base + i + 1u;
                                }
                            }
                        }
                    }
                }
            }
            // every pixel of the tile saturated -> nothing later can change an output
            const bool done = !in_bounds || (GAUSSIAN_MODE && A > 0.99f);
            if (__syncthreads_and(done)) break;
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
}

}  // namespace

int launch_rasterize(wdgs_device* dev, const RenderSettings& st, const TileInfo& ti, const void* splats, u32 num_splats, const void* ranges,
                     const void* sorted_keys, const void* sorted_vals, const void* count_ptr, u32 max_batches, void* out_rgba8, void* out_alpha,
                     void* out_ncontrib) {
    if (ti.total_tiles == 0) return WDGS_OK;
    if (st.gaussian_mode >= 0.5f) {
        WDGS_LAUNCH(dev, "rasterize", rasterize_kernel<true>, dim3(ti.total_tiles), dim3(256), 0, st, ti, (const u32*)splats, num_splats, (const u32*)ranges,
                    (const u32*)sorted_keys, (const u32*)sorted_vals, (const u32*)count_ptr, max_batches, (u32*)out_rgba8, (float*)out_alpha, (u32*)out_ncontrib);
    } else {
        WDGS_LAUNCH(dev, "rasterize_points", rasterize_kernel<false>, dim3(ti.total_tiles), dim3(256), 0, st, ti, (const u32*)splats, num_splats,
                    (const u32*)ranges, (const u32*)sorted_keys, (const u32*)sorted_vals, (const u32*)count_ptr, max_batches, (u32*)out_rgba8,
                    (float*)out_alpha, (u32*)out_ncontrib);
    }
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}
