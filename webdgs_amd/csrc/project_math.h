// K1's per-Gaussian arithmetic (projection, conic, extents, SH colour, tile box) and its workgroup epilogue, shared by project_count
// (project.hip) and by the kernel that projects a Gaussian for the NEXT view right after updating it (backward.hip:
// geometry_backward_adam_project).  Replaces count_main of the reference (src/shaders/tiled-forward.wgsl:161-294, helpers in
// src/shaders/common.wgsl:44-108).
#pragma once
#include "common.h"
#include "wgslm.h"

namespace {

__constant__ float SH_C2c[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
__constant__ float SH_C3c[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// The SH row of one Gaussian in registers: 48 fp16 in [k][rgb] order = 24 words, fetched as 16-byte loads at the TOP of the kernel
// together with the Gaussian itself (only the words the degree needs).  Fetching it where the colour is evaluated -- after the
// culling tests -- put a second, dependent HBM round trip on every wave's critical path; the kernel is latency-bound (18 us per wave
// for ~1000 instructions: profiles/r02a_pmc.json), so the 7 % of rows fetched for Gaussians that are then culled are well spent.
struct ShRow { u32 w[24]; };
WD_DEV ShRow load_sh_row(const u32* __restrict__ sh_buffer, u32 idx, u32 sh_deg) {
    ShRow r;
#pragma unroll
    for (u32 i = 0; i < 24u; i++) r.w[i] = 0u;
    const uint4* q = reinterpret_cast<const uint4*>(sh_buffer + (size_t)idx * 24);
    const u32 nq = (sh_deg == 0u) ? 1u : (sh_deg == 1u) ? 2u : (sh_deg == 2u) ? 4u : 6u;  // ceil(6 (deg+1)^2 / 16) 16-byte words
#pragma unroll
    for (u32 i = 0; i < 6u; i++)
        if (i < nq) { const uint4 v = q[i]; r.w[4 * i] = v.x; r.w[4 * i + 1] = v.y; r.w[4 * i + 2] = v.z; r.w[4 * i + 3] = v.w; }
    return r;
}
WD_DEV float sh_half(const ShRow& sh, u32 h) {  // element h of the 48; h is a compile-time constant at every call site
    const u32 w = sh.w[h >> 1];
    return (h & 1u) ? wd_unpack_hi(w) : wd_unpack_lo(w);
}
WD_DEV vec3 sh_coef(const ShRow& sh, u32 c_idx) { return V3(sh_half(sh, c_idx * 3u), sh_half(sh, c_idx * 3u + 1u), sh_half(sh, c_idx * 3u + 2u)); }

WD_DEV vec3 color_from_sh(const ShRow& sh, vec3 dir, u32 sh_deg) {
    const float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
    vec3 result = SH_C0 * sh_coef(sh, 0u);
    if (sh_deg > 0u) {
        const float x = dir.x, y = dir.y, z = dir.z;
        result = result + (-SH_C1 * y * sh_coef(sh, 1u) + SH_C1 * z * sh_coef(sh, 2u) - SH_C1 * x * sh_coef(sh, 3u));
        if (sh_deg > 1u) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            result = result + (SH_C2c[0] * xy * sh_coef(sh, 4u) + SH_C2c[1] * yz * sh_coef(sh, 5u) +
                               SH_C2c[2] * (2.0f * zz - xx - yy) * sh_coef(sh, 6u) + SH_C2c[3] * xz * sh_coef(sh, 7u) +
                               SH_C2c[4] * (xx - yy) * sh_coef(sh, 8u));
            if (sh_deg > 2u) {
                result = result + (SH_C3c[0] * y * (3.0f * xx - yy) * sh_coef(sh, 9u) + SH_C3c[1] * xy * z * sh_coef(sh, 10u) +
                                   SH_C3c[2] * y * (4.0f * zz - xx - yy) * sh_coef(sh, 11u) +
                                   SH_C3c[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh_coef(sh, 12u) +
                                   SH_C3c[4] * x * (4.0f * zz - xx - yy) * sh_coef(sh, 13u) +
                                   SH_C3c[5] * z * (xx - yy) * sh_coef(sh, 14u) + SH_C3c[6] * x * (xx - 3.0f * yy) * sh_coef(sh, 15u));
            }
        }
    }
    result = result + 0.5f;
    return vmax(V3(0.0f), result);
}

WD_DEV vec3 covariance2D(const Cov3D& c3, vec4 mean_view, vec2 focal, vec2 viewport, const mat4& vm) {
    vec3 t = xyz(mean_view);
    const float fovx = wd_div(viewport.x * 0.5f, focal.x), fovy = wd_div(viewport.y * 0.5f, focal.y);
    const float limx = 1.3f * fovx, limy = 1.3f * fovy;
    const float txtz = wd_div(t.x, t.z), tytz = wd_div(t.y, t.z);
    t.x = wd_min(limx, wd_max(-limx, txtz)) * t.z;
    t.y = wd_min(limy, wd_max(-limy, tytz)) * t.z;
    const mat3 J = M3(V3(wd_div(focal.x, t.z), 0.0f, wd_div(-(focal.x * t.x), t.z * t.z)),
                      V3(0.0f, wd_div(focal.y, t.z), wd_div(-(focal.y * t.y), t.z * t.z)), V3(0.0f, 0.0f, 0.0f));
    const mat3 W = M3(V3(vm.c[0].x, vm.c[1].x, vm.c[2].x), V3(vm.c[0].y, vm.c[1].y, vm.c[2].y), V3(vm.c[0].z, vm.c[1].z, vm.c[2].z));
    const mat3 T = W * J;
    const mat3 Vrk = M3(V3(c3.v[0], c3.v[1], c3.v[2]), V3(c3.v[1], c3.v[3], c3.v[4]), V3(c3.v[2], c3.v[4], c3.v[5]));
    const mat3 cov = transpose(T) * transpose(Vrk) * T;
    return V3(cov.c[0].x + 0.3f, cov.c[0].y, cov.c[1].y + 0.3f);
}

WD_DEV u32 ordered_uint(float x) {
    const u32 bits = wd_f2bits(x);
    return bits ^ ((bits & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u);
}

// Tile bounding box of a stored (fp16) splat: shared by count and emit so both see the same integers.
struct TileBox { u32 min_x, min_y, max_x, max_y; bool valid; };
WD_DEV TileBox tile_box(vec2 ndc_f16, vec2 extents_f16, vec2 viewport, u32 ntx, u32 nty, bool check_empty) {
    TileBox b; b.valid = false; b.min_x = b.min_y = b.max_x = b.max_y = 0u;
    const vec2 pixel_center = (ndc_f16 * V2(0.5f, -0.5f) + 0.5f) * viewport;
    const vec2 lo = pixel_center - extents_f16 - 2.0f;
    const vec2 hi = pixel_center + extents_f16 + 2.0f;
    if (hi.x < 0.0f || hi.y < 0.0f || lo.x >= viewport.x || lo.y >= viewport.y) return b;
    const float bminx = wd_max(lo.x, 0.0f), bminy = wd_max(lo.y, 0.0f);
    const float bmaxx = wd_min(hi.x, viewport.x - 1.0f), bmaxy = wd_min(hi.y, viewport.y - 1.0f);
    if (check_empty && (bmaxx < bminx || bmaxy < bminy)) return b;
    b.min_x = wd_to_u32(bminx) / 16u;
    b.min_y = wd_to_u32(bminy) / 16u;
    b.max_x = min(wd_to_u32(bmaxx) / 16u, ntx - 1u);
    b.max_y = min(wd_to_u32(bmaxy) / 16u, nty - 1u);
    b.valid = true;
    return b;
}

// K1 for one Gaussian under one camera (tiled-forward.wgsl:161-294): false = culled (nothing is written); true = visible: Splat and depth
// are written, the tile count and the box come back.  Shared by the per-view kernel and the view-batched one, so both evaluate the same
// operations in the same order.
WD_DEV bool project_one(u32 idx, const uint2 w01, const uint2 w23, const uint2 w45, const ShRow& sh_row, const float* __restrict__ camera_f,
                        const RenderSettings& settings, const TileInfo& ti, u32* __restrict__ splats, u32* __restrict__ depths, u32& num_tiles_out, u32& box_x0,
                        u32& box_x1, u32& box_rows) {
    const vec4 quaternion = V4(wd_unpack_lo(w23.x), wd_unpack_hi(w23.x), wd_unpack_lo(w23.y), wd_unpack_hi(w23.y));
    const vec3 gaussian_scale = vexp(V3(wd_unpack_lo(w45.x), wd_unpack_hi(w45.x), wd_unpack_lo(w45.y)));
    const vec3 pos = V3(wd_unpack_lo(w01.x), wd_unpack_hi(w01.x), wd_unpack_lo(w01.y));
    const float opacity_raw = wd_unpack_hi(w01.y);
    const float opacity_sigmoid = wd_div(1.0f, 1.0f + wd_exp(-opacity_raw));

    const CameraUniforms& cam = *reinterpret_cast<const CameraUniforms*>(camera_f);
    const mat4 view = cam.view;
    const vec4 world_to_view = view * V4(pos, 1.0f);
    const vec4 clip = cam.proj * world_to_view;
    if (clip.w == 0.0f) return false;
    const vec3 ndc = xyz(clip) / clip.w;
    if (ndc.x < -1.2f || ndc.x > 1.2f || ndc.y < -1.2f || ndc.y > 1.2f || ndc.z < 0.0f || ndc.z > 1.0f) return false;

    const Cov3D c3 = covariance3D(quaternion, gaussian_scale);
    const vec2 viewport = V2(settings.viewport_x, settings.viewport_y);
    const vec3 c2 = covariance2D(c3, world_to_view, cam.focal, viewport, view);
    const float det = (c2.x * c2.z) - (c2.y * c2.y);
    if (det <= 0.0f) return false;
    const float det_inv = wd_div(1.0f, det);
    const vec3 conic = V3(c2.z * det_inv, -c2.y * det_inv, c2.x * det_inv);
    const float disc = conic.y * conic.y - conic.x * conic.z;
    if (conic.x <= 0.0f || conic.z <= 0.0f || disc >= 0.0f) return false;

    const float t = 2.0f * wd_log(opacity_sigmoid * 128.0f);
    if (t <= 0.0f) return false;
    const float x_extent = wd_sqrt(wd_div(t * conic.z, -disc));
    const float y_extent = wd_sqrt(wd_div(t * conic.x, -disc));
    const float cap = (settings.max_splat_radius_px > 0.0f) ? settings.max_splat_radius_px : 1e9f;
    const float xec = wd_min(x_extent, cap), yec = wd_min(y_extent, cap);
    // Round-trip through fp16 so emit/raster/backward (which only see the Splat) agree on the bbox.
    const u32 ndc_packed = wd_pack2(wd_clamp(ndc.x, -60000.0f, 60000.0f), wd_clamp(ndc.y, -60000.0f, 60000.0f));
    const u32 ext_packed = wd_pack2(xec, yec);
    const vec2 ndc_store = V2(wd_unpack_lo(ndc_packed), wd_unpack_hi(ndc_packed));
    const vec2 ext_f16 = V2(wd_unpack_lo(ext_packed), wd_unpack_hi(ext_packed));
    const TileBox tb = tile_box(ndc_store, ext_f16, viewport, ti.num_tiles_x, ti.num_tiles_y, true);
    if (!tb.valid) return false;

    const vec3 cam_pos = xyz(cam.view_inv.c[3]);
    const vec3 dir = normalize(pos - cam_pos);
    const vec3 color = color_from_sh(sh_row, dir, wd_to_u32(settings.sh_deg));

    const u32 num_tiles = (tb.max_x - tb.min_x + 1u) * (tb.max_y - tb.min_y + 1u);
    if (num_tiles > 2048u) return false;

    u32* s = splats + (size_t)idx * 6;
    uint2 o01, o23, o45;
    o01.x = ndc_packed;
    o01.y = ext_packed;
    o23.x = wd_pack2(conic.x, conic.y);
    o23.y = wd_pack2(conic.z, 0.0f);
    o45.x = wd_pack2(wd_clamp(color.x, 0.0f, 1.0f), wd_clamp(color.y, 0.0f, 1.0f));
    o45.y = wd_pack2(wd_clamp(color.z, 0.0f, 1.0f), wd_clamp(opacity_sigmoid, 0.0f, 1.0f));
    *reinterpret_cast<uint2*>(s) = o01;
    *reinterpret_cast<uint2*>(s + 2) = o23;
    *reinterpret_cast<uint2*>(s + 4) = o45;
    depths[idx] = ordered_uint(world_to_view.z);
    num_tiles_out = num_tiles;
    box_x0 = tb.min_x; box_x1 = tb.max_x; box_rows = tb.max_y - tb.min_y + 1u;
    return true;
}

// The Gaussian's six words and its SH row (with the optimizer's deferred DC halves, adam.h), fetched together at the top
WD_DEV void load_gaussian_and_sh(u32 idx, const u32* __restrict__ gaussians, const u32* __restrict__ sh_buffer, const u32* __restrict__ dc_words, u32 sh_deg,
                                 uint2& w01, uint2& w23, uint2& w45, ShRow& sh_row) {
    const u32* g = gaussians + (size_t)idx * 6;
    w01 = *reinterpret_cast<const uint2*>(g);
    w23 = *reinterpret_cast<const uint2*>(g + 2);
    w45 = *reinterpret_cast<const uint2*>(g + 4);
    sh_row = load_sh_row(sh_buffer, idx, sh_deg);
    if (dc_words) {  // the optimizer defers its writes of the row's first six bytes: the current values are here
        const uint2 dcw = *reinterpret_cast<const uint2*>(dc_words + (size_t)idx * 2);
        sh_row.w[0] = dcw.x;
        sh_row.w[1] = (sh_row.w[1] & 0xFFFF0000u) | (dcw.y & 0xFFFFu);
    }
}

// The end of K1 for a workgroup of 256 consecutive Gaussians (every thread of the workgroup calls it, in uniform control flow):
//  * visible_gaussians: the reference does one atomicAdd per visible splat on ONE word (tiled-forward.wgsl:292).  Even one
//    atomic per wave on a single address serialises the kernel (~12 ns each, measured: 15.6 K waves = the whole 0.2 ms),
//    so the count goes to 64 shard words (one add per workgroup); update_stats folds the shards into stats[1].
//  * block_counts[b] = tile entries of this workgroup's 256 Gaussians: the first level of the offsets scan, produced where the counts
//    are (the scan of these ~N/256 sums and the emit kernel's own in-workgroup prefix replace a reduce and a down-sweep launch).
//  * column_counts (nullable): tile entries of this workgroup's Gaussians per tile COLUMN -- the digit counts of the first pass of the
//    tile sort, produced here where the boxes are, so that emit can write its entries straight into column order (emit_scatter).
//    s_col[256] must have been zeroed (and a barrier passed) before the first add.
WD_DEV void project_block_epilogue(bool visible, u32 num_tiles_out, u32 box_x0, u32 box_x1, u32 box_rows, u32 num_tiles_x, u32* s_col /*[256]*/, u32* s_vis /*[4]*/,
                                   u32* s_cnt /*[4]*/, u32* __restrict__ visible_shards, u32* __restrict__ block_counts, u32* __restrict__ column_counts) {
    if (column_counts)
        for (u32 x = box_x0; x <= box_x1; x++) atomicAdd(&s_col[x], box_rows);  // (an invisible Gaussian has an empty range)
    const unsigned long long mask = __ballot(visible);
    u32 wsum = num_tiles_out;
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) wsum += (u32)__shfl_xor((int)wsum, (int)d, 64);
    if ((threadIdx.x & 63u) == 0u) { s_vis[threadIdx.x >> 6] = (u32)__popcll(mask); s_cnt[threadIdx.x >> 6] = wsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 c = s_vis[0] + s_vis[1] + s_vis[2] + s_vis[3];
        if (c) atomicAdd(&visible_shards[blockIdx.x & 63u], c);
        if (block_counts) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
    if (column_counts && threadIdx.x < num_tiles_x) column_counts[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s_col[threadIdx.x];
}

}  // namespace
