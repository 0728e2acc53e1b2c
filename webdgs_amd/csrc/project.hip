// Per-splat projection + tile counting (K1) and tile-entry emission (K6).
//
// Replaces count_main / emit_main of the reference (src/shaders/tiled-forward.wgsl:161-294, 297-354, helpers in
// src/shaders/common.wgsl:44-108).  Both are N-wide streaming kernels bound by HBM:
//   project: reads 24 B Gaussian + 6K B SH (K = (deg+1)^2, visible only), writes 24 B Splat + 4 B depth (visible)
//            and 4 B tile count (all)           -> N(24+4) + V(6K+28) bytes
//   emit:    reads 24 B Splat + 4 B depth + 8 B count/offset, writes 8 B per tile entry -> 16V + 8N + 8E bytes
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
    // (a NaN depth counts as the canonical quiet NaN: which NaN an operation returns -- sign, payload -- differs between machines, and these bits are a sort key)
    const u32 bits = (x != x) ? 0x7FC00000u : wd_f2bits(x);
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
// nf_stamp (nullable) / stamp: a Splat with a NaN or an infinity among its fp16 fields stamps the tiles of its box, so that the compositing kernel
// takes those tiles in the oracle's own forms (raster.hip: EXACT) -- the fast forms assume ordinary operands.
WD_DEV bool has_nonfinite_half(u32 a, u32 b, u32 c, u32 d, u32 e, u32 f) {
    // a half is non-finite when its five exponent bits are all ones: (h & 0x7C00) + 0x0400 reaches bit 15 then, and only then (no carry into the other half)
    const u32 M = 0x7C007C00u, C = 0x04000400u;
    return ((((a & M) + C) | ((b & M) + C) | ((c & M) + C) | ((d & M) + C) | ((e & M) + C) | ((f & M) + C)) & 0x80008000u) != 0u;
}
WD_DEV bool project_one(u32 idx, const uint2 w01, const uint2 w23, const uint2 w45, const ShRow& sh_row, const float* __restrict__ camera_f,
                        const RenderSettings& settings, const TileInfo& ti, u32* __restrict__ splats, u32* __restrict__ depths, u32& num_tiles_out, u32& box_x0,
                        u32& box_x1, u32& box_rows, u32* __restrict__ nf_stamp, u32 stamp) {
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
    if (nf_stamp && has_nonfinite_half(o01.x, o01.y, o23.x, o23.y, o45.x, o45.y))   // (rare; every writer of a frame stores the same number)
        for (u32 ty = tb.min_y; ty <= tb.max_y; ty++)
            for (u32 tx = tb.min_x; tx <= tb.max_x; tx++) nf_stamp[ty * ti.num_tiles_x + tx] = stamp;
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

__global__ __launch_bounds__(256) void project_count_kernel(u32 n, const u32* __restrict__ gaussians, const u32* __restrict__ sh_buffer,
                                                             const float* __restrict__ camera_f, RenderSettings settings, TileInfo ti,
                                                             u32* __restrict__ splats, u32* __restrict__ depths,
                                                             u32* __restrict__ tile_counts, u32* __restrict__ visible_shards,
                                                             u32* __restrict__ block_counts, u32* __restrict__ column_counts /*[num_tiles_x][gridDim.x]*/,
                                                             const u32* __restrict__ dc_words /*nullable: u32[N][2], the trained SH-DC halves (adam.h)*/,
                                                             u32* __restrict__ nf_stamp /*nullable: u32[tiles]*/, const u32* __restrict__ nf_frame) {
    WD_STREAM_PRIO();
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 stamp = nf_stamp ? *nf_frame + 1u : 0u;   // the number the scan kernel gives this frame (scan.hip: stats_epilogue)
    bool visible = false;
    u32 num_tiles_out = 0u;
    // column_counts (nullable): tile entries of this workgroup's Gaussians per tile COLUMN -- the digit counts of the first pass of the
    // tile sort, produced here where the boxes are, so that emit can write its entries straight into column order (emit_scatter below).
    __shared__ u32 s_col[256];
    u32 box_x0 = 1u, box_x1 = 0u, box_rows = 0u;
    if (column_counts) {
        s_col[threadIdx.x] = 0u;
        __syncthreads();
    }
    if (idx < n) {
        uint2 w01, w23, w45;
        ShRow sh_row;
        load_gaussian_and_sh(idx, gaussians, sh_buffer, dc_words, wd_to_u32(settings.sh_deg), w01, w23, w45, sh_row);
        visible = project_one(idx, w01, w23, w45, sh_row, camera_f, settings, ti, splats, depths, num_tiles_out, box_x0, box_x1, box_rows, nf_stamp, stamp);
        tile_counts[idx] = num_tiles_out;
    }
    // visible_gaussians: the reference does one atomicAdd per visible splat on ONE word (tiled-forward.wgsl:292).  Even one
    // atomic per wave on a single address serialises the kernel (~12 ns each, measured: 15.6 K waves = the whole 0.2 ms),
    // so the count goes to 64 shard words (one add per workgroup); update_stats folds the shards into stats[1].
    // block_counts[b] = tile entries of this workgroup's 256 Gaussians: the first level of the offsets scan, produced where the counts
    // are (the scan of these ~N/256 sums and the emit kernel's own in-workgroup prefix replace a reduce and a down-sweep launch).
    if (column_counts)
        for (u32 x = box_x0; x <= box_x1; x++) atomicAdd(&s_col[x], box_rows);  // (an invisible Gaussian has an empty range)
    __shared__ u32 s_vis[4], s_cnt[4];
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
    if (column_counts && threadIdx.x < ti.num_tiles_x) column_counts[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s_col[threadIdx.x];
}

// K1 for ALL the views of a batched step (wdgs_tiled_forward_project_views): the thread fetches its Gaussian and SH row once and projects
// it under each camera in turn into that view's own buffers.  Per view the epilogue of project_count: tile count, visible-count shard,
// the workgroup's entry count and its per-column counts.
struct ProjectViews {
    u32 count;
    const float* camera[WDGS_MAX_BATCH_VIEWS];
    u32* splats[WDGS_MAX_BATCH_VIEWS];
    u32* depths[WDGS_MAX_BATCH_VIEWS];
    u32* tile_counts[WDGS_MAX_BATCH_VIEWS];
    u32* visible_shards[WDGS_MAX_BATCH_VIEWS];
    u32* block_counts[WDGS_MAX_BATCH_VIEWS];
    u32* column_counts[WDGS_MAX_BATCH_VIEWS];   // all null or none null
    u32* nf_stamp[WDGS_MAX_BATCH_VIEWS];        // (nullable) tiles of non-finite Splats, and each pass's frame number
    const u32* nf_frame[WDGS_MAX_BATCH_VIEWS];
};
// resident waves per SIMD the register allocation of the view-batched K1 aims at (make K1V_WAVES=n: a build-time choice, for same-box comparisons)
#ifndef WDGS_K1V_WAVES
#define WDGS_K1V_WAVES 4
#endif
__global__ __launch_bounds__(256, WDGS_K1V_WAVES) void project_count_views_kernel(u32 n, const u32* __restrict__ gaussians, const u32* __restrict__ sh_buffer, RenderSettings settings,
                                                                   TileInfo ti, ProjectViews pv, const u32* __restrict__ dc_words) {
    WD_STREAM_PRIO();
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ u32 s_col[256];
    __shared__ u32 s_vis[4], s_cnt[4];
    const bool columns = pv.column_counts[0] != nullptr;
    uint2 w01 = make_uint2(0u, 0u), w23 = w01, w45 = w01;
    ShRow sh_row;
    if (idx < n) load_gaussian_and_sh(idx, gaussians, sh_buffer, dc_words, wd_to_u32(settings.sh_deg), w01, w23, w45, sh_row);
    for (u32 v = 0; v < pv.count; v++) {
        if (columns) s_col[threadIdx.x] = 0u;
        __syncthreads();   // (also: the previous view's s_vis / s_cnt / s_col have been read)
        bool visible = false;
        u32 num_tiles_out = 0u, box_x0 = 1u, box_x1 = 0u, box_rows = 0u;
        if (idx < n) {
            visible = project_one(idx, w01, w23, w45, sh_row, pv.camera[v], settings, ti, pv.splats[v], pv.depths[v], num_tiles_out, box_x0, box_x1, box_rows, pv.nf_stamp[v],
                                  pv.nf_stamp[v] ? *pv.nf_frame[v] + 1u : 0u);
            pv.tile_counts[v][idx] = num_tiles_out;
        }
        if (columns)
            for (u32 x = box_x0; x <= box_x1; x++) atomicAdd(&s_col[x], box_rows);
        const unsigned long long mask = __ballot(visible);
        u32 wsum = num_tiles_out;
#pragma unroll
        for (u32 d = 32; d >= 1; d >>= 1) wsum += (u32)__shfl_xor((int)wsum, (int)d, 64);
        if ((threadIdx.x & 63u) == 0u) { s_vis[threadIdx.x >> 6] = (u32)__popcll(mask); s_cnt[threadIdx.x >> 6] = wsum; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const u32 c = s_vis[0] + s_vis[1] + s_vis[2] + s_vis[3];
            if (c) atomicAdd(&pv.visible_shards[v][blockIdx.x & 63u], c);
            pv.block_counts[v][blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        }
        if (columns && threadIdx.x < ti.num_tiles_x) pv.column_counts[v][(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s_col[threadIdx.x];
    }
}

// stats[0] = total tile entries (update_stats, src/shaders/update-stats.wgsl:19-35); stats[2] = overflow flag.
__global__ void update_stats_kernel(u32 n, const u32* __restrict__ offsets, const u32* __restrict__ counts, u32 capacity, u32* __restrict__ stats,
                                    u32* __restrict__ visible_shards, u32* __restrict__ host_mirror) {
    WD_STREAM_PRIO();
    // 64 threads: fold the visible-count shards (and clear them for the next encode)
    u32 v = visible_shards[threadIdx.x];
    visible_shards[threadIdx.x] = 0u;
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) v += (u32)__shfl_xor((int)v, (int)d, 64);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        stats[1] = v;
        const u32 total = (n == 0u) ? 0u : offsets[n - 1] + counts[n - 1];
        stats[0] = min(total, capacity);  // consumers only ever touch [0, capacity)
        stats[2] = (total > capacity) ? total : 0u;
        if (host_mirror) {  // pinned host memory: the host's per-step overflow check and stats read need no copy
            host_mirror[0] = min(total, capacity);
            host_mirror[1] = v;
            if (total > capacity) host_mirror[2] = total;  // sticky until the host check clears it (scan.hip)
            host_mirror[3] = 0u;
        }
        // a wrapped scan (sum >= 2^32) also shows as an offset going backwards; tile counts are <= 2048 each, so
        // N * 2048 < 2^32 for N < 2^21; beyond that the forward pass sizes its capacity from a 64-bit bound (api).
    }
}

// K6 emit_main (tiled-forward.wgsl:297-354): every visible splat writes one (key, index) entry per covered tile at its scanned
// offset.  The reference (and a thread-per-splat port) runs a divergent double loop with stores 24 B apart across lanes.  Here a
// wave expands its 64 splats cooperatively: their ranges are adjacent in the output (offsets are an exclusive scan), so the wave
// owns one contiguous run of T entries; lane j of each 64-entry slice finds the owning splat by a 6-step search over the wave's
// count prefix (wave-private LDS) and derives its tile from the entry's rank inside that splat's box.  Stores are coalesced and
// every lane does the same work regardless of the box sizes.
__global__ __launch_bounds__(256) void emit_kernel(u32 n, const u32* __restrict__ splats, const u32* __restrict__ depths,
                                                    const u32* __restrict__ tile_counts, u32* __restrict__ tile_offsets,
                                                    const u32* __restrict__ block_offsets, RenderSettings settings, TileInfo ti,
                                                    u32* __restrict__ keys, u32* __restrict__ values, u32 capacity) {
    WD_STREAM_PRIO();
    __shared__ u32 s_pre[4][64], s_org[4][64], s_w[4][64], s_inv[4][64], s_dep[4][64];
    __shared__ u32 s_wtot[4];
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u32 cnt = 0u, org = 0u, width = 1u, depth16 = 0u;
    if (idx < n) {
        const u32 num_tiles = tile_counts[idx];
        if (num_tiles != 0u) {
            const uint2 w01 = *reinterpret_cast<const uint2*>(splats + (size_t)idx * 6);
            const vec2 ndc = V2(wd_unpack_lo(w01.x), wd_unpack_hi(w01.x));
            const float cap = (settings.max_splat_radius_px > 0.0f) ? settings.max_splat_radius_px : 1e9f;
            const vec2 ext = V2(wd_min(wd_unpack_lo(w01.y), cap), wd_min(wd_unpack_hi(w01.y), cap));
            const vec2 viewport = V2(settings.viewport_x, settings.viewport_y);
            const TileBox tb = tile_box(ndc, ext, viewport, ti.num_tiles_x, ti.num_tiles_y, false);
            if (tb.valid) {
                // the same box project_count counted (it keeps splats with 1..2048 tiles), row-major as the reference's double loop
                width = tb.max_x - tb.min_x + 1u;
                cnt = width * (tb.max_y - tb.min_y + 1u);
                org = tb.min_y * ti.num_tiles_x + tb.min_x;
                depth16 = depths[idx] >> 16u;
            }
        }
    }
    // exclusive prefix of the counts over the wave
    u32 inc = cnt;
#pragma unroll
    for (u32 d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    const u32 total = (u32)__shfl((int)inc, 63, 64);
    const u32 wave_first = idx - lane;  // Gaussian of lane 0
    u32 start0;
    if (block_offsets) {
        // offsets are not scanned per Gaussian beforehand: this workgroup's start comes from the scan of the per-workgroup sums
        // (project_count wrote them), the wave's start from the totals of the waves before it, the Gaussian's from the wave prefix
        // above -- and the per-Gaussian table the reference exposes (getTileOffsetsBuffer) is written here as a by-product.
        if (lane == 0u) s_wtot[wave] = total;
        __syncthreads();
        start0 = block_offsets[blockIdx.x];
#pragma unroll
        for (u32 w = 0; w < 4u; w++) start0 += (w < wave) ? s_wtot[w] : 0u;
        if (idx < n) tile_offsets[idx] = start0 + inc - cnt;
        if (total == 0u) return;
    } else {
        if (total == 0u) return;
        start0 = tile_offsets[wave_first];  // (< n, or total would be 0)
    }
    s_pre[wave][lane] = inc - cnt;
    s_org[wave][lane] = org;
    s_w[wave][lane] = width;
    s_inv[wave][lane] = 0xFFFFFFFFu / width + 1u;  // floor(k / width) == umulhi(k, inv) for k * width < 2^32 (k <= 2048 here)
    s_dep[wave][lane] = depth16;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u32* pre = s_pre[wave];
    for (u32 e = lane; e < total; e += 64u) {
        u32 o = 0u;  // last lane whose prefix is <= e: the splat that owns entry e
#pragma unroll
        for (u32 step = 32u; step >= 1u; step >>= 1) o += (pre[o + step] <= e) ? step : 0u;
        const u32 k = e - pre[o];
        const u32 w = s_w[wave][o];
        const u32 row = (w == 1u) ? k : __umulhi(k, s_inv[wave][o]);  // (the reciprocal of 1 does not fit 32 bits)
        const u32 tile_id = s_org[wave][o] + row * ti.num_tiles_x + (k - row * w);
        const u32 key_idx = start0 + e;
        if (key_idx < capacity) {
            keys[key_idx] = ((tile_id + 1u) << 16u) | s_dep[wave][o];
            values[key_idx] = wave_first + o;
        }
    }
}


// K6 emit fused with the FIRST stable pass of the tile sort.  The sort key's tile field is tile = ty * num_tiles_x + tx; the forward
// pass sorts it as a two-digit mixed-radix number (tx, then ty: sort.hip).  The digit counts of the first pass per workgroup are known
// before a single key exists -- project_count counted its Gaussians' entries per tile column, forward_scan turned them into each
// workgroup's offset inside each column -- so emit does not write its entries in emission order for a histogram and a scatter pass to
// read back (8E + 4E + 16E bytes): it ranks them by column in LDS and writes them where the first pass would have put them (8E bytes).
//
// Order.  A stable pass keeps entries of one column in emission order; what the later passes need of that order is only that the
// entries of one TILE stay in ascending Gaussian order (two entries of one Gaussian never share a tile).  Entries are ranked in the
// workgroup's emission order (Gaussian-major), workgroups own consecutive Gaussians and consecutive slices of every column: same order.
//
// A workgroup expands its 256 Gaussians' entries in chunks of at most ES_CHUNK.  Owner of entry e = the last Gaussian whose count
// prefix is <= e: every Gaussian marks the chunk position where its entries start, an inclusive max-scan over the positions (8 per
// thread + one block scan) resolves every entry's owner at once -- instead of a dependent 8-step search per entry.  Wave w then ranks a
// contiguous quarter of the chunk in rounds of 64 (ballot match on the column, per-wave column counters: the scheme of sort_scatter), the
// chunk is reordered in LDS into column runs and leaves with consecutive lanes writing consecutive addresses.  Workgroups are numbered
// so that neighbours -- which write neighbouring slices of every column run -- sit on the same XCD and meet in its L2.  The per-Gaussian
// offsets table the reference exposes (getTileOffsetsBuffer) is written as a by-product, as emit does.
constexpr u32 ES_CHUNK = 2048;
constexpr u32 ES_ROUNDS = ES_CHUNK / 256;
__global__ __launch_bounds__(256) void emit_scatter_kernel(u32 n, const u32* __restrict__ splats, const u32* __restrict__ depths,
                                                            const u32* __restrict__ tile_counts, u32* __restrict__ tile_offsets,
                                                            const u32* __restrict__ block_offsets, RenderSettings settings, TileInfo ti,
                                                            const u32* __restrict__ column_offsets /*[num_tiles_x][gridDim.x], scanned*/,
                                                            const u32* __restrict__ column_totals, u32 inv_ntx /*2^32 / num_tiles_x, rounded up*/,
                                                            u32* __restrict__ keys, u32* __restrict__ values, u32 capacity) {
    WD_STREAM_PRIO();
    __shared__ u32 s_pre[256];                     // exclusive prefix of the entry counts over the workgroup's Gaussians
    __shared__ u32 s_box[256];                     // min_x | min_y << 8 | width << 16  (all < 256 on this path)
    __shared__ u32 s_inv[256], s_dep[256];
    __shared__ u32 whist[4][256];                  // per-wave column counters, then per-wave column starts inside the chunk
    __shared__ u32 s_gbase[256];                   // where this workgroup's next entry of column c goes (global)
    __shared__ u32 s_delta[256];                   // global position = s_delta[column] + position in the chunk's column order
    __shared__ __align__(16) u32 s_keys[ES_CHUNK];
    __shared__ __align__(16) u32 s_vals[ES_CHUNK];
    __shared__ u32 s_wsum[4], s_tsum[4];
    // owner (Gaussian of the workgroup, + 1) of every entry of the chunk: lives in the first 4 KB of s_vals, which is written only after
    // the ranking has read the owners (26 KB per workgroup: six fit a CU, five at 30 KB -- 3907 workgroups are 2.5 rounds of 1536 instead of
    // 3.05 rounds of 1280)
    unsigned short* const s_own = reinterpret_cast<unsigned short*>(s_vals);
    const u32 wg = xcd_contiguous(blockIdx.x, gridDim.x);
    const u32 idx = wg * 256u + threadIdx.x;
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u32 ntx = ti.num_tiles_x;
    u32 cnt = 0u, ox = 0u, oy = 0u, width = 1u, depth16 = 0u;
    // everything this workgroup reads from memory is requested up front, in ONE round trip: the tile count, the Splat's first 8 bytes and
    // the depth of every Gaussian (also of the ~7 % that turn out to be invisible: the rows exist, their content is ignored), the
    // workgroup's offset and this thread's column total / offset.  Fetching Splat and depth behind the tests on the count was three
    // dependent HBM round trips of ~2 us each on a kernel whose waves live ~9 us (profiles/r03c_pmc.json).
    const u32 col = threadIdx.x;
    const bool in_range = idx < n;
    const u32 num_tiles = in_range ? tile_counts[idx] : 0u;
    const uint2 w01 = in_range ? *reinterpret_cast<const uint2*>(splats + (size_t)idx * 6) : make_uint2(0u, 0u);
    const u32 depth_all = in_range ? depths[idx] : 0u;
    const u32 tot_c = (col < ntx) ? column_totals[col] : 0u;
    const u32 off_c = (col < ntx) ? column_offsets[(size_t)col * gridDim.x + wg] : 0u;
    const u32 wg_offset = block_offsets[wg];
    if (num_tiles != 0u) {
        const vec2 ndc = V2(wd_unpack_lo(w01.x), wd_unpack_hi(w01.x));
        const float cap = (settings.max_splat_radius_px > 0.0f) ? settings.max_splat_radius_px : 1e9f;
        const vec2 ext = V2(wd_min(wd_unpack_lo(w01.y), cap), wd_min(wd_unpack_hi(w01.y), cap));
        const vec2 viewport = V2(settings.viewport_x, settings.viewport_y);
        const TileBox tb = tile_box(ndc, ext, viewport, ti.num_tiles_x, ti.num_tiles_y, false);
        if (tb.valid) {  // the same box project_count counted
            width = tb.max_x - tb.min_x + 1u;
            cnt = width * (tb.max_y - tb.min_y + 1u);
            ox = tb.min_x; oy = tb.min_y;
            depth16 = depth_all >> 16u;
        }
    }
    // exclusive prefix of the counts over the workgroup
    u32 inc = cnt;
#pragma unroll
    for (u32 d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63u) s_wsum[wave] = inc;
    // global start of column c for this workgroup: exclusive scan of the column totals (done by every workgroup: a dozen instructions,
    // instead of a one-workgroup kernel in between) + the workgroup's offset inside the column
    u32 tinc = tot_c;
#pragma unroll
    for (u32 d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_up(tinc, d, 64);
        if (lane >= d) tinc += t;
    }
    if (lane == 63u) s_tsum[wave] = tinc;
    __syncthreads();
    u32 woff = 0u, total = 0u, tbase = tinc - tot_c;
#pragma unroll
    for (u32 w = 0; w < 4u; w++) {
        const u32 v = s_wsum[w];
        woff += (w < wave) ? v : 0u;
        total += v;
        tbase += (w < wave) ? s_tsum[w] : 0u;
    }
    const u32 pre = woff + inc - cnt;
    if (idx < n) tile_offsets[idx] = wg_offset + pre;
    if (total == 0u) return;  // (uniform)
    s_pre[threadIdx.x] = pre;
    s_box[threadIdx.x] = ox | (oy << 8u) | (width << 16u);
    s_inv[threadIdx.x] = 0xFFFFFFFFu / width + 1u;  // floor(k / width) == umulhi(k, inv) for k * width < 2^32 (k <= 2048 here)
    s_dep[threadIdx.x] = depth16;
    s_gbase[col] = tbase + off_c;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const u32 wg_first = wg * 256u;
    auto column_of = [&](u32 key) { const u32 t = (key >> 16u) - 1u; return t - __umulhi(t, inv_ntx) * ntx; };
    for (u32 c0 = 0u; c0 < total; c0 += ES_CHUNK) {
        const u32 n_here = min(ES_CHUNK, total - c0);
        const u32 per_wave = ((n_here + 3u) / 4u + 63u) & ~63u;  // multiple of 64; 4 * per_wave >= n_here
        const u32 rounds = per_wave / 64u;
        // ---- owners: marks, then an inclusive max-scan over the chunk (thread t resolves entries [8t, 8t + 8))
#pragma unroll
        for (u32 w = 0; w < 4u; w++) whist[w][threadIdx.x] = 0u;
        reinterpret_cast<uint4*>(s_own)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);   // 8 x u16
        __syncthreads();
        if (cnt != 0u && pre + cnt > c0 && pre < c0 + n_here) s_own[max(pre, c0) - c0] = (unsigned short)(threadIdx.x + 1u);
        __syncthreads();
        {
            const uint4 q = reinterpret_cast<const uint4*>(s_own)[threadIdx.x];
            u32 o[8] = {q.x & 0xFFFFu, q.x >> 16, q.y & 0xFFFFu, q.y >> 16, q.z & 0xFFFFu, q.z >> 16, q.w & 0xFFFFu, q.w >> 16};
#pragma unroll
            for (u32 j = 1; j < 8u; j++) o[j] = max(o[j], o[j - 1u]);
            u32 minc = o[7];  // inclusive max-scan of the per-thread maxima over the workgroup
#pragma unroll
            for (u32 d = 1; d < 64; d <<= 1) {
                const u32 t = __shfl_up(minc, d, 64);
                if (lane >= d) minc = max(minc, t);
            }
            if (lane == 63u) s_wsum[wave] = minc;
            __syncthreads();
            u32 carry = __shfl_up(minc, 1, 64);
            if (lane == 0u) carry = 0u;
#pragma unroll
            for (u32 w = 0; w < 4u; w++) carry = max(carry, (w < wave) ? s_wsum[w] : 0u);
#pragma unroll
            for (u32 j = 0; j < 8u; j++) o[j] = max(o[j], carry);
            reinterpret_cast<uint4*>(s_own)[threadIdx.x] = make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
        }
        __syncthreads();
        // ---- rank by column: wave w owns entries [w * per_wave, (w + 1) * per_wave) of the chunk, 64 per round
        u32 kk[ES_ROUNDS], vv[ES_ROUNDS], rk[ES_ROUNDS];
#pragma unroll
        for (u32 j = 0; j < ES_ROUNDS; j++) {
            kk[j] = 0xFFFFFFFFu; vv[j] = 0u; rk[j] = 0u;
            if (j < rounds) {  // (uniform per workgroup)
                const u32 i = wave * per_wave + j * 64u + lane;
                const bool valid = i < n_here;
                const u32 o = valid ? (u32)s_own[i] - 1u : 0u;   // (every valid entry has an owner: the chunk starts inside or at a Gaussian)
                const u32 k = c0 + (valid ? i : 0u) - s_pre[o];
                const u32 box = s_box[o];
                const u32 w = box >> 16u;
                const u32 row = (w == 1u) ? k : __umulhi(k, s_inv[o]);  // (the reciprocal of 1 does not fit 32 bits)
                const u32 cx = (box & 0xFFu) + (k - row * w);
                const u32 tile_id = (((box >> 8u) & 0xFFu) + row) * ntx + cx;
                kk[j] = valid ? (((tile_id + 1u) << 16u) | s_dep[o]) : 0xFFFFFFFFu;
                vv[j] = wg_first + o;
                const u32 digit = valid ? cx : 0u;
                unsigned long long m = __ballot(valid);
#pragma unroll
                for (u32 b = 0; b < 8; b++) {
                    const bool bit = (digit >> b) & 1u;
                    const unsigned long long bal = __ballot(bit);
                    m &= bit ? bal : ~bal;
                }
                // m = valid lanes of this wave holding the same column
                const u32 pre_c = whist[wave][digit];
                const u32 below = (u32)__popcll(m & lt_mask);
                rk[j] = pre_c + below;
                if (valid && below == 0u) whist[wave][digit] = pre_c + (u32)__popcll(m);  // group leader bumps the wave counter
            }
        }
        __syncthreads();
        {   // column starts inside the chunk's column order (exclusive scan of the column counts), per-wave starts, global deltas
            u32 cnt_c = 0u;
#pragma unroll
            for (u32 w = 0; w < 4u; w++) cnt_c += whist[w][col];
            u32 cinc = cnt_c;
#pragma unroll
            for (u32 d = 1; d < 64; d <<= 1) {
                const u32 t = __shfl_up(cinc, d, 64);
                if (lane >= d) cinc += t;
            }
            if (lane == 63u) s_wsum[wave] = cinc;
            __syncthreads();
            u32 lstart = cinc - cnt_c;
#pragma unroll
            for (u32 w = 0; w < 4u; w++) lstart += (w < wave) ? s_wsum[w] : 0u;
            const u32 gb = s_gbase[col];
            s_delta[col] = gb - lstart;
            s_gbase[col] = gb + cnt_c;  // the next chunk's entries of this column follow
            u32 run = lstart;
#pragma unroll
            for (u32 w = 0; w < 4u; w++) {
                const u32 c = whist[w][col];
                whist[w][col] = run;
                run += c;
            }
        }
        __syncthreads();
#pragma unroll
        for (u32 j = 0; j < ES_ROUNDS; j++) {
            if (j < rounds) {
                const u32 i = wave * per_wave + j * 64u + lane;
                if (i < n_here) {
                    const u32 lpos = whist[wave][column_of(kk[j])] + rk[j];
                    s_keys[lpos] = kk[j];
                    s_vals[lpos] = vv[j];
                }
            }
        }
        __syncthreads();
        for (u32 e = threadIdx.x; e < n_here; e += 256u) {
            const u32 key = s_keys[e];
            const u32 pos = s_delta[column_of(key)] + e;
            if (pos < capacity) {  // (a truncated list is reported as WDGS_E_CAPACITY and never used: api.hip)
                keys[pos] = key;
                values[pos] = s_vals[e];
            }
        }
        __syncthreads();  // the chunk has left: its LDS is reused
    }
}

}  // namespace

int launch_project_count(wdgs_device* dev, u32 n, const void* gaussians, const void* sh, const void* camera, const RenderSettings& st,
                         const TileInfo& ti, void* splats, void* depths, void* counts, void* visible_shards, void* block_counts, void* column_counts,
                         const void* dc_words, void* nf_stamp, const void* nf_frame) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "project_count", project_count_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const u32*)gaussians, (const u32*)sh,
                (const float*)camera, st, ti, (u32*)splats, (u32*)depths, (u32*)counts, (u32*)visible_shards, (u32*)block_counts, (u32*)column_counts,
                (const u32*)dc_words, (u32*)nf_stamp, (const u32*)nf_frame);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_project_count_views(wdgs_device* dev, u32 n, u32 count, const void* gaussians, const void* sh, const void* const* cameras, const RenderSettings& st,
                               const TileInfo& ti, void* const* splats, void* const* depths, void* const* counts, void* const* visible_shards, void* const* block_counts,
                               void* const* column_counts, const void* dc_words, void* const* nf_stamp, const void* const* nf_frame) {
    if (n == 0 || count == 0) return WDGS_OK;
    ProjectViews pv{};
    pv.count = count;
    for (u32 v = 0; v < count; v++) {
        pv.camera[v] = (const float*)cameras[v]; pv.splats[v] = (u32*)splats[v]; pv.depths[v] = (u32*)depths[v]; pv.tile_counts[v] = (u32*)counts[v];
        pv.visible_shards[v] = (u32*)visible_shards[v]; pv.block_counts[v] = (u32*)block_counts[v]; pv.column_counts[v] = column_counts ? (u32*)column_counts[v] : nullptr;
        pv.nf_stamp[v] = nf_stamp ? (u32*)nf_stamp[v] : nullptr; pv.nf_frame[v] = nf_frame ? (const u32*)nf_frame[v] : nullptr;
    }
    WDGS_LAUNCH(dev, "project_count_views", project_count_views_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const u32*)gaussians, (const u32*)sh, st, ti, pv,
                (const u32*)dc_words);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_update_stats(wdgs_device* dev, u32 n, const void* offsets, const void* counts, u32 capacity, void* stats, void* visible_shards, void* host_mirror) {
    WDGS_LAUNCH(dev, "update_stats", update_stats_kernel, dim3(1), dim3(64), 0, n, (const u32*)offsets, (const u32*)counts, capacity, (u32*)stats,
                (u32*)visible_shards, (u32*)host_mirror);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_emit_scatter(wdgs_device* dev, u32 n, const void* splats, const void* depths, const void* counts, void* offsets, const void* block_offsets,
                        const RenderSettings& st, const TileInfo& ti, const void* column_offsets, const void* column_totals, void* keys, void* values, u32 capacity) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "emit_scatter", emit_scatter_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const u32*)splats, (const u32*)depths, (const u32*)counts,
                (u32*)offsets, (const u32*)block_offsets, st, ti, (const u32*)column_offsets, (const u32*)column_totals, 0xFFFFFFFFu / ti.num_tiles_x + 1u /*num_tiles_x >= 2 on this path*/,
                (u32*)keys, (u32*)values, capacity);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_emit(wdgs_device* dev, u32 n, const void* splats, const void* depths, const void* counts, void* offsets, const void* block_offsets,
                const RenderSettings& st, const TileInfo& ti, void* keys, void* values, u32 capacity) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "emit", emit_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const u32*)splats, (const u32*)depths, (const u32*)counts,
                (u32*)offsets, (const u32*)block_offsets, st, ti, (u32*)keys, (u32*)values, capacity);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}
