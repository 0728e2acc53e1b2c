// ORACLE (test infrastructure only; PARITY UNPINNED -- see wgsl_shim.hpp).
//
// CPU restatement of the reference's forward half, kernel by kernel.  File:line
// citations are relative to /root/reference/src.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library.
//
//   K1  count_main        shaders/tiled-forward.wgsl:161-294  + shaders/common.wgsl:44-108
//   K2-4 exclusive scan   prefix/prefix_sum.wgsl:88-220 (semantics: exclusive u32 scan)
//   K5  update_stats      shaders/update-stats.wgsl:19-35
//   K6  emit_main         shaders/tiled-forward.wgsl:297-354
//   K7-11 radix sort      sort/radix_sort.wgsl (semantics: stable ascending sort of (key,value) on the 32-bit key)
//   K12-13 tile ranges    shaders/tile-ranges.wgsl:46-104
//   K14 tiled_rasterize   shaders/tiled-rasterizer.wgsl:82-273
#include "wgsl_shim.hpp"

#include <algorithm>
#include <vector>

using namespace wgsl;

namespace {

struct CameraUniforms {  // common.wgsl:1-8 ; camera/camera.ts:165-195 (column-major mat4 x4, viewport, focal)
    mat4 view, view_inv, proj, proj_inv;
    vec2 viewport, focal;
};
static CameraUniforms load_camera(const f32* c) {
    CameraUniforms cam;
    std::memcpy(&cam, c, sizeof(f32) * 68);
    return cam;
}
struct RenderSettings {  // common.wgsl:10-18
    f32 gaussian_scaling, sh_deg, viewport_x, viewport_y, point_size_px, gaussian_mode, max_splat_radius_px;
};

struct Cov3D { f32 v[6]; };

// common.wgsl:44-68
static Cov3D covariance3D(vec4 quaternion, vec3 scale) {
    const f32 x = quaternion.y, y = quaternion.z, z = quaternion.w, r = quaternion.x;
    const mat3 R = M3(
        V3(1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y - r * z), 2.0f * (x * z + r * y)),
        V3(2.0f * (x * y + r * z), 1.0f - 2.0f * (x * x + z * z), 2.0f * (y * z - r * x)),
        V3(2.0f * (x * z - r * y), 2.0f * (y * z + r * x), 1.0f - 2.0f * (x * x + y * y)));
    const mat3 S = M3(V3(scale.x, 0.0f, 0.0f), V3(0.0f, scale.y, 0.0f), V3(0.0f, 0.0f, scale.z));
    const mat3 M = S * R;
    const mat3 cov_mat = transpose(M) * M;
    return Cov3D{{cov_mat[0][0], cov_mat[0][1], cov_mat[0][2], cov_mat[1][1], cov_mat[1][2], cov_mat[2][2]}};
}

// common.wgsl:71-108
static vec3 covariance2D(const Cov3D& cov_3D, vec4 mean_view, vec2 focal, vec2 viewport, const mat4& viewmatrix) {
    vec3 t = mean_view.xyz();
    const f32 focal_x = focal.x, focal_y = focal.y;
    const f32 fovx = viewport.x * 0.5f / focal_x;
    const f32 fovy = viewport.y * 0.5f / focal_y;
    const f32 limx = 1.3f * fovx, limy = 1.3f * fovy;
    const f32 txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = wmin(limx, wmax(-limx, txtz)) * t.z;
    t.y = wmin(limy, wmax(-limy, tytz)) * t.z;

    const mat3 J = M3(V3(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z)),
                      V3(0.0f, focal_y / t.z, -(focal_y * t.y) / (t.z * t.z)),
                      V3(0.0f, 0.0f, 0.0f));
    const mat3 W = M3(viewmatrix[0].x, viewmatrix[1].x, viewmatrix[2].x,
                      viewmatrix[0].y, viewmatrix[1].y, viewmatrix[2].y,
                      viewmatrix[0].z, viewmatrix[1].z, viewmatrix[2].z);
    const mat3 T = W * J;
    const mat3 Vrk = M3(cov_3D.v[0], cov_3D.v[1], cov_3D.v[2],
                        cov_3D.v[1], cov_3D.v[3], cov_3D.v[4],
                        cov_3D.v[2], cov_3D.v[4], cov_3D.v[5]);
    mat3 cov = transpose(T) * transpose(Vrk) * T;
    cov[0][0] = cov[0][0] + 0.3f;
    cov[1][1] = cov[1][1] + 0.3f;
    return V3(cov[0][0], cov[0][1], cov[1][1]);
}

// tiled-forward.wgsl:7-24
const f32 SH_C0 = 0.28209479177387814f;
const f32 SH_C1 = 0.4886025119029199f;
const f32 SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
const f32 SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                      -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// tiled-forward.wgsl:63-86
static vec3 sh_coef(const u32* sh_buffer, u32 splat_idx, u32 c_idx) {
    const u32 base_word = splat_idx * 24u;
    f32 e[3];
    for (u32 k = 0; k < 3; k++) {
        const u32 elem = c_idx * 3u + k;
        const vec2 halves = unpack2x16float(sh_buffer[base_word + (elem >> 1u)]);
        e[k] = ((elem & 1u) == 0u) ? halves.x : halves.y;
    }
    return V3(e[0], e[1], e[2]);
}

// tiled-forward.wgsl:88-119
static vec3 computeColorFromSH(const u32* sh, vec3 dir, u32 v_idx, u32 sh_deg) {
    vec3 result = SH_C0 * sh_coef(sh, v_idx, 0u);
    if (sh_deg > 0u) {
        const f32 x = dir.x, y = dir.y, z = dir.z;
        result = result + (-SH_C1 * y * sh_coef(sh, v_idx, 1u) + SH_C1 * z * sh_coef(sh, v_idx, 2u) - SH_C1 * x * sh_coef(sh, v_idx, 3u));
        if (sh_deg > 1u) {
            const f32 xx = dir.x * dir.x, yy = dir.y * dir.y, zz = dir.z * dir.z;
            const f32 xy = dir.x * dir.y, yz = dir.y * dir.z, xz = dir.x * dir.z;
            result = result + (SH_C2[0] * xy * sh_coef(sh, v_idx, 4u) + SH_C2[1] * yz * sh_coef(sh, v_idx, 5u) +
                               SH_C2[2] * (2.0f * zz - xx - yy) * sh_coef(sh, v_idx, 6u) + SH_C2[3] * xz * sh_coef(sh, v_idx, 7u) +
                               SH_C2[4] * (xx - yy) * sh_coef(sh, v_idx, 8u));
            if (sh_deg > 2u) {
                result = result + (SH_C3[0] * y * (3.0f * xx - yy) * sh_coef(sh, v_idx, 9u) + SH_C3[1] * xy * z * sh_coef(sh, v_idx, 10u) +
                                   SH_C3[2] * y * (4.0f * zz - xx - yy) * sh_coef(sh, v_idx, 11u) +
                                   SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh_coef(sh, v_idx, 12u) +
                                   SH_C3[4] * x * (4.0f * zz - xx - yy) * sh_coef(sh, v_idx, 13u) +
                                   SH_C3[5] * z * (xx - yy) * sh_coef(sh, v_idx, 14u) + SH_C3[6] * x * (xx - 3.0f * yy) * sh_coef(sh, v_idx, 15u));
            }
        }
    }
    result = result + 0.5f;
    return max(V3(0.0f), result);
}

// tiled-forward.wgsl:121-136
// PIN: bitcast<u32> of a NaN is implementation-defined (which NaN an operation returns differs between machines: sign and payload), and here the bits
// become a sort key.  A NaN depth counts as the canonical quiet NaN 0x7FC00000 -- it sorts behind every number.
static u32 float_to_ordered_uint(f32 x) {
    const u32 bits = (x != x) ? 0x7FC00000u : f2bits(x);
    const u32 mask = ((bits & 0x80000000u) != 0u) ? 0xFFFFFFFFu : 0x80000000u;
    return bits ^ mask;
}
static u32 make_tile_key(u32 tile_id, u32 depth_ordered) { return ((tile_id + 1u) << 16u) | (depth_ordered >> 16u); }

static vec2 to_f16_precision(vec2 v) { return unpack2x16float(pack2x16float(v)); }
static vec2 clamp_finite(vec2 v, f32 limit) { return clamp(v, V2(-limit), V2(limit)); }

}  // namespace

extern "C" {

// K1  tiled-forward.wgsl:161-294.  splats/depths of rejected Gaussians are left untouched
// (the reference never clears them); tile_counts[idx] is always written.
void orc_project_count(u32 n, const u32* gaussians, const u32* sh_buffer, const f32* camera_f, const f32* settings_f,
                       const u32* tile_info, u32* splats, u32* depths, u32* tile_counts, u32* stats) {
    const CameraUniforms camera = load_camera(camera_f);
    RenderSettings settings;
    std::memcpy(&settings, settings_f, sizeof(settings));
    const u32 num_tiles_x = tile_info[0], num_tiles_y = tile_info[1];
    const u32 tileWidth = 16u, tileHeight = 16u;
    u32 visible = 0;
#pragma omp parallel for schedule(static) reduction(+ : visible)
    for (u32 idx = 0; idx < n; idx++) {
        tile_counts[idx] = 0u;
        const u32* g = gaussians + (size_t)idx * 6;
        const vec2 rot_0 = unpack2x16float(g[2]), rot_1 = unpack2x16float(g[3]);
        const vec4 quaternion = V4(rot_0.x, rot_0.y, rot_1.x, rot_1.y);
        const vec2 scale_0 = unpack2x16float(g[4]), scale_1 = unpack2x16float(g[5]);
        const vec3 gaussian_scale = exp(V3(scale_0.x, scale_0.y, scale_1.x));
        const vec2 pos_0 = unpack2x16float(g[0]), pos_1 = unpack2x16float(g[1]);
        const vec3 gaussian_position = V3(pos_0.x, pos_0.y, pos_1.x);
        const f32 gaussian_opacity = pos_1.y;
        const f32 opacity_sigmoid = 1.0f / (1.0f + wd_exp(-gaussian_opacity));

        const vec4 position_world = V4(gaussian_position, 1.0f);
        const vec4 world_to_view = camera.view * position_world;
        const vec4 view_to_clip = camera.proj * world_to_view;
        if (view_to_clip.w == 0.0f) continue;
        const vec3 gaussian_ndc = view_to_clip.xyz() / view_to_clip.w;
        if (gaussian_ndc.x < -1.2f || gaussian_ndc.x > 1.2f || gaussian_ndc.y < -1.2f || gaussian_ndc.y > 1.2f ||
            gaussian_ndc.z < 0.0f || gaussian_ndc.z > 1.0f)
            continue;

        const Cov3D C3D = covariance3D(quaternion, gaussian_scale);
        const vec2 viewport = V2(settings.viewport_x, settings.viewport_y);
        const vec3 C2D = covariance2D(C3D, world_to_view, camera.focal, viewport, camera.view);
        const f32 det = (C2D.x * C2D.z) - (C2D.y * C2D.y);
        if (det <= 0.0f) continue;
        const f32 det_inv = 1.0f / det;
        const vec3 conic = V3(C2D.z * det_inv, -C2D.y * det_inv, C2D.x * det_inv);
        const f32 disc = conic.y * conic.y - conic.x * conic.z;
        if (conic.x <= 0.0f || conic.z <= 0.0f || disc >= 0.0f) continue;

        const f32 opacity_threshold = 128.0f;
        const f32 t = 2.0f * wd_log(opacity_sigmoid * opacity_threshold);
        if (t <= 0.0f) continue;

        const f32 x_extent = wd_sqrt(t * conic.z / (-disc));
        const f32 y_extent = wd_sqrt(t * conic.x / (-disc));
        const f32 cap = (settings.max_splat_radius_px > 0.0f) ? settings.max_splat_radius_px : 1e9f;
        const f32 x_extent_cap = wmin(x_extent, cap), y_extent_cap = wmin(y_extent, cap);
        const vec2 ndc_store = to_f16_precision(clamp_finite(V2(gaussian_ndc.x, gaussian_ndc.y), 60000.0f));
        const vec2 pixel_center = (ndc_store * V2(0.5f, -0.5f) + 0.5f) * viewport;
        const f32 tile_margin = 2.0f;
        const vec2 extents_f16 = to_f16_precision(V2(x_extent_cap, y_extent_cap));
        const vec2 bbox_min_raw = pixel_center - extents_f16 - tile_margin;
        const vec2 bbox_max_raw = pixel_center + extents_f16 + tile_margin;
        if (bbox_max_raw.x < 0.0f || bbox_max_raw.y < 0.0f || bbox_min_raw.x >= viewport.x || bbox_min_raw.y >= viewport.y) continue;
        const vec2 bbox_min = max(bbox_min_raw, V2(0.0f));
        const vec2 bbox_max = min(bbox_max_raw, viewport - V2(1.0f));
        if (bbox_max.x < bbox_min.x || bbox_max.y < bbox_min.y) continue;

        const vec3 camera_world_pos = camera.view_inv[3].xyz();
        const vec3 color_direction = normalize(gaussian_position - camera_world_pos);
        const vec3 color = computeColorFromSH(sh_buffer, color_direction, idx, to_u32(settings.sh_deg));

        const u32 tile_min_x = to_u32(bbox_min.x) / tileWidth;
        const u32 tile_min_y = to_u32(bbox_min.y) / tileHeight;
        const u32 tile_max_x = std::min(to_u32(bbox_max.x) / tileWidth, num_tiles_x - 1u);
        const u32 tile_max_y = std::min(to_u32(bbox_max.y) / tileHeight, num_tiles_y - 1u);
        const u32 tiles_x = tile_max_x - tile_min_x + 1u, tiles_y = tile_max_y - tile_min_y + 1u;
        const u32 num_tiles = tiles_x * tiles_y;
        if (num_tiles > 2048u) continue;

        u32* s = splats + (size_t)idx * 6;
        s[0] = pack2x16float(ndc_store);
        s[1] = pack2x16float(V2(x_extent_cap, y_extent_cap));
        s[2] = pack2x16float(V2(conic.x, conic.y));
        s[3] = pack2x16float(V2(conic.z, 0.0f));
        s[4] = pack2x16float(clamp(V2(color.x, color.y), V2(0.0f), V2(1.0f)));
        s[5] = pack2x16float(V2(clamp(color.z, 0.0f, 1.0f), clamp(opacity_sigmoid, 0.0f, 1.0f)));
        depths[idx] = float_to_ordered_uint(world_to_view.z);
        tile_counts[idx] = num_tiles;
        visible += 1;
    }
    stats[1] = visible;  // TilePipelineStats.visible_gaussians (tiled-forward.wgsl:292)
}

// K2-K4 (prefix/prefix_sum.wgsl): exclusive scan, wrapping u32.
void orc_exclusive_scan(u32 n, const u32* in, u32* out) {
    u32 acc = 0;
    for (u32 i = 0; i < n; i++) { out[i] = acc; acc += in[i]; }
}

// K5 update-stats.wgsl:19-35
void orc_update_stats(u32 n, const u32* tile_offsets, const u32* tile_counts, u32* stats) {
    stats[0] = (n == 0u) ? 0u : tile_offsets[n - 1] + tile_counts[n - 1];
}

// K6 tiled-forward.wgsl:297-354.  `capacity` = allocated entries; the reference has no bounds check
//  -- the restatement reports overflow instead of writing out of bounds: returns the number
// of entries that did not fit.
u32 orc_emit(u32 n, const u32* splats, const u32* depths, const u32* tile_counts, const u32* tile_offsets,
             const f32* settings_f, const u32* tile_info, u32* tile_keys, u32* tile_indices, u32 capacity) {
    RenderSettings settings;
    std::memcpy(&settings, settings_f, sizeof(settings));
    const u32 num_tiles_x = tile_info[0], num_tiles_y = tile_info[1];
    u32 dropped = 0;
#pragma omp parallel for schedule(static) reduction(+ : dropped)
    for (u32 idx = 0; idx < n; idx++) {
        const u32 num_tiles = tile_counts[idx];
        if (num_tiles == 0u) continue;
        const u32 start_offset = tile_offsets[idx];
        const u32* s = splats + (size_t)idx * 6;
        const vec2 pos_packed = unpack2x16float(s[0]);
        const vec2 extents_raw = unpack2x16float(s[1]);
        const f32 cap = (settings.max_splat_radius_px > 0.0f) ? settings.max_splat_radius_px : 1e9f;
        const vec2 extents = min(extents_raw, V2(cap));
        const vec2 viewport = V2(settings.viewport_x, settings.viewport_y);
        const vec2 pixel_center = (pos_packed * V2(0.5f, -0.5f) + 0.5f) * viewport;
        const f32 tile_margin = 2.0f;
        const vec2 bbox_min_raw = pixel_center - extents - tile_margin;
        const vec2 bbox_max_raw = pixel_center + extents + tile_margin;
        if (bbox_max_raw.x < 0.0f || bbox_max_raw.y < 0.0f || bbox_min_raw.x >= viewport.x || bbox_min_raw.y >= viewport.y) continue;
        const vec2 bbox_min = max(bbox_min_raw, V2(0.0f));
        const vec2 bbox_max = min(bbox_max_raw, viewport - V2(1.0f));
        const u32 tile_min_x = to_u32(bbox_min.x) / 16u, tile_min_y = to_u32(bbox_min.y) / 16u;
        const u32 tile_max_x = std::min(to_u32(bbox_max.x) / 16u, num_tiles_x - 1u);
        const u32 tile_max_y = std::min(to_u32(bbox_max.y) / 16u, num_tiles_y - 1u);
        const u32 depth_ordered = depths[idx];
        u32 offset = 0u;
        for (u32 ty = tile_min_y; ty <= tile_max_y; ty++) {
            for (u32 tx = tile_min_x; tx <= tile_max_x; tx++) {
                const u32 tile_id = ty * num_tiles_x + tx;
                const u32 key_idx = start_offset + offset;
                if (key_idx < capacity) {
                    tile_keys[key_idx] = make_tile_key(tile_id, depth_ordered);
                    tile_indices[key_idx] = idx;
                } else {
                    dropped++;
                }
                offset++;
            }
        }
    }
    return dropped;
}

// K7-K11 sort/radix_sort.wgsl + sort_dynamic.ts:370-386: stable ascending LSD sort on the full 32-bit key,
// result back in the buffers K6 wrote.  Semantics only (4 stable counting passes).
void orc_sort_pairs(u32 count, u32* keys, u32* values) {
    std::vector<u32> k2(count), v2(count);
    u32* ks = keys; u32* vs = values; u32* kd = k2.data(); u32* vd = v2.data();
    for (int pass = 0; pass < 4; pass++) {
        size_t hist[257] = {0};
        const int shift = pass * 8;
        for (u32 i = 0; i < count; i++) hist[((ks[i] >> shift) & 0xFFu) + 1]++;
        for (int b = 0; b < 256; b++) hist[b + 1] += hist[b];
        for (u32 i = 0; i < count; i++) {
            const size_t d = hist[(ks[i] >> shift) & 0xFFu]++;
            kd[d] = ks[i]; vd[d] = vs[i];
        }
        std::swap(ks, kd); std::swap(vs, vd);
    }
    // 4 passes: data is back in keys/values.
}

// K12-K13 tile-ranges.wgsl:46-104.  ranges has total_tiles+1 entries.
void orc_tile_ranges(u32 total_entries, const u32* sorted_keys, u32 total_tiles, u32* ranges) {
    for (u32 i = 0; i <= total_tiles; i++) ranges[i] = 0xFFFFFFFFu;
    ranges[total_tiles] = total_entries;
    for (u32 i = 0; i < total_entries; i++) {
        const u32 key = sorted_keys[i];
        if (key == 0u) continue;
        const u32 encoded_tile = key >> 16u;
        if (encoded_tile == 0u) continue;
        const u32 tile_id = encoded_tile - 1u;
        if (tile_id >= total_tiles) continue;
        if (i < ranges[tile_id]) ranges[tile_id] = i;
    }
}

// K14 tiled-rasterizer.wgsl:82-273.
//   max_batches = 32 reproduces the reference's 8192-splats-per-tile cap; 0 lifts it.
//   Contraction choice (WGSL allows FMA anywhere; pinned here so the GPU path can match bit for bit):
//     q   = fma(fma(cx,dx, (2cy)*dy), dx, (cz*dy)*dy)
//     w   = alpha*(1-A);  C = fma(color, w, C);  A = A + w
//   rgba8unorm store: floor(clamp(v,0,1)*255 + 0.5) in binary32.
void orc_rasterize(const f32* settings_f, const u32* tile_info, const u32* splats, u32 num_splats, const u32* tile_offsets,
                   const u32* sorted_keys, const u32* sorted_indices, u32 total_entries, u32 max_batches,
                   uint8_t* out_rgba8, f32* out_alpha, u32* out_n_contrib) {
    RenderSettings settings;
    std::memcpy(&settings, settings_f, sizeof(settings));
    const u32 num_tiles_x = tile_info[0], num_tiles_y = tile_info[1], total_tiles = tile_info[2];
    const vec2 viewport = V2(settings.viewport_x, settings.viewport_y);
    const u32 W = to_u32(viewport.x), H = to_u32(viewport.y);
    const u32 BATCH_SIZE = 256u;
    struct SharedSplat { vec2 center_px, extents_px; vec3 conic, color; f32 opacity; };

#pragma omp parallel for schedule(dynamic, 4)
    for (u32 tile_id = 0; tile_id < num_tiles_x * num_tiles_y; tile_id++) {
        const u32 wg_x = tile_id % num_tiles_x, wg_y = tile_id / num_tiles_x;
        const bool tile_valid = tile_id < total_tiles;
        const u32 padded_count = total_entries;
        u32 start = tile_valid ? tile_offsets[tile_id] : 0u;
        const bool tile_has_data = tile_valid && start < padded_count && start < 0xFFFFFFFFu;

        // Stage the tile's entries exactly as the batches would see them: entries of this tile are
        // contiguous from `start`; a batch with no entry of this tile ends the walk (lines 187-192).
        std::vector<SharedSplat> staged;
        if (tile_has_data) {
            for (u32 batch = 0; max_batches == 0u || batch < max_batches; batch++) {
                bool has_any = false;
                for (u32 l = 0; l < BATCH_SIZE; l++) {
                    const u32 entry_idx = start + batch * BATCH_SIZE + l;
                    if (entry_idx >= padded_count) continue;
                    const u32 key = sorted_keys[entry_idx];
                    if (key == 0u) continue;
                    const u32 encoded_tile = key >> 16u;
                    const u32 entry_tile_id = (encoded_tile == 0u) ? 0xFFFFFFFFu : encoded_tile - 1u;
                    if (entry_tile_id != tile_id) continue;
                    const u32 gaussian_idx = sorted_indices[entry_idx];
                    if (gaussian_idx >= num_splats) continue;
                    const u32* s = splats + (size_t)gaussian_idx * 6;
                    const vec2 pos_ndc = unpack2x16float(s[0]);
                    const vec2 conic_xy = unpack2x16float(s[2]);
                    const vec2 conic_z = unpack2x16float(s[3]);
                    const vec2 color_rg = unpack2x16float(s[4]);
                    const vec2 color_ba = unpack2x16float(s[5]);
                    const vec2 extents_px_raw = unpack2x16float(s[1]);
                    SharedSplat sp;
                    sp.center_px = (pos_ndc * V2(0.5f, -0.5f) + 0.5f) * viewport;
                    const f32 cap = (settings.max_splat_radius_px > 0.0f) ? settings.max_splat_radius_px : 1e9f;
                    sp.extents_px = min(extents_px_raw, V2(cap));
                    sp.conic = V3(conic_xy.x, conic_xy.y, conic_z.x);
                    sp.color = V3(color_rg.x, color_rg.y, color_ba.x);
                    sp.opacity = color_ba.y;
                    staged.push_back(sp);
                    has_any = true;
                }
                if (!has_any) break;
            }
        }

        for (u32 ly = 0; ly < 16u; ly++) {
            for (u32 lx = 0; lx < 16u; lx++) {
                const u32 pixel_x = wg_x * 16u + lx, pixel_y = wg_y * 16u + ly;
                const bool in_bounds = pixel_x < W && pixel_y < H;
                if (!(tile_valid && in_bounds)) continue;
                const vec2 pixel = V2((f32)pixel_x + 0.5f, (f32)pixel_y + 0.5f);
                vec3 accum_color = V3(0.0f);
                f32 accum_alpha = 0.0f;
                u32 processed_in_tile = 0u, last_contributor = 0u;
                for (const SharedSplat& sp : staged) {
                    processed_in_tile += 1u;
                    const vec2 delta = pixel - sp.center_px;
                    if (std::fabs(delta.x) > sp.extents_px.x || std::fabs(delta.y) > sp.extents_px.y) continue;
                    if (settings.gaussian_mode < 0.5f) {  // point-cloud preview branch, lines 212-222
                        const f32 dist_sq = dot(delta, delta);
                        const f32 cap = (settings.max_splat_radius_px > 0.0f) ? settings.max_splat_radius_px : 1e9f;
                        const f32 limit = wmin(settings.point_size_px, cap);
                        if (dist_sq <= limit * limit) {
                            accum_color = V3(1.0f, 1.0f, 0.0f);
                            accum_alpha = 1.0f;
                            last_contributor = processed_in_tile;
                        }
                        continue;
                    }
                    if (accum_alpha > 0.99f) continue;
                    f32 exp_q;
                    if (g_literal_order) {
                        exp_q = (sp.conic.x * delta.x * delta.x) + (2.0f * sp.conic.y * delta.x * delta.y) + (sp.conic.z * delta.y * delta.y);
                    } else {
                        const f32 t1 = std::fmaf(sp.conic.x, delta.x, (2.0f * sp.conic.y) * delta.y);
                        exp_q = std::fmaf(t1, delta.x, (sp.conic.z * delta.y) * delta.y);
                    }
                    const f32 gaussian_weight = wd_exp(-0.5f * exp_q);
                    const f32 alpha = clamp(gaussian_weight * sp.opacity, 0.0f, 0.99f);
                    const f32 vis = 1.0f - accum_alpha;
                    if (g_literal_order) {
                        accum_color = accum_color + sp.color * alpha * vis;  // (color * alpha) * vis, then the add
                        accum_alpha = accum_alpha + alpha * vis;
                    } else {
                        const f32 w = alpha * vis;
                        accum_color.x = std::fmaf(sp.color.x, w, accum_color.x);
                        accum_color.y = std::fmaf(sp.color.y, w, accum_color.y);
                        accum_color.z = std::fmaf(sp.color.z, w, accum_color.z);
                        accum_alpha = accum_alpha + w;
                    }
                    if (alpha >= (1.0f / 255.0f)) last_contributor = processed_in_tile;
                }
                const size_t p = (size_t)pixel_y * W + pixel_x;
                const f32 vis = 1.0f - accum_alpha;
                const vec3 final_color = accum_color + V3(0.0f) * vis;  // BACKGROUND_COLOR.rgb = 0
                for (int c = 0; c < 3; c++) out_rgba8[p * 4 + c] = (uint8_t)to_u32(clamp(final_color[c], 0.0f, 1.0f) * 255.0f + 0.5f);
                out_rgba8[p * 4 + 3] = 255;
                out_alpha[p] = 1.0f - accum_alpha;
                out_n_contrib[p] = last_contributor;
            }
        }
    }
}

}  // extern "C"

namespace wgsl { int g_literal_order = 0; }
extern "C" void orc_set_literal_order(int on) { wgsl::g_literal_order = on ? 1 : 0; }
extern "C" int orc_get_literal_order() { return wgsl::g_literal_order; }

// ---- test hooks for tests/test_oracle_math.py (dmath accuracy vs libm, fp16 conversions vs numpy)
extern "C" {
void orc_test_exp(u32 n, const f32* in, f32* out) { for (u32 i = 0; i < n; i++) out[i] = wd_exp(in[i]); }
void orc_test_log(u32 n, const f32* in, f32* out) { for (u32 i = 0; i < n; i++) out[i] = wd_log(in[i]); }
void orc_test_f32_to_f16(u32 n, const f32* in, uint16_t* out) { for (u32 i = 0; i < n; i++) out[i] = f32_to_f16(in[i]); }
void orc_test_f16_to_f32(u32 n, const uint16_t* in, f32* out) { for (u32 i = 0; i < n; i++) out[i] = f16_to_f32(in[i]); }
void orc_test_to_i32(u32 n, const f32* in, i32* out) { for (u32 i = 0; i < n; i++) out[i] = to_i32(in[i]); }
void orc_test_to_u32(u32 n, const f32* in, u32* out) { for (u32 i = 0; i < n; i++) out[i] = to_u32(in[i]); }
void orc_test_sqrt(u32 n, const f32* in, f32* out) { for (u32 i = 0; i < n; i++) out[i] = wd_sqrt(in[i]); }
void orc_test_rcp(u32 n, const f32* in, f32* out) { for (u32 i = 0; i < n; i++) out[i] = 1.0f / in[i]; }
}
