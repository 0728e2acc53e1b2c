// ORACLE (test infrastructure only; PARITY UNPINNED -- see wgsl_shim.hpp).
//
// CPU restatement of the reference's loss / backward / optimizer kernels.  Citations relative
// to /root/reference/src.
//
//   K15 compute_loss_grad        shaders/loss.wgsl:85-115 (computeSSIMGrad 30-82)
//   K16 backward_rasterize_main  shaders/tiled-backward-rasterize.wgsl:34-172 + common.wgsl:110-121
//   K17 main_geometry_backward   shaders/tiled-backward.wgsl:41-298
//   K18 Adam                     shaders/adam.wgsl:53-175
//   K19 re-pack                  shaders/update-gaussians.wgsl:35-77
//   K20 optimizer unpack         renderers/optimizer.ts:166-223
#include "wgsl_shim.hpp"

#include <vector>

using namespace wgsl;

namespace {

struct CameraUniforms { mat4 view, view_inv, proj, proj_inv; vec2 viewport, focal; };
struct RenderSettings { f32 gaussian_scaling, sh_deg, viewport_x, viewport_y, point_size_px, gaussian_mode, max_splat_radius_px; };

struct Cov3D { f32 v[6]; };
static Cov3D covariance3D(vec4 quaternion, vec3 scale) {  // common.wgsl:44-68
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

static const f32 GRAD_SCALE = 1000000.0f;  // common.wgsl:111
static inline i32 to_fixed(f32 value) { return to_i32(value * GRAD_SCALE); }              // common.wgsl:113-116
// atomicAdd on atomic<i32> wraps (two's complement); accumulate through u32 so the host addition wraps too.
static inline void acc_add(i32* p, i32 v) {
    u32* up = reinterpret_cast<u32*>(p);
#pragma omp atomic
    *up += (u32)v;
}
static inline f32 from_fixed(i32 fixed_val) { return (f32)fixed_val / GRAD_SCALE; }       // common.wgsl:118-121

static inline vec3 texel_rgb(const uint8_t* img, u32 W, u32 H, i32 x, i32 y) {  // clamp-to-edge textureLoad of rgba8unorm
    const i32 cx = x < 0 ? 0 : (x > (i32)W - 1 ? (i32)W - 1 : x);
    const i32 cy = y < 0 ? 0 : (y > (i32)H - 1 ? (i32)H - 1 : y);
    const uint8_t* p = img + ((size_t)cy * W + (size_t)cx) * 4;
    return V3((f32)p[0] / 255.0f, (f32)p[1] / 255.0f, (f32)p[2] / 255.0f);
}

// loss.wgsl:30-82
static vec3 computeSSIMGrad(const uint8_t* pred_img, const uint8_t* targ_img, u32 W, u32 H, i32 cx, i32 cy, f32 c1, f32 c2) {
    vec3 mu_x = V3(0.0f), mu_y = V3(0.0f);
    const i32 half_window = 2;
    const f32 n = 25.0f;
    for (i32 dy = -half_window; dy <= half_window; dy++)
        for (i32 dx = -half_window; dx <= half_window; dx++) {
            mu_x = mu_x + texel_rgb(pred_img, W, H, cx + dx, cy + dy);
            mu_y = mu_y + texel_rgb(targ_img, W, H, cx + dx, cy + dy);
        }
    mu_x = mu_x / n;
    mu_y = mu_y / n;
    vec3 sigma_x2 = V3(0.0f), sigma_y2 = V3(0.0f), sigma_xy = V3(0.0f);
    for (i32 dy = -half_window; dy <= half_window; dy++)
        for (i32 dx = -half_window; dx <= half_window; dx++) {
            const vec3 x = texel_rgb(pred_img, W, H, cx + dx, cy + dy);
            const vec3 y = texel_rgb(targ_img, W, H, cx + dx, cy + dy);
            const vec3 dx_val = x - mu_x, dy_val = y - mu_y;
            if (g_literal_order) {
                sigma_x2 = sigma_x2 + dx_val * dx_val;
                sigma_y2 = sigma_y2 + dy_val * dy_val;
                sigma_xy = sigma_xy + dx_val * dy_val;
            } else {
                // pinned contraction (WGSL may fuse a multiply into the add that consumes it): the window accumulations are FMAs
                for (int c = 0; c < 3; c++) {
                    sigma_x2[c] = std::fmaf(dx_val[c], dx_val[c], sigma_x2[c]);
                    sigma_y2[c] = std::fmaf(dy_val[c], dy_val[c], sigma_y2[c]);
                    sigma_xy[c] = std::fmaf(dx_val[c], dy_val[c], sigma_xy[c]);
                }
            }
        }
    sigma_x2 = sigma_x2 / n;
    sigma_y2 = sigma_y2 / n;
    sigma_xy = sigma_xy / n;
    const vec3 num1 = 2.0f * mu_x * mu_y + c1;
    const vec3 num2 = 2.0f * sigma_xy + c2;
    const vec3 den1 = mu_x * mu_x + mu_y * mu_y + c1;
    const vec3 den2 = sigma_x2 + sigma_y2 + c2;
    const vec3 ssim = (num1 * num2) / (den1 * den2);
    const vec3 pred = texel_rgb(pred_img, W, H, cx, cy);
    const vec3 targ = texel_rgb(targ_img, W, H, cx, cy);
    const vec3 dssim = (V3(1.0f) - ssim) * 0.5f;
    return dssim * (pred - targ);
}

}  // namespace

extern "C" {

// K15 loss.wgsl:85-115.  cfg = {lambda_l1, lambda_l2, lambda_dssim, c1, c2}; out = rgba32float (a = 1).
void orc_loss_grad(u32 W, u32 H, const uint8_t* pred, const uint8_t* targ, const f32* cfg, f32* out) {
    const f32 lambda_l1 = cfg[0], lambda_l2 = cfg[1], lambda_dssim = cfg[2], c1 = cfg[3], c2 = cfg[4];
#pragma omp parallel for schedule(static)
    for (u32 y = 0; y < H; y++)
        for (u32 x = 0; x < W; x++) {
            const vec3 p = texel_rgb(pred, W, H, (i32)x, (i32)y);
            const vec3 t = texel_rgb(targ, W, H, (i32)x, (i32)y);
            const vec3 diff = p - t;
            const vec3 grad_l1 = sign(diff);
            const vec3 grad_l2 = diff;
            vec3 grad_dssim = V3(0.0f);
            if (lambda_dssim > 0.0f) grad_dssim = computeSSIMGrad(pred, targ, W, H, (i32)x, (i32)y, c1, c2);
            const vec3 total_grad = lambda_l1 * grad_l1 + lambda_l2 * grad_l2 + lambda_dssim * grad_dssim;
            f32* o = out + ((size_t)y * W + x) * 4;
            o[0] = total_grad.x; o[1] = total_grad.y; o[2] = total_grad.z; o[3] = 1.0f;
        }
}

// K16 tiled-backward-rasterize.wgsl:34-172.  Accumulators are the reference's four i32 arrays
// (means[2N], conics[4N] slots 0,1,3, opacity[N], colors[3N]); they are added to, not cleared here
// (the reference clears them with clearBuffer before the pass, tiled-backward-pass.ts:624-627).
// Contraction choices pinned as in K14: power = fma(fma(cx,dx,(2cy)*dy), dx, (cz*dy)*dy); the accum_rec, dL_dalpha and dpow
// multiply-adds are FMAs (see the loop).
void orc_backward_rasterize(const f32* settings_f, const u32* tile_offsets, const u32* tile_instances, const u32* splats,
                            const f32* final_Ts, const u32* n_contrib_tex, const f32* loss_gradient,
                            i32* grad_means_2d, i32* grad_conics, i32* grad_opacity, i32* grad_colors) {
    RenderSettings settings;
    std::memcpy(&settings, settings_f, sizeof(settings));
    const u32 width = to_u32(settings.viewport_x), height = to_u32(settings.viewport_y);
    const u32 BLOCK_SIZE = 16u;
    const u32 num_tiles_x = (width + BLOCK_SIZE - 1u) / BLOCK_SIZE;
    const vec2 viewport = V2(settings.viewport_x, settings.viewport_y);
    // The reference runs one invocation per pixel and adds every contribution to the global accumulators with atomicAdd.  Integer
    // addition is associative and commutative (wrapping), so the sums do not depend on the order or grouping of the adds: here the
    // pixels of one 16x16 tile (one workgroup of the reference) first add into a tile-local table indexed by the entry's position
    // in the tile list (no atomics, thread-private), and the table is added to the global arrays once per (tile, entry).  The
    // per-pixel body is the reference's, statement for statement.
    const u32 num_tiles_y = (height + BLOCK_SIZE - 1u) / BLOCK_SIZE;
#pragma omp parallel for schedule(dynamic, 4)
    for (u32 tile_idx = 0; tile_idx < num_tiles_x * num_tiles_y; tile_idx++) {
        const u32 tile_x = tile_idx % num_tiles_x, tile_y = tile_idx / num_tiles_x;
        const u32 range_start = tile_offsets[tile_idx];
        const u32 range_end = tile_offsets[tile_idx + 1];
        const u32 tile_entries = (range_end > range_start) ? range_end - range_start : 0u;
        u32 max_n = 0u;
        for (u32 ly = 0; ly < 16u; ly++)
            for (u32 lx = 0; lx < 16u; lx++) {
                const u32 px = tile_x * 16u + lx, py = tile_y * 16u + ly;
                if (px < width && py < height) max_n = std::max(max_n, std::min(n_contrib_tex[(size_t)py * width + px], tile_entries));
            }
        if (max_n == 0u) continue;
        std::vector<u32> local((size_t)max_n * 9u, 0u);  // per entry: mean.x, mean.y, conic.x, conic.y, conic.z, opacity, r, g, b
        for (u32 lpix = 0; lpix < 256u; lpix++) {
            const u32 px = tile_x * 16u + (lpix & 15u), py = tile_y * 16u + (lpix >> 4);
            if (px >= width || py >= height) continue;
            const size_t p = (size_t)py * width + px;
            const u32 n_contrib_val = n_contrib_tex[p];
            if (n_contrib_val == 0u) continue;
            const u32 pix_n_contrib = std::min(n_contrib_val, tile_entries);
            f32 T = final_Ts[p];
            const f32* dL_dpixel = loss_gradient + p * 4;
            const vec2 pixf = V2((f32)px, (f32)py) + 0.5f;
            vec3 accum_rec = V3(0.0f), last_color = V3(0.0f);
            f32 last_alpha = 0.0f;
            for (u32 i = pix_n_contrib; i > 0u; i--) {
                const u32 idx_in_tile = i - 1u;
                const u32 g = tile_instances[range_start + idx_in_tile];
                u32* const L = &local[(size_t)idx_in_tile * 9u];
                const u32* s = splats + (size_t)g * 6;
                const vec2 pos_ndc = unpack2x16float(s[0]);
                const vec2 conic_xy = unpack2x16float(s[2]);
                const vec2 conic_z = unpack2x16float(s[3]);
                const vec2 color_rg = unpack2x16float(s[4]);
                const vec2 color_ba = unpack2x16float(s[5]);
                const vec2 extents_raw = unpack2x16float(s[1]);
                const vec2 center_px = (pos_ndc * V2(0.5f, -0.5f) + 0.5f) * viewport;
                const vec3 conic = V3(conic_xy.x, conic_xy.y, conic_z.x);
                const vec3 color = V3(color_rg.x, color_rg.y, color_ba.x);
                const f32 opacity = color_ba.y;
                const f32 cap = (settings.max_splat_radius_px > 0.0f) ? settings.max_splat_radius_px : 1e9f;
                const vec2 extents = min(extents_raw, V2(cap));
                const vec2 delta = pixf - center_px;
                if (std::fabs(delta.x) > extents.x || std::fabs(delta.y) > extents.y) continue;
                f32 power;
                if (g_literal_order) {
                    power = (conic.x * delta.x * delta.x) + (2.0f * conic.y * delta.x * delta.y) + (conic.z * delta.y * delta.y);
                } else {
                    const f32 t1 = std::fmaf(conic.x, delta.x, (2.0f * conic.y) * delta.y);
                    power = std::fmaf(t1, delta.x, (conic.z * delta.y) * delta.y);
                }
                const f32 G = wd_exp(-0.5f * power);
                const f32 alpha = wmin(0.99f, opacity * G);
                if (alpha < 1.0f / 255.0f) continue;
                T = T / (1.0f - alpha);
                f32 dL_dalpha = 0.0f;
                // Pinned contraction (WGSL may fuse a multiply into the add that consumes it; `literal` keeps one rounding per operator):
                //   accum_rec = fma(last_alpha, last_color, (1 - last_alpha) * accum_rec);  dL_dalpha = fma(color - accum_rec, grad, dL_dalpha);
                //   dpow_dx = fma(2 cx, dx, (2 cy) dy);  dpow_dy = fma(2 cz, dy, (2 cy) dx)
                if (g_literal_order) {
                    accum_rec = last_alpha * last_color + (1.0f - last_alpha) * accum_rec;
                } else {
                    for (u32 ch = 0; ch < 3u; ch++) accum_rec[ch] = std::fmaf(last_alpha, last_color[ch], (1.0f - last_alpha) * accum_rec[ch]);
                }
                for (u32 ch = 0; ch < 3u; ch++) {
                    const f32 grad_pix = dL_dpixel[ch];
                    const f32 dchannel_dcolor = alpha * T;
                    const f32 dL_dc = dchannel_dcolor * grad_pix;
                    L[6u + ch] += (u32)to_fixed(dL_dc);
                    if (g_literal_order) dL_dalpha += (color[ch] - accum_rec[ch]) * grad_pix;
                    else dL_dalpha = std::fmaf(color[ch] - accum_rec[ch], grad_pix, dL_dalpha);
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                last_color = color;
                const f32 dL_dG = opacity * dL_dalpha;
                const f32 dL_dopacity = G * dL_dalpha;
                const f32 dpow_dx = g_literal_order ? 2.0f * conic.x * delta.x + 2.0f * conic.y * delta.y
                                                    : std::fmaf(2.0f * conic.x, delta.x, (2.0f * conic.y) * delta.y);
                const f32 dpow_dy = g_literal_order ? 2.0f * conic.z * delta.y + 2.0f * conic.y * delta.x
                                                    : std::fmaf(2.0f * conic.z, delta.y, (2.0f * conic.y) * delta.x);
                const f32 dG_ddelta_x = -0.5f * G * dpow_dx;
                const f32 dG_ddelta_y = -0.5f * G * dpow_dy;
                const f32 dL_dmean_x = dL_dG * (-dG_ddelta_x);
                const f32 dL_dmean_y = dL_dG * (-dG_ddelta_y);
                const f32 dL_dconic_x = dL_dG * (-0.5f * G * delta.x * delta.x);
                const f32 dL_dconic_y = dL_dG * (-0.5f * G * 2.0f * delta.x * delta.y);
                const f32 dL_dconic_z = dL_dG * (-0.5f * G * delta.y * delta.y);
                L[5] += (u32)to_fixed(dL_dopacity);
                L[0] += (u32)to_fixed(dL_dmean_x);
                L[1] += (u32)to_fixed(dL_dmean_y);
                L[2] += (u32)to_fixed(dL_dconic_x);
                L[3] += (u32)to_fixed(dL_dconic_y);
                L[4] += (u32)to_fixed(dL_dconic_z);
            }
        }
        for (u32 e = 0; e < max_n; e++) {
            const u32* L = &local[(size_t)e * 9u];
            const u32 g = tile_instances[range_start + e];
            if (L[0]) acc_add(&grad_means_2d[(size_t)g * 2u + 0u], (i32)L[0]);
            if (L[1]) acc_add(&grad_means_2d[(size_t)g * 2u + 1u], (i32)L[1]);
            if (L[2]) acc_add(&grad_conics[(size_t)g * 4u + 0u], (i32)L[2]);
            if (L[3]) acc_add(&grad_conics[(size_t)g * 4u + 1u], (i32)L[3]);
            if (L[4]) acc_add(&grad_conics[(size_t)g * 4u + 3u], (i32)L[4]);
            if (L[5]) acc_add(&grad_opacity[g], (i32)L[5]);
            for (u32 ch = 0; ch < 3u; ch++)
                if (L[6u + ch]) acc_add(&grad_colors[(size_t)g * 3u + ch], (i32)L[6u + ch]);
        }
    }
}

// TEST-ONLY switches (tests/test_oracle_independent.py, default 0 = the reference's K17 as written).  They undo, one by one, the two
// places where tiled-backward.wgsl disagrees with the forward pass it differentiates, so that a float64 finite-difference check can show
// that these two are the ONLY deviations of the restated chain rule from the true gradient:
//   bit 0  SURVEY Q10: rebuild the 2D covariance with W = transpose(view3x3), as the forward does (common.wgsl:90-95), instead of view3x3
//          (tiled-backward.wgsl:134-140);
//   bit 1  SURVEY Q11: dL/dndc.y = dL/dpx.y * (-0.5 * viewport.y), the derivative of px.y = (-0.5 ndc.y + 0.5) * viewport.y
//          (tiled-forward.wgsl:237), instead of +0.5 * viewport.y (tiled-backward.wgsl:92).
//   bit 2  "Q23" (found by the finite-difference check itself, not listed in SURVEY): K16 accumulates the FULL partial derivative with
//          respect to conic.y (tiled-backward-rasterize.wgsl:163: -0.5 G * 2.0 * dx * dy), while K17's expressions for dL/da, dL/db,
//          dL/dc (tiled-backward.wgsl:161-163) are the CUDA rasterizer's, which expect HALF of it (its K16 stores -0.5 G dx dy):
//          the off-diagonal path is counted twice.  The switch halves dL/dconic.y on entry.
// No product path and no parity test sets them.
static int g_k17_fix = 0;
void orc_set_k17_fix(int flags) { g_k17_fix = flags; }

// K17 tiled-backward.wgsl:41-298.  gradients = GaussianGradient[n] (8 u32 each).
void orc_geometry_backward(u32 n, const f32* camera_f, const f32* settings_f, const u32* gaussians,
                           const i32* grad_means_2d, const i32* grad_conics, const i32* grad_opacity, const i32* grad_colors,
                           u32* gradients) {
    CameraUniforms camera;
    std::memcpy(&camera, camera_f, sizeof(f32) * 68);
    RenderSettings settings;
    std::memcpy(&settings, settings_f, sizeof(settings));
#pragma omp parallel for schedule(static)
    for (u32 idx = 0; idx < n; idx++) {
        vec2 dL_dmean2D_px = V2(from_fixed(grad_means_2d[(size_t)idx * 2u + 0u]), from_fixed(grad_means_2d[(size_t)idx * 2u + 1u]));
        vec3 dL_dconic = V3(from_fixed(grad_conics[(size_t)idx * 4u + 0u]), from_fixed(grad_conics[(size_t)idx * 4u + 1u]),
                            from_fixed(grad_conics[(size_t)idx * 4u + 3u]));
        if (g_k17_fix & 4) dL_dconic.y = dL_dconic.y * 0.5f;  // (test-only: Q23 undone)
        const f32 dL_dopac = from_fixed(grad_opacity[idx]);

        const u32* g = gaussians + (size_t)idx * 6;
        const vec2 p_xy = unpack2x16float(g[0]), p_z_op = unpack2x16float(g[1]);
        const vec3 mean3D = V3(p_xy.x, p_xy.y, p_z_op.x);
        const f32 opacity_raw = p_z_op.y;
        const f32 opacity_sigmoid = 1.0f / (1.0f + wd_exp(-opacity_raw));
        const vec2 r_xy = unpack2x16float(g[2]), r_zw = unpack2x16float(g[3]);
        const vec4 rot = V4(r_xy.x, r_xy.y, r_zw.x, r_zw.y);
        const vec2 s_xy = unpack2x16float(g[4]), s_z_ = unpack2x16float(g[5]);
        const vec3 log_scale = V3(s_xy.x, s_xy.y, s_z_.x);
        const vec3 scale = exp(log_scale);

        const Cov3D cov3D_flat = covariance3D(rot, scale);
        const mat4 view = camera.view;
        const vec3 t = (view * V4(mean3D, 1.0f)).xyz();

        const vec2 viewport = V2(settings.viewport_x, settings.viewport_y);
        vec2 dL_dmean2D_ndc = dL_dmean2D_px * 0.5f * viewport;
        if (g_k17_fix & 2) dL_dmean2D_ndc.y = -dL_dmean2D_ndc.y;  // (test-only: Q11 undone)

        const mat4 view_proj = camera.proj * camera.view;
        const vec4 p_hom = view_proj * V4(mean3D, 1.0f);
        const f32 rw = 1.0f / (p_hom.w + 0.0000001f);
        const f32 rw2 = rw * rw;
        const vec4 dL_dphom = V4(dL_dmean2D_ndc.x * rw, dL_dmean2D_ndc.y * rw, 0.0f,
                                 -(dL_dmean2D_ndc.x * p_hom.x + dL_dmean2D_ndc.y * p_hom.y) * rw2);
        const vec3 dL_dmean3D_proj = (transpose(view_proj) * dL_dphom).xyz();

        const f32 focal_x = camera.focal.x, focal_y = camera.focal.y;
        const f32 limx = 1.3f * viewport.x * 0.5f / focal_x;
        const f32 limy = 1.3f * viewport.y * 0.5f / focal_y;
        const f32 txtz = t.x / t.z, tytz = t.y / t.z;
        const f32 t_clamped_x = wmin(limx, wmax(-limx, txtz)) * t.z;
        const f32 t_clamped_y = wmin(limy, wmax(-limy, tytz)) * t.z;
        f32 x_grad_mul = 0.0f;
        if (txtz >= -limx && txtz <= limx) x_grad_mul = 1.0f;
        f32 y_grad_mul = 0.0f;
        if (tytz >= -limy && tytz <= limy) y_grad_mul = 1.0f;

        const mat3 J = M3(V3(focal_x / t.z, 0.0f, -(focal_x * t_clamped_x) / (t.z * t.z)),
                          V3(0.0f, focal_y / t.z, -(focal_y * t_clamped_y) / (t.z * t.z)),
                          V3(0.0f, 0.0f, 0.0f));
        const mat3 W_ref = M3(view[0].xyz(), view[1].xyz(), view[2].xyz());
        const mat3 W = (g_k17_fix & 1) ? transpose(W_ref) : W_ref;  // (test-only: Q10 undone)
        const mat3 T_mat = W * J;
        const mat3 Vrk = M3(V3(cov3D_flat.v[0], cov3D_flat.v[1], cov3D_flat.v[2]),
                            V3(cov3D_flat.v[1], cov3D_flat.v[3], cov3D_flat.v[4]),
                            V3(cov3D_flat.v[2], cov3D_flat.v[4], cov3D_flat.v[5]));
        const mat3 cov2D = transpose(T_mat) * Vrk * T_mat;
        const f32 a = cov2D[0][0] + 0.3f;
        const f32 b = cov2D[0][1];
        const f32 c = cov2D[1][1] + 0.3f;

        const f32 denom = a * c - b * b;
        const f32 denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        f32 dL_da = 0.0f, dL_db = 0.0f, dL_dc = 0.0f;
        if (denom2inv != 0.0f) {
            dL_da = denom2inv * (-c * c * dL_dconic.x + 2.0f * b * c * dL_dconic.y + (denom - a * c) * dL_dconic.z);
            dL_dc = denom2inv * (-a * a * dL_dconic.z + 2.0f * a * b * dL_dconic.y + (denom - a * c) * dL_dconic.x);
            dL_db = denom2inv * 2.0f * (b * c * dL_dconic.x - (denom + 2.0f * b * b) * dL_dconic.y + a * b * dL_dconic.z);
        }

        f32 dL_dcov3D_flat[6];
        dL_dcov3D_flat[0] = (T_mat[0][0] * T_mat[0][0] * dL_da + T_mat[0][0] * T_mat[1][0] * dL_db + T_mat[1][0] * T_mat[1][0] * dL_dc);
        dL_dcov3D_flat[3] = (T_mat[0][1] * T_mat[0][1] * dL_da + T_mat[0][1] * T_mat[1][1] * dL_db + T_mat[1][1] * T_mat[1][1] * dL_dc);
        dL_dcov3D_flat[5] = (T_mat[0][2] * T_mat[0][2] * dL_da + T_mat[0][2] * T_mat[1][2] * dL_db + T_mat[1][2] * T_mat[1][2] * dL_dc);
        dL_dcov3D_flat[1] = 2.0f * T_mat[0][0] * T_mat[0][1] * dL_da + (T_mat[0][0] * T_mat[1][1] + T_mat[0][1] * T_mat[1][0]) * dL_db + 2.0f * T_mat[1][0] * T_mat[1][1] * dL_dc;
        dL_dcov3D_flat[2] = 2.0f * T_mat[0][0] * T_mat[0][2] * dL_da + (T_mat[0][0] * T_mat[1][2] + T_mat[0][2] * T_mat[1][0]) * dL_db + 2.0f * T_mat[1][0] * T_mat[1][2] * dL_dc;
        dL_dcov3D_flat[4] = 2.0f * T_mat[0][2] * T_mat[0][1] * dL_da + (T_mat[0][1] * T_mat[1][2] + T_mat[0][2] * T_mat[1][1]) * dL_db + 2.0f * T_mat[1][1] * T_mat[1][2] * dL_dc;

        const f32 dL_dT00 = 2.0f * (T_mat[0][0] * Vrk[0][0] + T_mat[0][1] * Vrk[0][1] + T_mat[0][2] * Vrk[0][2]) * dL_da +
                            (T_mat[1][0] * Vrk[0][0] + T_mat[1][1] * Vrk[0][1] + T_mat[1][2] * Vrk[0][2]) * dL_db;
        const f32 dL_dT01 = 2.0f * (T_mat[0][0] * Vrk[1][0] + T_mat[0][1] * Vrk[1][1] + T_mat[0][2] * Vrk[1][2]) * dL_da +
                            (T_mat[1][0] * Vrk[1][0] + T_mat[1][1] * Vrk[1][1] + T_mat[1][2] * Vrk[1][2]) * dL_db;
        const f32 dL_dT02 = 2.0f * (T_mat[0][0] * Vrk[2][0] + T_mat[0][1] * Vrk[2][1] + T_mat[0][2] * Vrk[2][2]) * dL_da +
                            (T_mat[1][0] * Vrk[2][0] + T_mat[1][1] * Vrk[2][1] + T_mat[1][2] * Vrk[2][2]) * dL_db;
        const f32 dL_dT10 = 2.0f * (T_mat[1][0] * Vrk[0][0] + T_mat[1][1] * Vrk[0][1] + T_mat[1][2] * Vrk[0][2]) * dL_dc +
                            (T_mat[0][0] * Vrk[0][0] + T_mat[0][1] * Vrk[0][1] + T_mat[0][2] * Vrk[0][2]) * dL_db;
        const f32 dL_dT11 = 2.0f * (T_mat[1][0] * Vrk[1][0] + T_mat[1][1] * Vrk[1][1] + T_mat[1][2] * Vrk[1][2]) * dL_dc +
                            (T_mat[0][0] * Vrk[1][0] + T_mat[0][1] * Vrk[1][1] + T_mat[0][2] * Vrk[1][2]) * dL_db;
        const f32 dL_dT12 = 2.0f * (T_mat[1][0] * Vrk[2][0] + T_mat[1][1] * Vrk[2][1] + T_mat[1][2] * Vrk[2][2]) * dL_dc +
                            (T_mat[0][0] * Vrk[2][0] + T_mat[0][1] * Vrk[2][1] + T_mat[0][2] * Vrk[2][2]) * dL_db;

        const f32 dL_dJ00 = W[0][0] * dL_dT00 + W[0][1] * dL_dT01 + W[0][2] * dL_dT02;
        const f32 dL_dJ02 = W[2][0] * dL_dT00 + W[2][1] * dL_dT01 + W[2][2] * dL_dT02;
        const f32 dL_dJ11 = W[1][0] * dL_dT10 + W[1][1] * dL_dT11 + W[1][2] * dL_dT12;
        const f32 dL_dJ12 = W[2][0] * dL_dT10 + W[2][1] * dL_dT11 + W[2][2] * dL_dT12;

        const f32 tz = 1.0f / t.z;
        const f32 tz2 = tz * tz;
        const f32 tz3 = tz2 * tz;
        const f32 dL_dtx = x_grad_mul * -focal_x * tz2 * dL_dJ02;
        const f32 dL_dty = y_grad_mul * -focal_y * tz2 * dL_dJ12;
        const f32 dL_dtz = -focal_x * tz2 * dL_dJ00 - focal_y * tz2 * dL_dJ11 + (2.0f * focal_x * t_clamped_x) * tz3 * dL_dJ02 +
                           (2.0f * focal_y * t_clamped_y) * tz3 * dL_dJ12;
        const vec3 dL_dmean3D_cov = (transpose(view) * V4(dL_dtx, dL_dty, dL_dtz, 0.0f)).xyz();

        const f32 x = rot.y, y = rot.z, z = rot.w, r = rot.x;
        const mat3 R = M3(
            V3(1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y - r * z), 2.0f * (x * z + r * y)),
            V3(2.0f * (x * y + r * z), 1.0f - 2.0f * (x * x + z * z), 2.0f * (y * z - r * x)),
            V3(2.0f * (x * z - r * y), 2.0f * (y * z + r * x), 1.0f - 2.0f * (x * x + y * y)));
        const mat3 S = M3(V3(scale.x, 0.0f, 0.0f), V3(0.0f, scale.y, 0.0f), V3(0.0f, 0.0f, scale.z));
        const mat3 M = S * R;
        const mat3 dL_dSigma = M3(V3(dL_dcov3D_flat[0], 0.5f * dL_dcov3D_flat[1], 0.5f * dL_dcov3D_flat[2]),
                                  V3(0.5f * dL_dcov3D_flat[1], dL_dcov3D_flat[3], 0.5f * dL_dcov3D_flat[4]),
                                  V3(0.5f * dL_dcov3D_flat[2], 0.5f * dL_dcov3D_flat[4], dL_dcov3D_flat[5]));
        const mat3 dL_dM = 2.0f * M * dL_dSigma;  // (2.0*M)*dL_dSigma, left to right
        const mat3 dL_dMt = transpose(dL_dM);
        const mat3 Rt = transpose(R);
        const vec3 dL_dscale = V3(dot(Rt[0], dL_dMt[0]), dot(Rt[1], dL_dMt[1]), dot(Rt[2], dL_dMt[2]));
        mat3 dL_dMt_scaled = dL_dMt;
        dL_dMt_scaled[0] = dL_dMt[0] * scale.x;
        dL_dMt_scaled[1] = dL_dMt[1] * scale.y;
        dL_dMt_scaled[2] = dL_dMt[2] * scale.z;
        const mat3& D = dL_dMt_scaled;
        const f32 dL_drot_x = 2.0f * z * (D[0][1] - D[1][0]) + 2.0f * y * (D[2][0] - D[0][2]) + 2.0f * x * (D[1][2] - D[2][1]);
        const f32 dL_drot_y = 2.0f * y * (D[1][0] + D[0][1]) + 2.0f * z * (D[2][0] + D[0][2]) + 2.0f * r * (D[1][2] - D[2][1]) - 4.0f * x * (D[2][2] + D[1][1]);
        const f32 dL_drot_z = 2.0f * x * (D[1][0] + D[0][1]) + 2.0f * r * (D[2][0] - D[0][2]) + 2.0f * z * (D[1][2] + D[2][1]) - 4.0f * y * (D[2][2] + D[0][0]);
        const f32 dL_drot_w = 2.0f * r * (D[0][1] - D[1][0]) + 2.0f * x * (D[2][0] + D[0][2]) + 2.0f * y * (D[1][2] + D[2][1]) - 4.0f * z * (D[1][1] + D[0][0]);

        const vec3 final_dL_dmean3D = dL_dmean3D_proj + dL_dmean3D_cov;
        const f32 dL_dopacity_raw = dL_dopac * opacity_sigmoid * (1.0f - opacity_sigmoid);
        vec3 dL_dlog_scale = dL_dscale * scale;
        {
            const f32 cap_px = settings.max_splat_radius_px;
            if (cap_px > 0.0f) {
                const f32 denom_cap = a * c - b * b;
                if (denom_cap > 0.0f) {
                    const f32 conic_x = c / denom_cap, conic_y = -b / denom_cap, conic_z = a / denom_cap;
                    const f32 disc = conic_y * conic_y - conic_x * conic_z;
                    const f32 opacity_threshold = 128.0f;
                    const f32 t_cap = 2.0f * wd_log(opacity_sigmoid * opacity_threshold);
                    if (t_cap > 0.0f && disc < 0.0f) {
                        const f32 x_extent = wd_sqrt(t_cap * conic_z / (-disc));
                        const f32 y_extent = wd_sqrt(t_cap * conic_x / (-disc));
                        if (wmax(x_extent, y_extent) >= cap_px) dL_dlog_scale = max(dL_dlog_scale, V3(0.0f));
                    }
                }
            }
        }
        u32* o = gradients + (size_t)idx * 8;
        o[0] = pack2x16float(V2(final_dL_dmean3D.x, final_dL_dmean3D.y));
        o[1] = pack2x16float(V2(final_dL_dmean3D.z, dL_dopacity_raw));
        o[2] = pack2x16float(V2(dL_drot_x, dL_drot_y));
        o[3] = pack2x16float(V2(dL_drot_z, dL_drot_w));
        o[4] = pack2x16float(V2(dL_dlog_scale.x, dL_dlog_scale.y));
        o[5] = pack2x16float(V2(dL_dlog_scale.z, 0.0f));
        const f32 dL_dcr = from_fixed(grad_colors[(size_t)idx * 3u + 0u]);
        const f32 dL_dcg = from_fixed(grad_colors[(size_t)idx * 3u + 1u]);
        const f32 dL_dcb = from_fixed(grad_colors[(size_t)idx * 3u + 2u]);
        o[6] = pack2x16float(V2(dL_dcr, dL_dcg));
        o[7] = pack2x16float(V2(dL_dcb, 0.0f));
    }
}

// adam.wgsl:53-65
static inline vec3 adam_step(const f32* cfg, f32 param, f32 grad, f32 m, f32 v, f32 lr) {
    const f32 beta1 = cfg[5], beta2 = cfg[6], eps = cfg[7];
    const f32 m_new = beta1 * m + (1.0f - beta1) * grad;
    const f32 v_new = beta2 * v + (1.0f - beta2) * grad * grad;
    const f32 step = -lr * m_new / (wd_sqrt(v_new) + eps);
    const f32 param_new = param + step;
    return V3(param_new, m_new, v_new);
}

// K18 adam.wgsl:67-175.  cfg = {lr_pos, lr_color, lr_opacity, lr_scale, lr_rot, beta1, beta2, epsilon}
// (adam.wgsl:6-16; `iteration` is uploaded but never read).
// State layouts (optimizer.ts:7-11): opt_pos/rot/scale = OptVec4{param,m,v : vec4f} = 12 f32 each;
// opt_opacity = OptFloat{param,m,v} = 3 f32; param_sh 48 f32; state_sh 48 x (m,v).
// The body of adam.wgsl:78-174 for one Gaussian, with its 14 gradient scalars already unpacked to f32.
static inline void adam_one(u32 idx, const f32* cfg, vec3 grad_pos, f32 grad_opac, vec4 grad_rot, vec3 grad_scale, vec3 grad_color, f32* opt_pos,
                            f32* opt_rot, f32* opt_scale, f32* opt_opacity, f32* param_sh, f32* state_sh) {
    const f32 lr_pos = cfg[0], lr_color = cfg[1], lr_opacity = cfg[2], lr_scale = cfg[3], lr_rot = cfg[4];
    {
        f32* P = opt_pos + (size_t)idx * 12;
        const vec3 rx = adam_step(cfg, P[0], grad_pos.x, P[4], P[8], lr_pos);
        const vec3 ry = adam_step(cfg, P[1], grad_pos.y, P[5], P[9], lr_pos);
        const vec3 rz = adam_step(cfg, P[2], grad_pos.z, P[6], P[10], lr_pos);
        P[0] = rx.x; P[1] = ry.x; P[2] = rz.x; P[3] = 1.0f;
        P[4] = rx.y; P[5] = ry.y; P[6] = rz.y; P[7] = 0.0f;
        P[8] = rx.z; P[9] = ry.z; P[10] = rz.z; P[11] = 0.0f;
    }
    {
        f32* P = opt_rot + (size_t)idx * 12;
        const vec3 rx = adam_step(cfg, P[0], grad_rot.x, P[4], P[8], lr_rot);
        const vec3 ry = adam_step(cfg, P[1], grad_rot.y, P[5], P[9], lr_rot);
        const vec3 rz = adam_step(cfg, P[2], grad_rot.z, P[6], P[10], lr_rot);
        const vec3 rw = adam_step(cfg, P[3], grad_rot.w, P[7], P[11], lr_rot);
        const vec4 new_rot = normalize(V4(rx.x, ry.x, rz.x, rw.x));
        P[0] = new_rot.x; P[1] = new_rot.y; P[2] = new_rot.z; P[3] = new_rot.w;
        P[4] = rx.y; P[5] = ry.y; P[6] = rz.y; P[7] = rw.y;
        P[8] = rx.z; P[9] = ry.z; P[10] = rz.z; P[11] = rw.z;
    }
    {
        f32* P = opt_scale + (size_t)idx * 12;
        const vec3 rx = adam_step(cfg, P[0], grad_scale.x, P[4], P[8], lr_scale);
        const vec3 ry = adam_step(cfg, P[1], grad_scale.y, P[5], P[9], lr_scale);
        const vec3 rz = adam_step(cfg, P[2], grad_scale.z, P[6], P[10], lr_scale);
        P[0] = rx.x; P[1] = ry.x; P[2] = rz.x; P[3] = 0.0f;
        P[4] = rx.y; P[5] = ry.y; P[6] = rz.y; P[7] = 0.0f;
        P[8] = rx.z; P[9] = ry.z; P[10] = rz.z; P[11] = 0.0f;
    }
    {
        f32* P = opt_opacity + (size_t)idx * 3;
        const vec3 res = adam_step(cfg, P[0], grad_opac, P[1], P[2], lr_opacity);
        P[0] = res.x; P[1] = res.y; P[2] = res.z;
    }
    for (u32 c = 0; c < 3u; c++) {
        const size_t sh_idx = (size_t)idx * 48u + c;
        const vec3 res = adam_step(cfg, param_sh[sh_idx], grad_color[c], state_sh[sh_idx * 2], state_sh[sh_idx * 2 + 1], lr_color);
        param_sh[sh_idx] = res.x;
        state_sh[sh_idx * 2] = res.y;
        state_sh[sh_idx * 2 + 1] = res.z;
    }
}

void orc_adam(u32 n, const f32* cfg, const u32* tile_counts, const u32* gradients, f32* opt_pos, f32* opt_rot, f32* opt_scale,
              f32* opt_opacity, f32* param_sh, f32* state_sh) {
#pragma omp parallel for schedule(static)
    for (u32 idx = 0; idx < n; idx++) {
        if (tile_counts[idx] == 0u) continue;
        const u32* gp = gradients + (size_t)idx * 8;
        const vec2 g_pos_xy = unpack2x16float(gp[0]), g_pos_z_op = unpack2x16float(gp[1]);
        const vec2 g_rot_xy = unpack2x16float(gp[2]), g_rot_zw = unpack2x16float(gp[3]);
        const vec2 g_scale_xy = unpack2x16float(gp[4]), g_scale_z_ = unpack2x16float(gp[5]);
        const vec2 g_col_rg = unpack2x16float(gp[6]), g_col_b_ = unpack2x16float(gp[7]);
        adam_one(idx, cfg, V3(g_pos_xy.x, g_pos_xy.y, g_pos_z_op.x), g_pos_z_op.y, V4(g_rot_xy.x, g_rot_xy.y, g_rot_zw.x, g_rot_zw.y),
                 V3(g_scale_xy.x, g_scale_xy.y, g_scale_z_.x), V3(g_col_rg.x, g_col_rg.y, g_col_b_.x), opt_pos, opt_rot, opt_scale, opt_opacity, param_sh,
                 state_sh);
    }
}

// The view-batched step this repo adds on top of the reference (no counterpart there: trainer.ts:573 trains one view per step):
// the same Adam, fed the fp32 SUM over the batch's views of the per-view GaussianGradients (14 scalars per Gaussian in
// GaussianGradient component order: pos3, opacity, rot4, log-sigma3, rgb3) and run where the Gaussian touched a tile in at least
// one view.  With one view it is orc_adam (fp16 -> f32 unpacking is exact).
void orc_adam_f32(u32 n, const f32* cfg, const u32* visible_counts, const f32* grad_f32, f32* opt_pos, f32* opt_rot, f32* opt_scale,
                  f32* opt_opacity, f32* param_sh, f32* state_sh) {
#pragma omp parallel for schedule(static)
    for (u32 idx = 0; idx < n; idx++) {
        if (visible_counts[idx] == 0u) continue;
        const f32* g = grad_f32 + (size_t)idx * 14;
        adam_one(idx, cfg, V3(g[0], g[1], g[2]), g[3], V4(g[4], g[5], g[6], g[7]), V3(g[8], g[9], g[10]), V3(g[11], g[12], g[13]), opt_pos, opt_rot,
                 opt_scale, opt_opacity, param_sh, state_sh);
    }
}

// K19 update-gaussians.wgsl:35-77
void orc_repack(u32 n, const f32* opt_pos, const f32* opt_rot, const f32* opt_scale, const f32* opt_opacity, const f32* param_sh,
                u32* gaussians, u32* sh_buffer) {
#pragma omp parallel for schedule(static)
    for (u32 idx = 0; idx < n; idx++) {
        const f32* p = opt_pos + (size_t)idx * 12;
        const f32 op = opt_opacity[(size_t)idx * 3];
        u32* g = gaussians + (size_t)idx * 6;
        g[0] = pack2x16float(V2(p[0], p[1]));
        g[1] = pack2x16float(V2(p[2], op));
        const f32* r = opt_rot + (size_t)idx * 12;
        g[2] = pack2x16float(V2(r[0], r[1]));
        g[3] = pack2x16float(V2(r[2], r[3]));
        const f32* s = opt_scale + (size_t)idx * 12;
        g[4] = pack2x16float(V2(s[0], s[1]));
        g[5] = pack2x16float(V2(s[2], 0.0f));
        const size_t base_word = (size_t)idx * 24u;
        const f32 c0 = param_sh[(size_t)idx * 48u + 0u], c1 = param_sh[(size_t)idx * 48u + 1u];
        sh_buffer[base_word + 0u] = pack2x16float(V2(c0, c1));
        const f32 c2 = param_sh[(size_t)idx * 48u + 2u];
        const vec2 unpacked_1 = unpack2x16float(sh_buffer[base_word + 1u]);
        sh_buffer[base_word + 1u] = pack2x16float(V2(c2, unpacked_1.y));
    }
}

// K20 optimizer.ts:166-223: fp16 point cloud -> fp32 master parameters; m, v untouched (zero-filled buffers).
void orc_unpack(u32 n, const u32* gaussians, const u32* sh_buffer, f32* opt_pos, f32* opt_rot, f32* opt_scale, f32* opt_opacity,
                f32* param_sh) {
#pragma omp parallel for schedule(static)
    for (u32 idx = 0; idx < n; idx++) {
        const u32* g = gaussians + (size_t)idx * 6;
        const vec2 po0 = unpack2x16float(g[0]), po1 = unpack2x16float(g[1]);
        f32* P = opt_pos + (size_t)idx * 12;
        P[0] = po0.x; P[1] = po0.y; P[2] = po1.x; P[3] = 1.0f;
        opt_opacity[(size_t)idx * 3] = po1.y;
        const vec2 r0 = unpack2x16float(g[2]), r1 = unpack2x16float(g[3]);
        f32* R = opt_rot + (size_t)idx * 12;
        R[0] = r0.x; R[1] = r0.y; R[2] = r1.x; R[3] = r1.y;
        const vec2 s0 = unpack2x16float(g[4]), s1 = unpack2x16float(g[5]);
        f32* S = opt_scale + (size_t)idx * 12;
        S[0] = s0.x; S[1] = s0.y; S[2] = s1.x; S[3] = 0.0f;
        for (u32 i = 0; i < 24u; i++) {
            const vec2 pair = unpack2x16float(sh_buffer[(size_t)idx * 24u + i]);
            param_sh[(size_t)idx * 48u + i * 2u] = pair.x;
            param_sh[(size_t)idx * 48u + i * 2u + 1u] = pair.y;
        }
    }
}

}  // extern "C"
