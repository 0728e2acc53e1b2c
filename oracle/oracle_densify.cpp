// ORACLE (test infrastructure only; PARITY UNPINNED -- see wgsl_shim.hpp).
//
// CPU restatement of the densify/prune path.  Citations relative to /root/reference/src.
//
//   K31 bilinear down-sample     shaders/blit.wgsl:27-31 via trainer.ts:303-328 (linear sampler, clamp-to-edge)
//   K21 metric_error_main        shaders/metric-map.wgsl:27-44
//   K22 metric_reduce_minmax     shaders/metric-map.wgsl:52-82 (semantics: global min / max)
//   K23 metric_threshold_main    shaders/metric-map.wgsl:96-117
//   K24 metric_count_main        shaders/metric-count.wgsl:20-88
//   K25 metric_normalize_main    shaders/metric-normalize.wgsl:18-28
//   K26 decide_main              shaders/densify-prune-decide.wgsl:42-89
//   K27 cap_main                 shaders/densify-prune-cap.wgsl:22-49
//   K28 total_main               shaders/densify-prune-total.wgsl:21-34
//   K29 scatter_gaussians_main   shaders/densify-prune-scatter-gaussians.wgsl:79-182
//   K30 scatter_opt_*            shaders/densify-prune-scatter-opt-{pos,vec4,scale,float,sh}.wgsl
#include "wgsl_shim.hpp"

using namespace wgsl;

namespace {

struct RenderSettings { f32 gaussian_scaling, sh_deg, viewport_x, viewport_y, point_size_px, gaussian_mode, max_splat_radius_px; };

static inline f32 sigmoid(f32 x) { return 1.0f / (1.0f + wd_exp(-x)); }

// densify-prune-scatter-gaussians.wgsl:30-77 (identical copies in scatter-opt-pos.wgsl:25-65)
static inline u32 hash_u32(u32 x) {
    u32 v = x;
    v = v ^ (v >> 16u);
    v = v * 0x7feb352du;
    v = v ^ (v >> 15u);
    v = v * 0x846ca68bu;
    v = v ^ (v >> 16u);
    return v;
}
static inline f32 rand01(u32 seed) { return (f32)hash_u32(seed) * (1.0f / 4294967296.0f); }
static inline vec3 clamp_log_scale(vec3 ls) { return clamp(ls, V3(-10.0f), V3(10.0f)); }
static inline vec4 quat_normalize(vec4 q) {
    const f32 len2 = wmax(1e-12f, dot(q, q));
    return q * wd_inverseSqrt(len2);
}
static inline vec3 quat_rotate(vec4 q_in, vec3 v) {
    const vec4 q = quat_normalize(q_in);
    const vec3 u = V3(q.y, q.z, q.w);
    const f32 s = q.x;
    return 2.0f * dot(u, v) * u + (s * s - dot(u, u)) * v + 2.0f * s * cross(u, v);
}
static inline f32 randn_approx(u32 seed) {
    f32 s = 0.0f;
    s = s + rand01(seed ^ 0xA2C79u);
    s = s + rand01(seed ^ 0x5E2D9u);
    s = s + rand01(seed ^ 0x1B873u);
    s = s + rand01(seed ^ 0xC0FFEu);
    s = s + rand01(seed ^ 0xBADC0u);
    s = s + rand01(seed ^ 0xDEADBu);
    return (s - 3.0f) * 1.41421356237f;
}

const f32 LN_1P6 = 0.4700036292457356f;
const f32 OPACITY_MAX = 0.8f;
const f32 OPACITY_MAX_RAW = 1.38629436112f;

}  // namespace

extern "C" {

// K31: the metric pass renders at (W/s, H/s) and compares against the GT drawn into an rgba8unorm target of that
// size with a linear sampler (trainer.ts:129, 303-328).  Sample position = destination texel centre in source
// texel space; weights in binary32; clamp-to-edge; unorm store as in K14.
void orc_downsample_bilinear(u32 srcW, u32 srcH, const uint8_t* src, u32 dstW, u32 dstH, uint8_t* dst) {
#pragma omp parallel for schedule(static)
    for (u32 y = 0; y < dstH; y++)
        for (u32 x = 0; x < dstW; x++) {
            const f32 u = ((f32)x + 0.5f) / (f32)dstW * (f32)srcW - 0.5f;
            const f32 v = ((f32)y + 0.5f) / (f32)dstH * (f32)srcH - 0.5f;
            const f32 fu = std::floor(u), fv = std::floor(v);
            const f32 wu = u - fu, wv = v - fv;
            i32 x0 = (i32)fu, y0 = (i32)fv, x1 = x0 + 1, y1 = y0 + 1;
            auto cl = [](i32 a, i32 hi) { return a < 0 ? 0 : (a > hi ? hi : a); };
            x0 = cl(x0, (i32)srcW - 1); x1 = cl(x1, (i32)srcW - 1);
            y0 = cl(y0, (i32)srcH - 1); y1 = cl(y1, (i32)srcH - 1);
            for (u32 c = 0; c < 4u; c++) {
                const f32 t00 = (f32)src[((size_t)y0 * srcW + x0) * 4 + c] / 255.0f;
                const f32 t10 = (f32)src[((size_t)y0 * srcW + x1) * 4 + c] / 255.0f;
                const f32 t01 = (f32)src[((size_t)y1 * srcW + x0) * 4 + c] / 255.0f;
                const f32 t11 = (f32)src[((size_t)y1 * srcW + x1) * 4 + c] / 255.0f;
                const f32 top = t00 + (t10 - t00) * wu;
                const f32 bot = t01 + (t11 - t01) * wu;
                const f32 val = top + (bot - top) * wv;
                dst[((size_t)y * dstW + x) * 4 + c] = (uint8_t)to_u32(clamp(val, 0.0f, 1.0f) * 255.0f + 0.5f);
            }
        }
}

// K21 + K22 + K23 (metric-map.wgsl).  err_scale = 1e6 (tiled-backward-pass.ts:436).  Outputs the per-pixel
// error (u32), the global (min,max) pair and the r32uint flag map.
void orc_metric_map(u32 W, u32 H, const uint8_t* pred, const uint8_t* targ, f32 err_scale, f32 threshold, u32* err_out, u32* minmax,
                    u32* flags) {
    u32 mn = 0xFFFFFFFFu, mx = 0u;
    for (u32 y = 0; y < H; y++)
        for (u32 x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            const vec3 pr = V3((f32)pred[p * 4] / 255.0f, (f32)pred[p * 4 + 1] / 255.0f, (f32)pred[p * 4 + 2] / 255.0f);
            const vec3 tg = V3((f32)targ[p * 4] / 255.0f, (f32)targ[p * 4 + 1] / 255.0f, (f32)targ[p * 4 + 2] / 255.0f);
            const vec3 diff = abs(pr - tg);
            const f32 l1 = (diff.x + diff.y + diff.z) / 3.0f;
            const f32 scaled = l1 * err_scale;
            const u32 v = to_u32(clamp(scaled, 0.0f, 4294967295.0f));
            err_out[p] = v;
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
    minmax[0] = mn; minmax[1] = mx;
    for (size_t p = 0; p < (size_t)W * H; p++) {
        f32 norm = 0.0f;
        if (mx > mn) norm = ((f32)(err_out[p] - mn)) / ((f32)(mx - mn));
        flags[p] = (norm > threshold) ? 1u : 0u;
    }
}

// K24 metric-count.wgsl:20-88 (no extent test: SURVEY Q21).  Power uses the contraction pinned in K14.
void orc_metric_count(const f32* settings_f, const u32* tile_offsets, const u32* tile_instances, u32 num_instances, const u32* splats,
                      u32 num_splats, const u32* metric_map, const u32* n_contrib_tex, u32* metric_counts, u32 num_counts) {
    RenderSettings settings;
    std::memcpy(&settings, settings_f, sizeof(settings));
    const u32 width = to_u32(settings.viewport_x), height = to_u32(settings.viewport_y);
    const u32 num_tiles_x = (width + 15u) / 16u;
    const vec2 viewport = V2(settings.viewport_x, settings.viewport_y);
    for (u32 py = 0; py < height; py++)
        for (u32 px = 0; px < width; px++) {
            const size_t p = (size_t)py * width + px;
            if (metric_map[p] == 0u) continue;
            const u32 n_contrib = n_contrib_tex[p];
            if (n_contrib == 0u) continue;
            const u32 tile_idx = (py / 16u) * num_tiles_x + (px / 16u);
            const u32 start = tile_offsets[tile_idx];
            if (start == 0xFFFFFFFFu) continue;
            const vec2 pixf = V2((f32)px, (f32)py) + 0.5f;
            for (u32 i = 0; i < n_contrib; i++) {
                const u32 entry = start + i;
                if (entry >= num_instances) break;
                const u32 gidx = tile_instances[entry];
                if (gidx >= num_splats || gidx >= num_counts) continue;
                const u32* s = splats + (size_t)gidx * 6;
                const vec2 pos_ndc = unpack2x16float(s[0]);
                const vec2 conic_xy = unpack2x16float(s[2]);
                const vec2 conic_z = unpack2x16float(s[3]);
                const vec2 color_ba = unpack2x16float(s[5]);
                const vec2 center_px = (pos_ndc * V2(0.5f, -0.5f) + 0.5f) * viewport;
                const vec3 conic = V3(conic_xy.x, conic_xy.y, conic_z.x);
                const f32 opacity = color_ba.y;
                const vec2 delta = pixf - center_px;
                const f32 t1 = std::fmaf(conic.x, delta.x, (2.0f * conic.y) * delta.y);
                const f32 power = std::fmaf(t1, delta.x, (conic.z * delta.y) * delta.y);
                const f32 G = wd_exp(-0.5f * power);
                const f32 alpha = wmin(0.99f, opacity * G);
                if (alpha < (1.0f / 255.0f)) continue;
                metric_counts[gidx] += 1u;
            }
        }
}

// K25 metric-normalize.wgsl:18-28
void orc_metric_normalize(u32 n, u32 divisor, u32* metric_counts) {
    const u32 d = divisor > 1u ? divisor : 1u;
    for (u32 i = 0; i < n; i++) metric_counts[i] = metric_counts[i] / d;
}

// K26 densify-prune-decide.wgsl:42-89
void orc_densify_decide(u32 n, const u32* gaussians, const u32* metric_counts, u32 clone_threshold_count, f32 prune_opacity,
                        f32 split_scale_threshold, u32* out_counts, u32* out_actions) {
    for (u32 idx = 0; idx < n; idx++) {
        const u32* g = gaussians + (size_t)idx * 6;
        const vec2 pos_1 = unpack2x16float(g[1]);
        const f32 opacity = sigmoid(pos_1.y);
        const u32 count = metric_counts ? metric_counts[idx] : 0u;
        u32 action = 0u, out_count = 1u;
        if (opacity < prune_opacity) {
            action = 3u; out_count = 0u;
        } else if (count >= clone_threshold_count) {
            const vec2 s0 = unpack2x16float(g[4]), s1 = unpack2x16float(g[5]);
            const vec3 scale3 = exp(V3(s0.x, s0.y, s1.x));
            const f32 max_scale = wmax(scale3.x, wmax(scale3.y, scale3.z));
            action = (max_scale >= split_scale_threshold) ? 2u : 1u;
            out_count = 2u;
        }
        out_counts[idx] = out_count;
        out_actions[idx] = action;
    }
}

// K27 densify-prune-cap.wgsl:22-49
void orc_densify_cap(u32 n, u32 max_out, const u32* out_offsets, u32* out_counts, u32* out_actions) {
    for (u32 idx = 0; idx < n; idx++) {
        const u32 off = out_offsets[idx], c = out_counts[idx];
        if (max_out == 0u) { out_counts[idx] = 0u; out_actions[idx] = 3u; continue; }
        if (off >= max_out) { out_counts[idx] = 0u; out_actions[idx] = 3u; continue; }
        if (c == 2u && off == max_out - 1u) { out_counts[idx] = 1u; out_actions[idx] = 0u; }
    }
}

// K28 densify-prune-total.wgsl:21-34
u32 orc_densify_total(u32 n, const u32* prefix, const u32* out_counts) { return n == 0u ? 0u : prefix[n - 1] + out_counts[n - 1]; }

// K29 densify-prune-scatter-gaussians.wgsl:79-182
void orc_scatter_gaussians(u32 in_points, u32 out_points, const u32* in_gaussians, const u32* in_sh, const u32* out_offsets,
                           const u32* out_counts, const u32* out_actions, u32* out_gaussians, u32* out_sh) {
    auto copy_point = [&](u32 dst_idx, u32 src_idx, u32 variant, u32 action) {
        u32 g[6];
        std::memcpy(g, in_gaussians + (size_t)src_idx * 6, 24);
        const vec2 p0 = unpack2x16float(g[0]), p1 = unpack2x16float(g[1]);
        const f32 op = sigmoid(p1.y);
        const bool opacity_clamped = op > OPACITY_MAX;
        const f32 opacity_raw = opacity_clamped ? OPACITY_MAX_RAW : p1.y;
        const vec2 r0 = unpack2x16float(g[2]), r1 = unpack2x16float(g[3]);
        const vec4 q = V4(r0.x, r0.y, r1.x, r1.y);
        const vec2 s0 = unpack2x16float(g[4]), s1 = unpack2x16float(g[5]);
        const vec3 log_sigma = clamp_log_scale(V3(s0.x, s0.y, s1.x));
        const vec3 sigma = exp(log_sigma);
        vec3 pos = V3(p0.x, p0.y, p1.x);
        const bool needs_transform = (action == 2u) || (action == 1u && variant == 1u);
        if (!(!needs_transform && !opacity_clamped)) {
            if (action == 1u && variant == 1u) {
                const u32 seed = src_idx * 1664525u + dst_idx * 1013904223u;
                const vec3 r = V3(rand01(seed ^ 0xA2C79u), rand01(seed ^ 0x5E2D9u), rand01(seed ^ 0x1B873u)) * 2.0f - 1.0f;
                const vec3 jitter_local = 0.25f * sigma * r;
                pos = pos + quat_rotate(q, jitter_local);
            }
            if (action == 2u) {
                const u32 seed = src_idx * 747796405u + 2891336453u;
                const vec3 d = V3(randn_approx(seed ^ 0x9E3779B9u), randn_approx(seed ^ 0x243F6A88u), randn_approx(seed ^ 0xB7E15162u));
                const vec3 offset_local = 0.5f * sigma * d;
                const f32 sgn = (variant == 1u) ? -1.0f : 1.0f;
                pos = pos + sgn * quat_rotate(q, offset_local);
                const vec3 log_child = log_sigma - V3(LN_1P6);
                g[4] = pack2x16float(V2(log_child.x, log_child.y));
                g[5] = pack2x16float(V2(log_child.z, 0.0f));
            }
            g[0] = pack2x16float(V2(pos.x, pos.y));
            g[1] = pack2x16float(V2(pos.z, opacity_raw));
        }
        std::memcpy(out_gaussians + (size_t)dst_idx * 6, g, 24);
        std::memcpy(out_sh + (size_t)dst_idx * 24, in_sh + (size_t)src_idx * 24, 96);
    };
    for (u32 idx = 0; idx < in_points; idx++) {
        const u32 act = out_actions[idx];
        const u32 c = out_counts[idx];
        if (c == 0u) continue;
        const u32 off = out_offsets[idx];
        if (off >= out_points) continue;
        copy_point(off, idx, 0u, act);
        if (c == 2u) {
            const u32 off1 = off + 1u;
            if (off1 < out_points) copy_point(off1, idx, 1u, act);
        }
    }
}

// K30: the five optimizer-state scatters, fused in one walk.  reset_new_state as ScatterInfo.reset_new_state.
//   pos    scatter-opt-pos.wgsl:66-137   (perturbation from the fp32 masters: SURVEY Q17)
//   rot    scatter-opt-vec4.wgsl:22-61
//   scale  scatter-opt-scale.wgsl:24-69
//   opac.  scatter-opt-float.wgsl:29-63   (clamp + m,v always zero: SURVEY Q16)
//   sh     scatter-opt-sh.wgsl:21-57
void orc_scatter_optimizer(u32 in_points, u32 out_points, u32 reset_new_state, const u32* out_offsets, const u32* out_counts,
                           const u32* out_actions, const f32* in_pos, const f32* in_rot, const f32* in_scale, const f32* in_opacity,
                           const f32* in_param_sh, const f32* in_state_sh, f32* out_pos, f32* out_rot, f32* out_scale, f32* out_opacity,
                           f32* out_param_sh, f32* out_state_sh) {
    auto write_slot = [&](u32 dst_idx, u32 src_idx, u32 variant, u32 action) {
        const bool is_new = (variant == 1u) || (action == 2u);
        const bool reset = (reset_new_state != 0u) && is_new;
        // --- pos
        {
            const f32* src = in_pos + (size_t)src_idx * 12;
            vec4 p = V4(src[0], src[1], src[2], src[3]);
            const f32* qp = in_rot + (size_t)src_idx * 12;
            const vec4 q = V4(qp[0], qp[1], qp[2], qp[3]);
            const f32* sp = in_scale + (size_t)src_idx * 12;
            const vec3 log_sigma = clamp_log_scale(V3(sp[0], sp[1], sp[2]));
            const vec3 sigma = exp(log_sigma);
            if (action == 1u && variant == 1u) {
                const u32 seed = src_idx * 1664525u + dst_idx * 1013904223u;
                const vec3 r = V3(rand01(seed ^ 0xA2C79u), rand01(seed ^ 0x5E2D9u), rand01(seed ^ 0x1B873u)) * 2.0f - 1.0f;
                const vec3 jitter_local = 0.25f * sigma * r;
                p = V4(p.xyz() + quat_rotate(q, jitter_local), p.w);
            } else if (action == 2u) {
                const u32 seed = src_idx * 747796405u + 2891336453u;
                const vec3 d = V3(randn_approx(seed ^ 0x9E3779B9u), randn_approx(seed ^ 0x243F6A88u), randn_approx(seed ^ 0xB7E15162u));
                const vec3 offset_local = 0.5f * sigma * d;
                const f32 sgn = (variant == 1u) ? -1.0f : 1.0f;
                p = V4(p.xyz() + sgn * quat_rotate(q, offset_local), p.w);
            }
            f32* dst = out_pos + (size_t)dst_idx * 12;
            dst[0] = p.x; dst[1] = p.y; dst[2] = p.z; dst[3] = p.w;
            for (int k = 4; k < 12; k++) dst[k] = reset ? 0.0f : src[k];
        }
        // --- rot (opt-vec4)
        {
            const f32* src = in_rot + (size_t)src_idx * 12;
            f32* dst = out_rot + (size_t)dst_idx * 12;
            for (int k = 0; k < 4; k++) dst[k] = src[k];
            for (int k = 4; k < 12; k++) dst[k] = reset ? 0.0f : src[k];
        }
        // --- scale
        {
            const f32* src = in_scale + (size_t)src_idx * 12;
            f32* dst = out_scale + (size_t)dst_idx * 12;
            vec4 p = V4(src[0], src[1], src[2], src[3]);
            if (action == 2u) p = V4(p.xyz() - V3(LN_1P6), p.w);
            dst[0] = p.x; dst[1] = p.y; dst[2] = p.z; dst[3] = p.w;
            for (int k = 4; k < 12; k++) dst[k] = reset ? 0.0f : src[k];
        }
        // --- opacity (opt-float): clamp in sigmoid space, always reset m and v
        {
            const f32 raw = in_opacity[(size_t)src_idx * 3];
            const f32 op = sigmoid(raw);
            f32* dst = out_opacity + (size_t)dst_idx * 3;
            dst[0] = (op > OPACITY_MAX) ? OPACITY_MAX_RAW : raw;
            dst[1] = 0.0f; dst[2] = 0.0f;
        }
        // --- sh
        {
            const size_t sb = (size_t)src_idx * 48, db = (size_t)dst_idx * 48;
            for (u32 i = 0; i < 48u; i++) {
                out_param_sh[db + i] = in_param_sh[sb + i];
                out_state_sh[(db + i) * 2] = reset ? 0.0f : in_state_sh[(sb + i) * 2];
                out_state_sh[(db + i) * 2 + 1] = reset ? 0.0f : in_state_sh[(sb + i) * 2 + 1];
            }
        }
    };
    for (u32 idx = 0; idx < in_points; idx++) {
        const u32 action = out_actions[idx];
        const u32 c = out_counts[idx];
        if (c == 0u) continue;
        const u32 off = out_offsets[idx];
        if (off >= out_points) continue;
        write_slot(off, idx, 0u, action);
        if (c == 2u) {
            const u32 off1 = off + 1u;
            if (off1 < out_points) write_slot(off1, idx, 1u, action);
        }
    }
}

}  // extern "C"
