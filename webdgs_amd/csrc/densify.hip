// Densify / prune path: metric map (K21-K23), per-Gaussian metric counts (K24-K25), bilinear GT down-sample (K31),
// decide / cap / total (K26-K28) and the fused rebuild scatter (K29 + the five K30 kernels).
//
// Replaces src/shaders/metric-map.wgsl, metric-count.wgsl, metric-normalize.wgsl, blit.wgsl (fs_main as used by
// trainer.ts:303-328), densify-prune-decide/cap/total.wgsl, densify-prune-scatter-gaussians.wgsl and
// densify-prune-scatter-opt-{pos,vec4,scale,float,sh}.wgsl, plus renderers/densify-prune.ts.
// All kernels are HBM-bound streaming passes; the reference's log256(P)+2 reduction launches become one pass with
// order-free u32 atomicMin/atomicMax, and its six scatter launches (each re-reading offsets/counts/actions) become one.
#include <algorithm>
#include <cstring>

#include "common.h"
#include "wgslm.h"
#include "blockcull.h"

namespace {

// ------------------------------------------------------------------ K31
__global__ __launch_bounds__(256) void downsample_kernel(const u32* __restrict__ src, u32 sw, u32 sh, u32* __restrict__ dst, u32 dw, u32 dh) {
    const u32 x = blockIdx.x * 16u + (threadIdx.x & 15u), y = blockIdx.y * 16u + (threadIdx.x >> 4);
    if (x >= dw || y >= dh) return;
    const float u = wd_div((float)x + 0.5f, (float)dw) * (float)sw - 0.5f;
    const float v = wd_div((float)y + 0.5f, (float)dh) * (float)sh - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float wu = u - fu, wv = v - fv;
    int x0 = (int)fu, y0 = (int)fv, x1 = x0 + 1, y1 = y0 + 1;
    x0 = min(max(x0, 0), (int)sw - 1); x1 = min(max(x1, 0), (int)sw - 1);
    y0 = min(max(y0, 0), (int)sh - 1); y1 = min(max(y1, 0), (int)sh - 1);
    const u32 t00 = src[(size_t)y0 * sw + x0], t10 = src[(size_t)y0 * sw + x1], t01 = src[(size_t)y1 * sw + x0], t11 = src[(size_t)y1 * sw + x1];
    u32 out = 0u;
#pragma unroll
    for (u32 c = 0; c < 4u; c++) {
        const float a = wd_div((float)((t00 >> (8u * c)) & 0xFFu), 255.0f), b = wd_div((float)((t10 >> (8u * c)) & 0xFFu), 255.0f);
        const float cc = wd_div((float)((t01 >> (8u * c)) & 0xFFu), 255.0f), d = wd_div((float)((t11 >> (8u * c)) & 0xFFu), 255.0f);
        const float top = a + (b - a) * wu;
        const float bot = cc + (d - cc) * wu;
        const float val = top + (bot - top) * wv;
        out |= wd_to_u32(wd_clamp(val, 0.0f, 1.0f) * 255.0f + 0.5f) << (8u * c);
    }
    dst[(size_t)y * dw + x] = out;
}

// ------------------------------------------------------------------ K21-K23
__global__ void metric_init_kernel(u32* __restrict__ minmax) {
    if (threadIdx.x == 0) { minmax[0] = 0xFFFFFFFFu; minmax[1] = 0u; }
}

__global__ __launch_bounds__(256) void metric_error_kernel(u32 npix, const u32* __restrict__ pred, const u32* __restrict__ targ, float err_scale,
                                                            u32* __restrict__ err, u32* __restrict__ minmax) {
    __shared__ u32 smin[4], smax[4];
    u32 mn = 0xFFFFFFFFu, mx = 0u;
    for (u32 p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const u32 a = pred[p], b = targ[p];
        const float dr = fabsf(wd_div((float)(a & 0xFFu), 255.0f) - wd_div((float)(b & 0xFFu), 255.0f));
        const float dg = fabsf(wd_div((float)((a >> 8) & 0xFFu), 255.0f) - wd_div((float)((b >> 8) & 0xFFu), 255.0f));
        const float db = fabsf(wd_div((float)((a >> 16) & 0xFFu), 255.0f) - wd_div((float)((b >> 16) & 0xFFu), 255.0f));
        const float l1 = wd_div(dr + dg + db, 3.0f);
        const u32 v = wd_to_u32(wd_clamp(l1 * err_scale, 0.0f, 4294967295.0f));
        err[p] = v;
        mn = min(mn, v);
        mx = max(mx, v);
    }
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) {
        mn = min(mn, (u32)__shfl_xor((int)mn, (int)d, 64));
        mx = max(mx, (u32)__shfl_xor((int)mx, (int)d, 64));
    }
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) { smin[wave] = mn; smax[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&minmax[0], min(min(smin[0], smin[1]), min(smin[2], smin[3])));
        atomicMax(&minmax[1], max(max(smax[0], smax[1]), max(smax[2], smax[3])));
    }
}

__global__ __launch_bounds__(256) void metric_threshold_kernel(u32 npix, const u32* __restrict__ err, const u32* __restrict__ minmax, float threshold,
                                                                u32* __restrict__ flags) {
    const u32 mn = minmax[0], mx = minmax[1];
    for (u32 p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        float norm = 0.0f;
        if (mx > mn) norm = wd_div((float)(err[p] - mn), (float)(mx - mn));
        flags[p] = (norm > threshold) ? 1u : 0u;
    }
}

// ------------------------------------------------------------------ K24-K25
// metric-count.wgsl: every flagged pixel walks its first n_contrib tile entries and adds 1 to the count of each Gaussian whose
// alpha at that pixel is >= 1/255 -- per pixel, with a global splat load per (pixel, entry) and one atomic per contributing pair.
// Here, as in backward_raster.hip: a wave owns an 8x8 block and walks the entries its flagged pixels need in chunks of 64
// (lane = entry), keeps the splats that can reach alpha >= 1/255 somewhere in the block (blockcull.h; conservative, so the counts do
// not depend on it), and for each kept splat the 64 pixels vote: ONE atomic per (wave, splat) adds the number of contributing
// pixels.  u32 sums are order-free, so the counts equal the per-pixel formulation bit for bit.
__global__ __launch_bounds__(256) void metric_count_kernel(RenderSettings settings, u32 num_tiles_x, const u32* __restrict__ ranges,
                                                            const u32* __restrict__ instances, u32 num_instances, const u32* __restrict__ splats,
                                                            u32 num_splats, const u32* __restrict__ flags, const u32* __restrict__ n_contrib,
                                                            u32* __restrict__ counts, u32 num_counts) {
    __shared__ float4 s_geo_all[4][64];  // centre.x, centre.y, opacity, gaussian index (bits)
    __shared__ float4 s_con_all[4][64];  // conic.x, 2*conic.y, conic.z, position of the entry in the tile's list (bits)
    const u32 tile_id = blockIdx.x, sub = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const u32 tile_x = tile_id % num_tiles_x, tile_y = tile_id / num_tiles_x;
    float4* const s_geo = s_geo_all[sub];  // wave-private: the four waves of a tile never synchronise
    float4* const s_con = s_con_all[sub];
    const u32 bx = tile_x * 16u + (sub & 1u) * 8u, by = tile_y * 16u + (sub >> 1) * 8u;
    const u32 px = bx + (lane & 7u), py = by + (lane >> 3);
    const float vx = settings.viewport_x, vy = settings.viewport_y;
    const u32 W = wd_to_u32(vx), H = wd_to_u32(vy);
    const u32 start = ranges[tile_id];
    u32 n_pix = 0u;
    if (px < W && py < H && start != 0xFFFFFFFFu) {
        const size_t p = (size_t)py * W + px;
        if (flags[p] != 0u) n_pix = n_contrib[p];
    }
    // entries past the end of the list are never read (metric-count.wgsl breaks out of its loop there)
    const u32 avail = (start != 0xFFFFFFFFu && start < num_instances) ? num_instances - start : 0u;
    n_pix = min(n_pix, avail);
    u32 wmax = n_pix;
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) wmax = max(wmax, (u32)__shfl_xor((int)wmax, (int)d, 64));
    if (wmax == 0u) return;
    const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
    const float blk_x0 = (float)bx + 0.5f, blk_x1 = (float)bx + 7.5f, blk_y0 = (float)by + 0.5f, blk_y1 = (float)by + 7.5f;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (u32 lo = 0; lo < wmax; lo += 64u) {
        const u32 pos = lo + lane;
        bool ok = false;
        u32 g = 0u;
        float cx = 0.f, cy = 0.f, A = 0.f, B = 0.f, Cc = 0.f, opacity = 0.f;
        if (pos < wmax) {
            g = instances[start + pos];
            if (g < num_splats && g < num_counts) {
                const uint2* sp = reinterpret_cast<const uint2*>(splats + (size_t)g * 6);
                const uint2 w01 = sp[0], w23 = sp[1], w45 = sp[2];
                cx = (wd_unpack_lo(w01.x) * 0.5f + 0.5f) * vx;
                cy = (wd_unpack_hi(w01.x) * -0.5f + 0.5f) * vy;
                A = wd_unpack_lo(w23.x); B = wd_unpack_hi(w23.x); Cc = wd_unpack_lo(w23.y);
                opacity = wd_unpack_hi(w45.y);
                ok = block_reaches_min_alpha(A, B, Cc, opacity, blk_x0 - cx, blk_x1 - cx, blk_y0 - cy, blk_y1 - cy);
            }
        }
        const unsigned long long m = __ballot(ok);
        const u32 n_list = (u32)__popcll(m);
        if (ok) {
            const u32 slot = (u32)__popcll(m & lt_mask);
            s_geo[slot] = make_float4(cx, cy, opacity, __uint_as_float(g));
            s_con[slot] = make_float4(A, 2.0f * B, Cc, __uint_as_float(pos));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (u32 i = 0; i < n_list; i++) {
            const float4 geo = s_geo[i];
            const float4 con = s_con[i];
            const float dx = pxf - geo.x, dy = pyf - geo.y;
            const float t1 = __builtin_fmaf(con.x, dx, con.y * dy);
            const float power = __builtin_fmaf(t1, dx, (con.z * dy) * dy);
            const float G = wd_exp(-0.5f * power);
            const float og = geo.z * G;
            const float alpha = (og < 0.99f) ? og : 0.99f;
            const bool act = (__float_as_uint(con.w) < n_pix) && !(alpha < (1.0f / 255.0f));
            const u32 votes = (u32)__popcll(__ballot(act));
            if (lane == 0u && votes != 0u) atomicAdd(&counts[__float_as_uint(geo.w)], votes);
        }
        __builtin_amdgcn_wave_barrier();  // the records are re-written by the next chunk
    }
}

__global__ __launch_bounds__(256) void metric_normalize_kernel(u32 n, u32 divisor, u32* __restrict__ counts) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) counts[i] = counts[i] / max(1u, divisor);
}

// ------------------------------------------------------------------ K26-K28
WD_DEV float sigmoidf(float x) { return wd_div(1.0f, 1.0f + wd_exp(-x)); }

__global__ __launch_bounds__(256) void decide_kernel(u32 n, const u32* __restrict__ gaussians, const u32* __restrict__ metric_counts, u32 clone_threshold,
                                                      float prune_opacity, float split_scale, u32* __restrict__ out_counts, u32* __restrict__ out_actions) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const u32* g = gaussians + (size_t)idx * 6;
    const float opacity = sigmoidf(wd_unpack_hi(g[1]));
    const u32 count = metric_counts ? metric_counts[idx] : 0u;
    u32 action = 0u, out_count = 1u;
    if (opacity < prune_opacity) {
        action = 3u; out_count = 0u;
    } else if (count >= clone_threshold) {
        const float sx = wd_exp(wd_unpack_lo(g[4])), sy = wd_exp(wd_unpack_hi(g[4])), sz = wd_exp(wd_unpack_lo(g[5]));
        const float max_scale = wd_max(sx, wd_max(sy, sz));
        action = (max_scale >= split_scale) ? 2u : 1u;
        out_count = 2u;
    }
    out_counts[idx] = out_count;
    out_actions[idx] = action;
}

__global__ __launch_bounds__(256) void cap_kernel(u32 n, u32 max_out, const u32* __restrict__ offsets, u32* __restrict__ counts, u32* __restrict__ actions) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const u32 off = offsets[idx], c = counts[idx];
    if (max_out == 0u || off >= max_out) { counts[idx] = 0u; actions[idx] = 3u; return; }
    if (c == 2u && off == max_out - 1u) { counts[idx] = 1u; actions[idx] = 0u; }
}

__global__ void total_kernel(u32 n, const u32* __restrict__ prefix, const u32* __restrict__ counts, u32* __restrict__ total) {
    if (blockIdx.x == 0 && threadIdx.x == 0) total[0] = (n == 0u) ? 0u : prefix[n - 1] + counts[n - 1];
}

// ------------------------------------------------------------------ K29 + K30
WD_DEV u32 hash_u32(u32 x) {
    u32 v = x;
    v = v ^ (v >> 16u); v = v * 0x7feb352du; v = v ^ (v >> 15u); v = v * 0x846ca68bu; v = v ^ (v >> 16u);
    return v;
}
WD_DEV float rand01(u32 seed) { return (float)hash_u32(seed) * (1.0f / 4294967296.0f); }
WD_DEV float randn_approx(u32 seed) {
    float s = 0.0f;
    s = s + rand01(seed ^ 0xA2C79u); s = s + rand01(seed ^ 0x5E2D9u); s = s + rand01(seed ^ 0x1B873u);
    s = s + rand01(seed ^ 0xC0FFEu); s = s + rand01(seed ^ 0xBADC0u); s = s + rand01(seed ^ 0xDEADBu);
    return (s - 3.0f) * 1.41421356237f;
}
WD_DEV vec3 quat_rotate(vec4 q_in, vec3 v) {
    const float len2 = wd_max(1e-12f, dot(q_in, q_in));
    const vec4 q = q_in * wd_div(1.0f, wd_sqrt(len2));
    const vec3 u = V3(q.y, q.z, q.w);
    const float s = q.x;
    return 2.0f * dot(u, v) * u + (s * s - dot(u, u)) * v + 2.0f * s * cross(u, v);
}
// The child's position offset in the parent's frame (clone jitter or +-split), densify-prune-scatter-gaussians.wgsl:111-134.
WD_DEV vec3 child_offset(u32 action, u32 variant, u32 src_idx, u32 dst_idx, vec4 q, vec3 sigma, bool* moved) {
    *moved = false;
    if (action == 1u && variant == 1u) {
        const u32 seed = src_idx * 1664525u + dst_idx * 1013904223u;
        const vec3 r = V3(rand01(seed ^ 0xA2C79u), rand01(seed ^ 0x5E2D9u), rand01(seed ^ 0x1B873u)) * 2.0f - 1.0f;
        *moved = true;
        return quat_rotate(q, 0.25f * sigma * r);
    }
    if (action == 2u) {
        const u32 seed = src_idx * 747796405u + 2891336453u;
        const vec3 d = V3(randn_approx(seed ^ 0x9E3779B9u), randn_approx(seed ^ 0x243F6A88u), randn_approx(seed ^ 0xB7E15162u));
        const float sgn = (variant == 1u) ? -1.0f : 1.0f;
        *moved = true;
        return sgn * quat_rotate(q, 0.5f * sigma * d);
    }
    return V3(0.0f);
}

constexpr float LN_1P6 = 0.4700036292457356f;
constexpr float OPACITY_MAX = 0.8f;
constexpr float OPACITY_MAX_RAW = 1.38629436112f;

struct ScatterArgs {
    u32 in_points, out_points, reset_new_state;
    const u32 *offsets, *counts, *actions;
    const u32 *in_gaussians, *in_sh;
    u32 *out_gaussians, *out_sh;
    bool has_state;
    const float4 *in_pos, *in_rot, *in_scale;
    const float* in_opacity;
    const float4* in_param_sh;   // 12 float4 per point
    const float4* in_state_sh;   // 24 float4 per point
    float4 *out_pos, *out_rot, *out_scale;
    float* out_opacity;
    float4* out_param_sh;
    float4* out_state_sh;
};

WD_DEV void scatter_slot(const ScatterArgs& a, u32 dst, u32 src, u32 variant, u32 action) {
    // ---- point cloud (fp16 working copy): perturbation computed from the fp16 values
    {
        const u32* gi = a.in_gaussians + (size_t)src * 6;
        uint2 w01 = *reinterpret_cast<const uint2*>(gi), w23 = *reinterpret_cast<const uint2*>(gi + 2), w45 = *reinterpret_cast<const uint2*>(gi + 4);
        const float raw = wd_unpack_hi(w01.y);
        const bool clamped = sigmoidf(raw) > OPACITY_MAX;
        const bool needs_transform = (action == 2u) || (action == 1u && variant == 1u);
        if (needs_transform || clamped) {
            const vec4 q = V4(wd_unpack_lo(w23.x), wd_unpack_hi(w23.x), wd_unpack_lo(w23.y), wd_unpack_hi(w23.y));
            const vec3 log_sigma = vclamp(V3(wd_unpack_lo(w45.x), wd_unpack_hi(w45.x), wd_unpack_lo(w45.y)), -10.0f, 10.0f);
            const vec3 sigma = vexp(log_sigma);
            vec3 pos = V3(wd_unpack_lo(w01.x), wd_unpack_hi(w01.x), wd_unpack_lo(w01.y));
            bool moved;
            const vec3 off = child_offset(action, variant, src, dst, q, sigma, &moved);
            if (moved) pos = pos + off;
            if (action == 2u) {
                const vec3 lc = log_sigma - LN_1P6;
                w45.x = wd_pack2(lc.x, lc.y);
                w45.y = wd_pack2(lc.z, 0.0f);
            }
            w01.x = wd_pack2(pos.x, pos.y);
            w01.y = wd_pack2(pos.z, clamped ? OPACITY_MAX_RAW : raw);
        }
        u32* go = a.out_gaussians + (size_t)dst * 6;
        *reinterpret_cast<uint2*>(go) = w01;
        *reinterpret_cast<uint2*>(go + 2) = w23;
        *reinterpret_cast<uint2*>(go + 4) = w45;
        // (the 96-byte SH row is copied by the wave, not by this lane: scatter_kernel)
    }
    if (!a.has_state) return;
    // ---- optimizer state (fp32 masters): perturbation computed from the fp32 values
    const bool is_new = (variant == 1u) || (action == 2u);
    const bool reset = (a.reset_new_state != 0u) && is_new;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        float4 p = a.in_pos[(size_t)src * 3];
        const float4 qf = a.in_rot[(size_t)src * 3];
        const float4 sf = a.in_scale[(size_t)src * 3];
        const vec3 sigma = vexp(vclamp(V3(sf.x, sf.y, sf.z), -10.0f, 10.0f));
        bool moved;
        const vec3 off = child_offset(action, variant, src, dst, V4(qf.x, qf.y, qf.z, qf.w), sigma, &moved);
        if (moved) { p.x = p.x + off.x; p.y = p.y + off.y; p.z = p.z + off.z; }
        a.out_pos[(size_t)dst * 3] = p;
        a.out_pos[(size_t)dst * 3 + 1] = reset ? zero4 : a.in_pos[(size_t)src * 3 + 1];
        a.out_pos[(size_t)dst * 3 + 2] = reset ? zero4 : a.in_pos[(size_t)src * 3 + 2];
        a.out_rot[(size_t)dst * 3] = qf;
        a.out_rot[(size_t)dst * 3 + 1] = reset ? zero4 : a.in_rot[(size_t)src * 3 + 1];
        a.out_rot[(size_t)dst * 3 + 2] = reset ? zero4 : a.in_rot[(size_t)src * 3 + 2];
        float4 sp = sf;
        if (action == 2u) { sp.x = sp.x - LN_1P6; sp.y = sp.y - LN_1P6; sp.z = sp.z - LN_1P6; }
        a.out_scale[(size_t)dst * 3] = sp;
        a.out_scale[(size_t)dst * 3 + 1] = reset ? zero4 : a.in_scale[(size_t)src * 3 + 1];
        a.out_scale[(size_t)dst * 3 + 2] = reset ? zero4 : a.in_scale[(size_t)src * 3 + 2];
    }
    {   // opacity: clamp in sigmoid space; m and v are zeroed for every survivor
        const float raw = a.in_opacity[(size_t)src * 3];
        a.out_opacity[(size_t)dst * 3] = (sigmoidf(raw) > OPACITY_MAX) ? OPACITY_MAX_RAW : raw;
        a.out_opacity[(size_t)dst * 3 + 1] = 0.0f;
        a.out_opacity[(size_t)dst * 3 + 2] = 0.0f;
    }
    // (param_sh 192 B and state_sh 384 B are copied by the wave: scatter_kernel)
}

// Thread per source Gaussian for the parts that need arithmetic (the 24-byte working copy, pos / rot / scale / opacity masters); the
// three wide rows -- SH 96 B, param_sh 192 B, state_sh 384 B, 79 % of the 852 bytes a Gaussian carries -- are then copied by the whole
// wave, one output slot per iteration with one 16-byte piece per lane, so every access is a contiguous run instead of 64 lanes striding
// 96..384 bytes apart (the per-lane form moved 1.7 GB at 1.5 TB/s).
__global__ __launch_bounds__(256) void scatter_kernel(ScatterArgs a) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 lane = threadIdx.x & 63u;
    u32 c = 0u, off = 0u, action = 0u;
    if (idx < a.in_points) {
        c = a.counts[idx];
        off = a.offsets[idx];
        if (off >= a.out_points) c = 0u;
        if (c != 0u) {
            action = a.actions[idx];
            if (c != 2u || !(off + 1u < a.out_points)) c = 1u;  // (counts are 0, 1 or 2: keep, clone / split)
            scatter_slot(a, off, idx, 0u, action);
            if (c == 2u) scatter_slot(a, off + 1u, idx, 1u, action);
        }
    }
    // lanes 0..5: SH row (uint4 x 6); 6..17: param_sh (float4 x 12); 18..41: state_sh (float4 x 24)
    const u32 first = idx - lane;  // the wave's first source Gaussian
    unsigned long long todo = __ballot(c != 0u);
    while (todo) {
        const u32 j = (u32)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const u32 cj = (u32)__shfl((int)c, (int)j, 64), offj = (u32)__shfl((int)off, (int)j, 64), actj = (u32)__shfl((int)action, (int)j, 64);
        const size_t src = (size_t)first + j;
        for (u32 variant = 0; variant < cj; variant++) {
            const size_t dst = (size_t)offj + variant;
            if (lane < 6u) {
                reinterpret_cast<uint4*>(a.out_sh + dst * 24)[lane] = reinterpret_cast<const uint4*>(a.in_sh + src * 24)[lane];
            } else if (a.has_state && lane < 18u) {
                a.out_param_sh[dst * 12 + (lane - 6u)] = a.in_param_sh[src * 12 + (lane - 6u)];
            } else if (a.has_state && lane < 42u) {
                const bool reset = (a.reset_new_state != 0u) && ((variant == 1u) || (actj == 2u));
                a.out_state_sh[dst * 24 + (lane - 18u)] = reset ? make_float4(0.f, 0.f, 0.f, 0.f) : a.in_state_sh[src * 24 + (lane - 18u)];
            }
        }
    }
}

}  // namespace

int launch_downsample(wdgs_device* dev, const void* src, u32 sw, u32 sh, void* dst, u32 dw, u32 dh) {
    WDGS_LAUNCH(dev, "downsample_rgba8", downsample_kernel, dim3(ceil_div(dw, 16), ceil_div(dh, 16)), dim3(256), 0, (const u32*)src, sw, sh, (u32*)dst, dw, dh);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_metric_map(wdgs_device* dev, u32 W, u32 H, const void* pred, const void* targ, float err_scale, float threshold, void* err, void* minmax,
                      void* /*scratch*/, void* flags) {
    const u32 npix = W * H;
    if (npix == 0) return WDGS_OK;
    const u32 grid = std::min<u32>(ceil_div(npix, 256), (u32)dev->num_cus * 8);
    // one workgroup per CU for the pass that ends in two atomics on the same pair of words: 2000 of them queue up behind each other
    // at the L2 (40 of the kernel's 51 us at 960x540); min / max are order-free, so the grid does not change the result
    const u32 grid_minmax = std::min<u32>(grid, (u32)dev->num_cus);
    WDGS_LAUNCH(dev, "metric_init", metric_init_kernel, dim3(1), dim3(64), 0, (u32*)minmax);
    WDGS_LAUNCH(dev, "metric_error", metric_error_kernel, dim3(grid_minmax), dim3(256), 0, npix, (const u32*)pred, (const u32*)targ, err_scale, (u32*)err, (u32*)minmax);
    WDGS_LAUNCH(dev, "metric_threshold", metric_threshold_kernel, dim3(grid), dim3(256), 0, npix, (const u32*)err, (const u32*)minmax, threshold, (u32*)flags);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_metric_count(wdgs_device* dev, const RenderSettings& st, u32 ntx, u32 nty, const void* ranges, const void* instances, u32 num_instances,
                        const void* splats, u32 num_splats, const void* flags, const void* n_contrib, void* counts, u32 num_counts) {
    if (ntx * nty == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "metric_count", metric_count_kernel, dim3(ntx * nty), dim3(256), 0, st, ntx, (const u32*)ranges, (const u32*)instances, num_instances,
                (const u32*)splats, num_splats, (const u32*)flags, (const u32*)n_contrib, (u32*)counts, num_counts);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_metric_normalize(wdgs_device* dev, u32 n, u32 divisor, void* counts) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "metric_normalize", metric_normalize_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, divisor, (u32*)counts);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

// ------------------------------------------------------------------ DensifyPrunePass (C ABI)
struct wdgs_densify_prune {
    wdgs_device* dev;
    wdgs_densify_config cfg;
    u32 capacity;  // points the work buffers are sized for
    u32 *actions, *counts, *offsets, *total;
    ScanScratch scan;
    u32 last_max_out;
};

static void densify_free(wdgs_densify_prune* op) {
    if (op->actions) wdgs_free(op->actions);
    if (op->counts) wdgs_free(op->counts);
    if (op->offsets) wdgs_free(op->offsets);
    op->actions = op->counts = op->offsets = nullptr;
    scan_scratch_destroy(&op->scan);
    op->capacity = 0;
}

extern "C" {

int wdgs_densify_prune_create(wdgs_device* dev, const wdgs_densify_config* cfg, wdgs_densify_prune** out) {
    WDGS_REQUIRE(dev && out, WDGS_E_INVALID, "wdgs_densify_prune_create: null argument");
    wdgs_densify_prune* op = new wdgs_densify_prune();
    std::memset(op, 0, sizeof(*op));
    op->dev = dev;
    if (cfg) op->cfg = *cfg;
    else op->cfg = wdgs_densify_config{1, 0, 0.f, 0.f, 0, 128ull * 1024 * 1024};  // densify-prune.ts:110-119
    int r = wdgs_alloc((void**)&op->total, 16, true, dev->stream);
    if (r != WDGS_OK) { delete op; return r; }
    *out = op;
    return WDGS_OK;
}

int wdgs_densify_prune_destroy(wdgs_densify_prune* op) {
    if (!op) return WDGS_OK;
    if (wdgs_device_alive(op->dev) && !op->dev->capturing) (void)wdgs_sync_lanes(op->dev);
    densify_free(op);
    if (op->total) wdgs_free(op->total);
    delete op;
    return WDGS_OK;
}

int wdgs_densify_prune_set_config(wdgs_densify_prune* op, const wdgs_densify_config* cfg) {
    WDGS_REQUIRE(op && cfg, WDGS_E_INVALID, "null argument");
    op->cfg = *cfg;
    return WDGS_OK;
}

int wdgs_densify_prune_ensure_size(wdgs_densify_prune* op, uint32_t n) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "null op");
    if (n <= op->capacity && op->actions) return WDGS_OK;
    (void)wdgs_sync_lanes(op->dev);
    densify_free(op);
    const size_t N = std::max(n, 1u);
    WDGS_TRY(wdgs_alloc((void**)&op->actions, N * 4, true, op->dev->stream));
    WDGS_TRY(wdgs_alloc((void**)&op->counts, N * 4, true, op->dev->stream));
    WDGS_TRY(wdgs_alloc((void**)&op->offsets, N * 4, true, op->dev->stream));
    WDGS_TRY(scan_scratch_create(&op->scan, (u32)N));
    op->capacity = (u32)N;
    return WDGS_OK;
}

// computeMaxOutPoints (densify-prune.ts:390-410): 24 B Gaussian, 96 B SH per point; max_buffer_bytes = 0 lifts the cap.
static u32 compute_max_out(const wdgs_densify_config& c, u32 n) {
    uint64_t m = 2ull * std::max(n, 1u);  // every input emits at most 2
    if (c.max_buffer_bytes) m = std::min<uint64_t>(c.max_buffer_bytes / 24, c.max_buffer_bytes / 96);
    if (c.max_new_points_per_step > 0) m = std::min<uint64_t>(m, (uint64_t)std::max(n, 1u) + c.max_new_points_per_step);
    return (u32)std::min<uint64_t>(m, 0xFFFFFFFFull);
}

// The four stages of encodePrepare, individually recordable like the reference's public methods.
int wdgs_densify_prune_encode_decision(wdgs_densify_prune* op, uint32_t n, const void* gaussians, const void* metric_counts) {
    WDGS_REQUIRE(op && gaussians, WDGS_E_INVALID, "wdgs_densify_prune_encode_decision: null argument");
    WDGS_TRY(wdgs_densify_prune_ensure_size(op, n));
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(op->dev, "densify_decide", decide_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const u32*)gaussians, (const u32*)metric_counts,
                op->cfg.clone_threshold, op->cfg.prune_threshold, op->cfg.split_threshold, op->counts, op->actions);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int wdgs_densify_prune_encode_prefix_sum(wdgs_densify_prune* op, uint32_t n) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "null op");
    WDGS_REQUIRE(op->actions && n <= op->capacity, WDGS_E_STATE, "encodePrefixSum before encodeDecision/ensureSize for %u points", n);
    if (n == 0) return WDGS_OK;
    return scan_exclusive_u32(op->dev, &op->scan, op->counts, op->offsets, n, nullptr);
}

int wdgs_densify_prune_encode_cap_to_max(wdgs_densify_prune* op, uint32_t n, uint32_t max_out_points) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "null op");
    WDGS_REQUIRE(op->actions && n <= op->capacity, WDGS_E_STATE, "encodeCapToMax before encodeDecision/ensureSize for %u points", n);
    op->last_max_out = max_out_points;
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(op->dev, "densify_cap", cap_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, max_out_points, op->offsets, op->counts, op->actions);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int wdgs_densify_prune_encode_total_out(wdgs_densify_prune* op, uint32_t n) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "null op");
    WDGS_REQUIRE(n == 0 || (op->actions && n <= op->capacity), WDGS_E_STATE, "encodeTotalOut before encodeDecision/ensureSize for %u points", n);
    WDGS_LAUNCH(op->dev, "densify_total", total_kernel, dim3(1), dim3(64), 0, n, op->offsets, op->counts, op->total);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int wdgs_densify_prune_compute_max_out_points(wdgs_densify_prune* op, uint32_t n, uint32_t* out) {
    WDGS_REQUIRE(op && out, WDGS_E_INVALID, "null argument");
    *out = compute_max_out(op->cfg, n);
    return WDGS_OK;
}

int wdgs_densify_prune_get_buffers(wdgs_densify_prune* op, wdgs_densify_prepared* out) {
    WDGS_REQUIRE(op && out, WDGS_E_INVALID, "null argument");
    WDGS_REQUIRE(op->actions, WDGS_E_STATE, "DensifyPrunePass: buffers not created yet (call ensureSize or an encode first)");
    out->action_buffer = op->actions;
    out->out_count_buffer = op->counts;
    out->out_offset_buffer = op->offsets;
    out->out_total_buffer = op->total;
    out->max_out_points = op->last_max_out;
    return WDGS_OK;
}

int wdgs_densify_prune_encode_prepare(wdgs_densify_prune* op, uint32_t n, const void* gaussians, const void* metric_counts, wdgs_densify_prepared* out) {
    WDGS_REQUIRE(op && gaussians && out, WDGS_E_INVALID, "wdgs_densify_prune_encode_prepare: null argument");
    const u32 max_out = compute_max_out(op->cfg, n);
    WDGS_TRY(wdgs_densify_prune_encode_decision(op, n, gaussians, metric_counts));
    WDGS_TRY(wdgs_densify_prune_encode_prefix_sum(op, n));
    WDGS_TRY(wdgs_densify_prune_encode_cap_to_max(op, n, max_out));
    WDGS_TRY(wdgs_densify_prune_encode_prefix_sum(op, n));
    WDGS_TRY(wdgs_densify_prune_encode_total_out(op, n));
    out->action_buffer = op->actions;
    out->out_count_buffer = op->counts;
    out->out_offset_buffer = op->offsets;
    out->out_total_buffer = op->total;
    out->max_out_points = max_out;
    return WDGS_OK;
}

int wdgs_densify_prune_read_total(wdgs_densify_prune* op, uint32_t* total_out) {
    WDGS_REQUIRE(op && total_out, WDGS_E_INVALID, "null argument");
    return wdgs_copy_to_host(op->dev, total_out, op->total, 4);
}

int wdgs_densify_prune_encode_scatter(wdgs_densify_prune* op, uint32_t in_points, const void* in_gaussians, const void* in_sh,
                                      const wdgs_optimizer_state* in_state, uint32_t out_num_points, int reset_new, void* out_gaussians, void* out_sh,
                                      const wdgs_optimizer_state* out_state) {
    WDGS_REQUIRE(op && in_gaussians && in_sh && out_gaussians && out_sh, WDGS_E_INVALID, "wdgs_densify_prune_encode_scatter: null argument");
    WDGS_REQUIRE(op->actions && in_points <= op->capacity, WDGS_E_STATE, "encode_scatter before encode_prepare for %u points", in_points);
    WDGS_REQUIRE((in_state == nullptr) == (out_state == nullptr), WDGS_E_INVALID, "in_state and out_state must both be given or both be null");
    WDGS_REQUIRE(out_num_points > 0, WDGS_E_INVALID, "encodeScatter: outPointCloud.num_points must be > 0");
    if (in_points == 0) return WDGS_OK;
    ScatterArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in_points = in_points; a.out_points = out_num_points; a.reset_new_state = reset_new ? 1u : 0u;
    a.offsets = op->offsets; a.counts = op->counts; a.actions = op->actions;
    a.in_gaussians = (const u32*)in_gaussians; a.in_sh = (const u32*)in_sh;
    a.out_gaussians = (u32*)out_gaussians; a.out_sh = (u32*)out_sh;
    a.has_state = in_state != nullptr;
    if (in_state) {
        a.in_pos = (const float4*)in_state->opt_pos; a.in_rot = (const float4*)in_state->opt_rot; a.in_scale = (const float4*)in_state->opt_scale;
        a.in_opacity = (const float*)in_state->opt_opacity; a.in_param_sh = (const float4*)in_state->param_sh; a.in_state_sh = (const float4*)in_state->state_sh;
        a.out_pos = (float4*)out_state->opt_pos; a.out_rot = (float4*)out_state->opt_rot; a.out_scale = (float4*)out_state->opt_scale;
        a.out_opacity = (float*)out_state->opt_opacity; a.out_param_sh = (float4*)out_state->param_sh; a.out_state_sh = (float4*)out_state->state_sh;
    }
    WDGS_LAUNCH(op->dev, "densify_scatter", scatter_kernel, dim3(ceil_div(in_points, 256)), dim3(256), 0, a);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

}  // extern "C"
