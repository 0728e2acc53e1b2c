// Per-Gaussian geometry backward (K17): chain rule from (mean2D, conic, opacity, colour) gradients to mean3D, log-scale,
// quaternion and raw opacity; colour passes through.
//
// Replaces main_geometry_backward (src/shaders/tiled-backward.wgsl:41-298): N-wide, HBM-bound (24 B Gaussian + 48 B
// accumulators in, 32 B packed fp16 gradient out), runs on culled Gaussians too.  The reference's quirks are
// kept: W = view3x3 here vs its transpose in the forward (Q10), +0.5*viewport for both NDC axes (Q11), fp16 output (Q13).
#include "common.h"
#include "wgslm.h"
#include "adam.h"

namespace {

constexpr u32 ACC_STRIDE = 12;  // i32 per Gaussian: mean.xy, conic.xyz, opacity, rgb, 3 pad (backward_raster.hip)

WD_DEV float from_fixed(int v) { return wd_div((float)v, 1000000.0f); }

// Where a batched step wants a view's gradient: the fp32 block the views of the step are summed in (parallel.py), its visibility counts
// and its guard word.  mode 1 stores (the first view of the step: no clearing pass), mode 2 adds.
struct ViewAccumulate {
    float* sums;              // f32[N][14]
    u32* visible;             // u32[N]
    const u32* tile_counts;   // the forward pass's tile count of each Gaussian: 0 = not in this view
    u32* guard;               // guard word of the step
    const u32* overflow;      // the forward pass's overflow word of this view
    u32 mode;
};

// The reference's single-view step runs Adam and the re-pack (K18 + K19) right behind K17 on the gradient K17 has just written; here the
// same thread goes on with its Gaussian: the fp16-rounded gradient is taken from the registers that were packed, the state of the
// Gaussian is read, updated and re-packed (adam.h) -- no second pass over N, no read-back of the gradient.
struct ViewAdam {
    wdgs_adam_hyperparameters h;
    const u32* tile_counts;
    float4* opt_rot;
    float* opt_opacity;
    CsView cs;       // compact training copy (adam.h)
    u32* gaussians;  // the same buffer K17 reads: this thread's own 24-byte row, read above, re-packed below
    u32* sh;
    const u32* guard;
    u32* dc_words;   // nullable: deferred SH writes (adam.h)
};

// MODE 1: the view's gradient also goes into the step's fp32 block -- the values that accumulate_gradients / store_gradients
// (optimizer.hip) would read back from the packed fp16 gradient, taken from the registers that were just packed.  MODE 2: Adam.
// K17 for one Gaussian under one camera: the three accumulator quads and the Gaussian's six words in, the packed GaussianGradient out
// (tiled-backward.wgsl:41-298).  Shared by the per-view kernel and the view-batched one, so both evaluate the same operations in the same order.
WD_DEV void geometry_chain(const int4 a0, const int4 a1, const int4 a2, const uint2 w01, const uint2 w23, const uint2 w45, const float* __restrict__ camera_f,
                           const RenderSettings& settings, uint4& o0, uint4& o1) {
    const vec2 dL_dmean2D_px = V2(from_fixed(a0.x), from_fixed(a0.y));
    const vec3 dL_dconic = V3(from_fixed(a0.z), from_fixed(a0.w), from_fixed(a1.x));
    const float dL_dopac = from_fixed(a1.y);

    const vec3 mean3D = V3(wd_unpack_lo(w01.x), wd_unpack_hi(w01.x), wd_unpack_lo(w01.y));
    const float opacity_raw = wd_unpack_hi(w01.y);
    const float opacity_sigmoid = wd_div(1.0f, 1.0f + wd_exp(-opacity_raw));
    const vec4 rot = V4(wd_unpack_lo(w23.x), wd_unpack_hi(w23.x), wd_unpack_lo(w23.y), wd_unpack_hi(w23.y));
    const vec3 log_scale = V3(wd_unpack_lo(w45.x), wd_unpack_hi(w45.x), wd_unpack_lo(w45.y));
    const vec3 scale = vexp(log_scale);

    const Cov3D c3 = covariance3D(rot, scale);
    const CameraUniforms& cam = *reinterpret_cast<const CameraUniforms*>(camera_f);
    const mat4 view = cam.view;
    const vec3 t = xyz(view * V4(mean3D, 1.0f));

    const vec2 viewport = V2(settings.viewport_x, settings.viewport_y);
    const vec2 dL_dmean2D_ndc = dL_dmean2D_px * 0.5f * viewport;
    const mat4 view_proj = cam.proj * cam.view;
    const vec4 p_hom = view_proj * V4(mean3D, 1.0f);
    const float rw = wd_div(1.0f, p_hom.w + 0.0000001f);
    const float rw2 = rw * rw;
    const vec4 dL_dphom = V4(dL_dmean2D_ndc.x * rw, dL_dmean2D_ndc.y * rw, 0.0f, -(dL_dmean2D_ndc.x * p_hom.x + dL_dmean2D_ndc.y * p_hom.y) * rw2);
    const vec3 dL_dmean3D_proj = xyz(transpose(view_proj) * dL_dphom);

    const float focal_x = cam.focal.x, focal_y = cam.focal.y;
    const float limx = wd_div(1.3f * viewport.x * 0.5f, focal_x);
    const float limy = wd_div(1.3f * viewport.y * 0.5f, focal_y);
    const float txtz = wd_div(t.x, t.z), tytz = wd_div(t.y, t.z);
    const float tcx = wd_min(limx, wd_max(-limx, txtz)) * t.z;
    const float tcy = wd_min(limy, wd_max(-limy, tytz)) * t.z;
    const float x_grad_mul = (txtz >= -limx && txtz <= limx) ? 1.0f : 0.0f;
    const float y_grad_mul = (tytz >= -limy && tytz <= limy) ? 1.0f : 0.0f;

    const mat3 J = M3(V3(wd_div(focal_x, t.z), 0.0f, wd_div(-(focal_x * tcx), t.z * t.z)),
                      V3(0.0f, wd_div(focal_y, t.z), wd_div(-(focal_y * tcy), t.z * t.z)), V3(0.0f, 0.0f, 0.0f));
    const mat3 Wm = M3(xyz(view.c[0]), xyz(view.c[1]), xyz(view.c[2]));
    const mat3 Tm = Wm * J;
    const mat3 Vrk = M3(V3(c3.v[0], c3.v[1], c3.v[2]), V3(c3.v[1], c3.v[3], c3.v[4]), V3(c3.v[2], c3.v[4], c3.v[5]));
    const mat3 cov2D = transpose(Tm) * Vrk * Tm;
    const float a = cov2D.c[0].x + 0.3f, b = cov2D.c[0].y, c = cov2D.c[1].y + 0.3f;

    const float denom = a * c - b * b;
    const float denom2inv = wd_div(1.0f, (denom * denom) + 0.0000001f);
    float dL_da = 0.0f, dL_db = 0.0f, dL_dc = 0.0f;
    if (denom2inv != 0.0f) {
        dL_da = denom2inv * (-c * c * dL_dconic.x + 2.0f * b * c * dL_dconic.y + (denom - a * c) * dL_dconic.z);
        dL_dc = denom2inv * (-a * a * dL_dconic.z + 2.0f * a * b * dL_dconic.y + (denom - a * c) * dL_dconic.x);
        dL_db = denom2inv * 2.0f * (b * c * dL_dconic.x - (denom + 2.0f * b * b) * dL_dconic.y + a * b * dL_dconic.z);
    }
#define TM(c_, r_) el(Tm, c_, r_)
#define VR(c_, r_) el(Vrk, c_, r_)
#define WW(c_, r_) el(Wm, c_, r_)
    float d3[6];
    d3[0] = (TM(0, 0) * TM(0, 0) * dL_da + TM(0, 0) * TM(1, 0) * dL_db + TM(1, 0) * TM(1, 0) * dL_dc);
    d3[3] = (TM(0, 1) * TM(0, 1) * dL_da + TM(0, 1) * TM(1, 1) * dL_db + TM(1, 1) * TM(1, 1) * dL_dc);
    d3[5] = (TM(0, 2) * TM(0, 2) * dL_da + TM(0, 2) * TM(1, 2) * dL_db + TM(1, 2) * TM(1, 2) * dL_dc);
    d3[1] = 2.0f * TM(0, 0) * TM(0, 1) * dL_da + (TM(0, 0) * TM(1, 1) + TM(0, 1) * TM(1, 0)) * dL_db + 2.0f * TM(1, 0) * TM(1, 1) * dL_dc;
    d3[2] = 2.0f * TM(0, 0) * TM(0, 2) * dL_da + (TM(0, 0) * TM(1, 2) + TM(0, 2) * TM(1, 0)) * dL_db + 2.0f * TM(1, 0) * TM(1, 2) * dL_dc;
    d3[4] = 2.0f * TM(0, 2) * TM(0, 1) * dL_da + (TM(0, 1) * TM(1, 2) + TM(0, 2) * TM(1, 1)) * dL_db + 2.0f * TM(1, 1) * TM(1, 2) * dL_dc;

    const float dL_dT00 = 2.0f * (TM(0, 0) * VR(0, 0) + TM(0, 1) * VR(0, 1) + TM(0, 2) * VR(0, 2)) * dL_da + (TM(1, 0) * VR(0, 0) + TM(1, 1) * VR(0, 1) + TM(1, 2) * VR(0, 2)) * dL_db;
    const float dL_dT01 = 2.0f * (TM(0, 0) * VR(1, 0) + TM(0, 1) * VR(1, 1) + TM(0, 2) * VR(1, 2)) * dL_da + (TM(1, 0) * VR(1, 0) + TM(1, 1) * VR(1, 1) + TM(1, 2) * VR(1, 2)) * dL_db;
    const float dL_dT02 = 2.0f * (TM(0, 0) * VR(2, 0) + TM(0, 1) * VR(2, 1) + TM(0, 2) * VR(2, 2)) * dL_da + (TM(1, 0) * VR(2, 0) + TM(1, 1) * VR(2, 1) + TM(1, 2) * VR(2, 2)) * dL_db;
    const float dL_dT10 = 2.0f * (TM(1, 0) * VR(0, 0) + TM(1, 1) * VR(0, 1) + TM(1, 2) * VR(0, 2)) * dL_dc + (TM(0, 0) * VR(0, 0) + TM(0, 1) * VR(0, 1) + TM(0, 2) * VR(0, 2)) * dL_db;
    const float dL_dT11 = 2.0f * (TM(1, 0) * VR(1, 0) + TM(1, 1) * VR(1, 1) + TM(1, 2) * VR(1, 2)) * dL_dc + (TM(0, 0) * VR(1, 0) + TM(0, 1) * VR(1, 1) + TM(0, 2) * VR(1, 2)) * dL_db;
    const float dL_dT12 = 2.0f * (TM(1, 0) * VR(2, 0) + TM(1, 1) * VR(2, 1) + TM(1, 2) * VR(2, 2)) * dL_dc + (TM(0, 0) * VR(2, 0) + TM(0, 1) * VR(2, 1) + TM(0, 2) * VR(2, 2)) * dL_db;

    const float dL_dJ00 = WW(0, 0) * dL_dT00 + WW(0, 1) * dL_dT01 + WW(0, 2) * dL_dT02;
    const float dL_dJ02 = WW(2, 0) * dL_dT00 + WW(2, 1) * dL_dT01 + WW(2, 2) * dL_dT02;
    const float dL_dJ11 = WW(1, 0) * dL_dT10 + WW(1, 1) * dL_dT11 + WW(1, 2) * dL_dT12;
    const float dL_dJ12 = WW(2, 0) * dL_dT10 + WW(2, 1) * dL_dT11 + WW(2, 2) * dL_dT12;
#undef TM
#undef VR
#undef WW
    const float tz = wd_div(1.0f, t.z);
    const float tz2 = tz * tz, tz3 = tz2 * tz;
    const float dL_dtx = x_grad_mul * -focal_x * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -focal_y * tz2 * dL_dJ12;
    const float dL_dtz = -focal_x * tz2 * dL_dJ00 - focal_y * tz2 * dL_dJ11 + (2.0f * focal_x * tcx) * tz3 * dL_dJ02 + (2.0f * focal_y * tcy) * tz3 * dL_dJ12;
    const vec3 dL_dmean3D_cov = xyz(transpose(view) * V4(dL_dtx, dL_dty, dL_dtz, 0.0f));

    const float x = rot.y, y = rot.z, z = rot.w, r = rot.x;
    const mat3 R = quat_to_R(rot);
    const mat3 M = diag3(scale) * R;
    const mat3 dL_dSigma = M3(V3(d3[0], 0.5f * d3[1], 0.5f * d3[2]), V3(0.5f * d3[1], d3[3], 0.5f * d3[4]), V3(0.5f * d3[2], 0.5f * d3[4], d3[5]));
    const mat3 dL_dM = (2.0f * M) * dL_dSigma;
    const mat3 dL_dMt = transpose(dL_dM);
    const mat3 Rt = transpose(R);
    const vec3 dL_dscale = V3(dot(Rt.c[0], dL_dMt.c[0]), dot(Rt.c[1], dL_dMt.c[1]), dot(Rt.c[2], dL_dMt.c[2]));
    mat3 D = dL_dMt;
    D.c[0] = dL_dMt.c[0] * scale.x;
    D.c[1] = dL_dMt.c[1] * scale.y;
    D.c[2] = dL_dMt.c[2] * scale.z;
#define DD(c_, r_) el(D, c_, r_)
    const float dL_drot_x = 2.0f * z * (DD(0, 1) - DD(1, 0)) + 2.0f * y * (DD(2, 0) - DD(0, 2)) + 2.0f * x * (DD(1, 2) - DD(2, 1));
    const float dL_drot_y = 2.0f * y * (DD(1, 0) + DD(0, 1)) + 2.0f * z * (DD(2, 0) + DD(0, 2)) + 2.0f * r * (DD(1, 2) - DD(2, 1)) - 4.0f * x * (DD(2, 2) + DD(1, 1));
    const float dL_drot_z = 2.0f * x * (DD(1, 0) + DD(0, 1)) + 2.0f * r * (DD(2, 0) - DD(0, 2)) + 2.0f * z * (DD(1, 2) + DD(2, 1)) - 4.0f * y * (DD(2, 2) + DD(0, 0));
    const float dL_drot_w = 2.0f * r * (DD(0, 1) - DD(1, 0)) + 2.0f * x * (DD(2, 0) + DD(0, 2)) + 2.0f * y * (DD(1, 2) + DD(2, 1)) - 4.0f * z * (DD(1, 1) + DD(0, 0));
#undef DD
    const vec3 final_dL_dmean3D = dL_dmean3D_proj + dL_dmean3D_cov;
    const float dL_dopacity_raw = dL_dopac * opacity_sigmoid * (1.0f - opacity_sigmoid);
    vec3 dL_dlog_scale = dL_dscale * scale;
    {
        const float cap_px = settings.max_splat_radius_px;
        if (cap_px > 0.0f) {
            const float denom_cap = a * c - b * b;
            if (denom_cap > 0.0f) {
                const float conic_x = wd_div(c, denom_cap), conic_y = wd_div(-b, denom_cap), conic_z = wd_div(a, denom_cap);
                const float disc = conic_y * conic_y - conic_x * conic_z;
                const float t_cap = 2.0f * wd_log(opacity_sigmoid * 128.0f);
                if (t_cap > 0.0f && disc < 0.0f) {
                    const float x_extent = wd_sqrt(wd_div(t_cap * conic_z, -disc));
                    const float y_extent = wd_sqrt(wd_div(t_cap * conic_x, -disc));
                    if (wd_max(x_extent, y_extent) >= cap_px) dL_dlog_scale = vmax(dL_dlog_scale, V3(0.0f));
                }
            }
        }
    }
    o0.x = wd_pack2(final_dL_dmean3D.x, final_dL_dmean3D.y);
    o0.y = wd_pack2(final_dL_dmean3D.z, dL_dopacity_raw);
    o0.z = wd_pack2(dL_drot_x, dL_drot_y);
    o0.w = wd_pack2(dL_drot_z, dL_drot_w);
    o1.x = wd_pack2(dL_dlog_scale.x, dL_dlog_scale.y);
    o1.y = wd_pack2(dL_dlog_scale.z, 0.0f);
    o1.z = wd_pack2(from_fixed(a1.z), from_fixed(a1.w));
    // blue arrives as four partial sums, one per 16-lane row of the waves that produced it (backward_raster.hip); i32 sums wrap, as atomicAdd does
    o1.w = wd_pack2(from_fixed((int)((unsigned)a2.x + (unsigned)a2.y + (unsigned)a2.z + (unsigned)a2.w)), 0.0f);
}

template <int MODE>
__global__ __launch_bounds__(256, 6) void geometry_backward_kernel(u32 n, const float* __restrict__ camera_f, RenderSettings settings,
                                                                 const u32* gaussians, int* __restrict__ acc, u32* __restrict__ acc_dirty,
                                                                 u32* __restrict__ gradients, ViewAccumulate va, ViewAdam ad) {
    WD_STREAM_PRIO();
    constexpr bool ACC = MODE == 1;
    // The Trainer's forms (MODE 1, 2) CONSUME the accumulators: a row that held sums is put back to zero by the thread that read it and
    // the state word says "clean", so the next view's clear has nothing to do (backward_raster.hip: acc_clear_if_dirty).  The plain
    // form (MODE 0: TiledBackwardPass.encode) leaves the sums in place for readers.
    constexpr bool CONSUME = MODE != 0;
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (ACC && idx == 0u) *va.guard = (va.mode == 1u ? 0u : *va.guard) | (*va.overflow != 0u ? 1u : 0u);  // guard_accumulate (optimizer.hip)
    if (CONSUME && idx == 0u) *acc_dirty = 0u;
    if (idx >= n) return;
    int4* ap = reinterpret_cast<int4*>(acc + (size_t)idx * ACC_STRIDE);
    const int4 a0 = ap[0], a1 = ap[1], a2 = ap[2];
    if (CONSUME && ((a0.x | a0.y | a0.z | a0.w | a1.x | a1.y | a1.z | a1.w | a2.x | a2.y | a2.z | a2.w) != 0)) {
        const int4 z = make_int4(0, 0, 0, 0);
        ap[0] = z; ap[1] = z; ap[2] = z;
    }
    const u32* gp = gaussians + (size_t)idx * 6;
    const uint2 w01 = *reinterpret_cast<const uint2*>(gp), w23 = *reinterpret_cast<const uint2*>(gp + 2), w45 = *reinterpret_cast<const uint2*>(gp + 4);
    uint4 o0, o1;
    geometry_chain(a0, a1, a2, w01, w23, w45, camera_f, settings, o0, o1);
    if (MODE != 2 || gradients) {   // (MODE 2, the fused step: only for a host that reads the packed gradient -- wdgs_tiled_backward_set_gradient_output)
        uint4* op = reinterpret_cast<uint4*>(gradients + (size_t)idx * 8);
        op[0] = o0;
        op[1] = o1;
    }
    if (ACC) {
        const bool vis = va.tile_counts[idx] != 0u;
        float* a = va.sums + (size_t)idx * 14;  // 56-byte rows: 8-byte aligned
        float2* a2p = reinterpret_cast<float2*>(a);
        // the fp16-rounded values, in the block's order: pos.xyz, opacity, rot.wxyz, scale.xyz, colour.rgb
        const float g[14] = {wd_unpack_lo(o0.x), wd_unpack_hi(o0.x), wd_unpack_lo(o0.y), wd_unpack_hi(o0.y), wd_unpack_lo(o0.z), wd_unpack_hi(o0.z), wd_unpack_lo(o0.w),
                             wd_unpack_hi(o0.w), wd_unpack_lo(o1.x), wd_unpack_hi(o1.x), wd_unpack_lo(o1.y), wd_unpack_lo(o1.z), wd_unpack_hi(o1.z), wd_unpack_lo(o1.w)};
        if (va.mode == 1u) {
#pragma unroll
            for (u32 k = 0; k < 7u; k++) a2p[k] = vis ? make_float2(g[2 * k], g[2 * k + 1]) : make_float2(0.f, 0.f);
            va.visible[idx] = vis ? 1u : 0u;
        } else if (vis) {
#pragma unroll
            for (u32 k = 0; k < 7u; k++) {
                const float2 s = a2p[k];
                a2p[k] = make_float2(s.x + g[2 * k], s.y + g[2 * k + 1]);
            }
            va.visible[idx] += 1u;
        }
    }
    if (MODE == 2) {
        // guard: a step whose tile-entry list overflowed is skipped (optimizer.hip: adam_repack_kernel)
        if (ad.guard && *ad.guard != 0u) return;
        const bool update = ad.tile_counts[idx] != 0u;
        Grad14 g = {};
        if (update) {
            g.pos[0] = wd_unpack_lo(o0.x); g.pos[1] = wd_unpack_hi(o0.x); g.pos[2] = wd_unpack_lo(o0.y); g.opac = wd_unpack_hi(o0.y);
            g.rot[0] = wd_unpack_lo(o0.z); g.rot[1] = wd_unpack_hi(o0.z); g.rot[2] = wd_unpack_lo(o0.w); g.rot[3] = wd_unpack_hi(o0.w);
            g.scale[0] = wd_unpack_lo(o1.x); g.scale[1] = wd_unpack_hi(o1.x); g.scale[2] = wd_unpack_lo(o1.y);
            g.color[0] = wd_unpack_lo(o1.z); g.color[1] = wd_unpack_hi(o1.z); g.color[2] = wd_unpack_lo(o1.w);
        }
        adam_and_repack(idx, update, g, ad.h, ad.opt_rot, ad.opt_opacity, ad.cs, ad.gaussians, ad.sh, nullptr, ad.dc_words);
    }
}


// K17 for ALL the views of a batched step in one pass over the Gaussians (the Trainer's view-batched step; no reference counterpart: the
// reference is batch-1).  Per view the thread reads that view's accumulator row (and puts it back to zero), evaluates the same chain rule
// under that view's camera, rounds the gradient to fp16 as K17 does, and adds it -- in VIEW ORDER, fp32, the first visible view's value
// stored, later ones added: exactly the sequence of the per-view accumulate kernels -- into a register copy of the step's fp32 block,
// which is written once.  The Gaussian is read once instead of once per view, the 60-byte fp32 row is written once instead of being
// read-modify-written per view, and the per-view kernels' cross-lane ordering (view k's sums behind view k-1's) disappears.
struct GeometryViews {
    u32 count;
    const float* camera[WDGS_MAX_BATCH_VIEWS];
    int* acc[WDGS_MAX_BATCH_VIEWS];
    u32* acc_dirty[WDGS_MAX_BATCH_VIEWS];
    const u32* tile_counts[WDGS_MAX_BATCH_VIEWS];
    const u32* overflow[WDGS_MAX_BATCH_VIEWS];
    u32* gradients[WDGS_MAX_BATCH_VIEWS];   // nullable per view: the packed per-view GaussianGradient, for readers of getGradientsBuffer()
};
__global__ __launch_bounds__(256, 3) void geometry_backward_views_kernel(u32 n, RenderSettings settings, const u32* __restrict__ gaussians, GeometryViews gv,
                                                                        float* __restrict__ sums, u32* __restrict__ visible, u32* __restrict__ guard,
                                                                        u32 continues /*0: these are the step's first views; 1: the block already holds earlier views*/) {
    WD_STREAM_PRIO();
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0u) {
        u32 g = continues ? *guard : 0u;
        for (u32 v = 0; v < gv.count; v++) { g |= (*gv.overflow[v] != 0u) ? 1u : 0u; *gv.acc_dirty[v] = 0u; }
        *guard = g;
    }
    if (idx >= n) return;
    const u32* gp = gaussians + (size_t)idx * 6;
    const uint2 w01 = *reinterpret_cast<const uint2*>(gp), w23 = *reinterpret_cast<const uint2*>(gp + 2), w45 = *reinterpret_cast<const uint2*>(gp + 4);
    float s[14];
    u32 nvis = 0u;
    float2* a2p = reinterpret_cast<float2*>(sums + (size_t)idx * 14);  // 56-byte rows: 8-byte aligned
    if (continues) {   // a later group of the step's views: go on from what the earlier groups left
#pragma unroll
        for (u32 k = 0; k < 7u; k++) { const float2 t = a2p[k]; s[2 * k] = t.x; s[2 * k + 1] = t.y; }
        nvis = visible[idx];
    } else {
#pragma unroll
        for (u32 k = 0; k < 14u; k++) s[k] = 0.0f;
    }
    for (u32 v = 0; v < gv.count; v++) {
        int4* ap = reinterpret_cast<int4*>(gv.acc[v] + (size_t)idx * ACC_STRIDE);
        const int4 a0 = ap[0], a1 = ap[1], a2 = ap[2];
        const bool vis = gv.tile_counts[v][idx] != 0u;
        if ((a0.x | a0.y | a0.z | a0.w | a1.x | a1.y | a1.z | a1.w | a2.x | a2.y | a2.z | a2.w) != 0) {
            const int4 z = make_int4(0, 0, 0, 0);
            ap[0] = z; ap[1] = z; ap[2] = z;
        }
        u32* gout = gv.gradients[v];
        if (!vis && !gout) continue;   // (a Gaussian outside the view contributes nothing; its packed gradient is only of interest to a reader)
        uint4 o0, o1;
        geometry_chain(a0, a1, a2, w01, w23, w45, gv.camera[v], settings, o0, o1);
        if (gout) {
            uint4* op = reinterpret_cast<uint4*>(gout + (size_t)idx * 8);
            op[0] = o0;
            op[1] = o1;
        }
        if (vis) {
            // the fp16-rounded values, in the block's order: pos.xyz, opacity, rot.wxyz, scale.xyz, colour.rgb
            const float g[14] = {wd_unpack_lo(o0.x), wd_unpack_hi(o0.x), wd_unpack_lo(o0.y), wd_unpack_hi(o0.y), wd_unpack_lo(o0.z), wd_unpack_hi(o0.z), wd_unpack_lo(o0.w),
                                 wd_unpack_hi(o0.w), wd_unpack_lo(o1.x), wd_unpack_hi(o1.x), wd_unpack_lo(o1.y), wd_unpack_lo(o1.z), wd_unpack_hi(o1.z), wd_unpack_lo(o1.w)};
            if (v == 0u && !continues) {   // the per-view path STORES the step's first view's value (a -0.0 stays -0.0) and adds the later ones
#pragma unroll
                for (u32 k = 0; k < 14u; k++) s[k] = g[k];
            } else {
#pragma unroll
                for (u32 k = 0; k < 14u; k++) s[k] = s[k] + g[k];
            }
            nvis++;
        }
    }
#pragma unroll
    for (u32 k = 0; k < 7u; k++) a2p[k] = make_float2(s[2 * k], s[2 * k + 1]);
    visible[idx] = nvis;
}

}  // namespace

int launch_geometry_backward_adam(wdgs_device* dev, u32 n, const void* camera, const RenderSettings& st, void* gaussians, void* acc, void* acc_dirty, void* gradients,
                                  const wdgs_adam_hyperparameters& h, const void* tile_counts, const wdgs_optimizer_state& state, const CsView& cs, void* sh,
                                  const void* guard, void* dc_words) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "geometry_backward_adam", geometry_backward_kernel<2>, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const float*)camera, st,
                (const u32*)gaussians, (int*)acc, (u32*)acc_dirty, (u32*)gradients, ViewAccumulate{},
                (ViewAdam{h, (const u32*)tile_counts, (float4*)state.opt_rot, (float*)state.opt_opacity, cs, (u32*)gaussians, (u32*)sh, (const u32*)guard,
                          (u32*)dc_words}));
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_geometry_backward(wdgs_device* dev, u32 n, const void* camera, const RenderSettings& st, const void* gaussians, void* acc, void* gradients) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "geometry_backward", geometry_backward_kernel<0>, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const float*)camera, st, (const u32*)gaussians,
                (int*)acc, (u32*)nullptr, (u32*)gradients, ViewAccumulate{}, ViewAdam{});
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_geometry_backward_accumulate(wdgs_device* dev, u32 n, const void* camera, const RenderSettings& st, const void* gaussians, void* acc, void* acc_dirty,
                                        void* gradients, void* sums, void* visible, const void* tile_counts, void* guard, const void* overflow, u32 mode) {
    // (n == 0 still runs one workgroup: the guard word must be written)
    WDGS_LAUNCH(dev, "geometry_backward", geometry_backward_kernel<1>, dim3(std::max(ceil_div(n, 256), 1u)), dim3(256), 0, n, (const float*)camera, st,
                (const u32*)gaussians, (int*)acc, (u32*)acc_dirty, (u32*)gradients,
                (ViewAccumulate{(float*)sums, (u32*)visible, (const u32*)tile_counts, (u32*)guard, (const u32*)overflow, mode}), ViewAdam{});
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_geometry_backward_views(wdgs_device* dev, u32 n, u32 count, const void* const* cameras, const RenderSettings& st, const void* gaussians, void* const* accs,
                                   void* const* acc_dirtys, const void* const* tile_counts, const void* const* overflows, void* const* gradients, void* sums, void* visible,
                                   void* guard, u32 continues) {
    GeometryViews gv{};
    gv.count = count;
    for (u32 v = 0; v < count; v++) {
        gv.camera[v] = (const float*)cameras[v]; gv.acc[v] = (int*)accs[v]; gv.acc_dirty[v] = (u32*)acc_dirtys[v]; gv.tile_counts[v] = (const u32*)tile_counts[v];
        gv.overflow[v] = (const u32*)overflows[v]; gv.gradients[v] = gradients ? (u32*)gradients[v] : nullptr;
    }
    // (n == 0 still runs one workgroup: the guard word must be written)
    WDGS_LAUNCH(dev, "geometry_backward_views", geometry_backward_views_kernel, dim3(std::max(ceil_div(n, 256), 1u)), dim3(256), 0, n, st, (const u32*)gaussians, gv,
                (float*)sums, (u32*)visible, (u32*)guard, continues);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}
