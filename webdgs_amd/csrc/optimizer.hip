// Adam without bias correction (K18), fp32 -> fp16 re-pack (K19), fp16 -> fp32 master unpack (K20), and the
// data-parallel helpers (fp32 gradient accumulation across views + Adam on the reduced buffer).
//
// Replaces src/shaders/adam.wgsl:53-175, src/shaders/update-gaussians.wgsl:35-77 and the inline unpack shader of
// src/renderers/optimizer.ts:166-223.  The reference runs Adam and the re-pack as two N-wide launches; here they are
// one kernel (the re-pack only reads what Adam just wrote).
//
// HBM layout note.  The reference keeps the SH-DC parameter in param_sh[idx*48 + c] (192-byte stride) and its moments in
// state_sh[idx*48 + c] (384-byte stride): 36 useful bytes cost three extra cache lines per Gaussian per step (measured:
// 1.06 GB moved for 0.49 GB algorithmic at N = 1 M, profiles/r01b_pmc.json).  The optimizer therefore trains a compact
// copy "dc" = float[N][9] {param rgb, m rgb, v rgb} and the reference-layout arrays are brought up to date by dc_flush at
// every hand-over point (get_state / release_state / destroy), loaded by dc_load when state is adopted or unpacked.
// Arithmetic and visible results are unchanged.  Per visible Gaussian: 32 B gradient + 4 B visibility + 2 x (3 x 48 + 12
// + 36) B state; per Gaussian 24 B + 8 B re-pack writes.
#include "common.h"
#include "wgslm.h"

namespace {

struct Adam3 { float p, m, v; };
WD_DEV Adam3 adam_step(const wdgs_adam_hyperparameters& h, float param, float grad, float m, float v, float lr) {
    const float m_new = h.beta1 * m + (1.0f - h.beta1) * grad;
    const float v_new = h.beta2 * v + (1.0f - h.beta2) * grad * grad;
    const float step = wd_div(-lr * m_new, wd_sqrt(v_new) + h.epsilon);
    return Adam3{param + step, m_new, v_new};
}

struct Grad14 { float pos[3], opac, rot[4], scale[3], color[3]; };

WD_DEV Grad14 unpack_gradient(const u32* __restrict__ gradients, u32 idx) {
    const uint4* gp = reinterpret_cast<const uint4*>(gradients + (size_t)idx * 8);
    const uint4 a = gp[0], b = gp[1];
    Grad14 g;
    g.pos[0] = wd_unpack_lo(a.x); g.pos[1] = wd_unpack_hi(a.x); g.pos[2] = wd_unpack_lo(a.y); g.opac = wd_unpack_hi(a.y);
    g.rot[0] = wd_unpack_lo(a.z); g.rot[1] = wd_unpack_hi(a.z); g.rot[2] = wd_unpack_lo(a.w); g.rot[3] = wd_unpack_hi(a.w);
    g.scale[0] = wd_unpack_lo(b.x); g.scale[1] = wd_unpack_hi(b.x); g.scale[2] = wd_unpack_lo(b.y);
    g.color[0] = wd_unpack_lo(b.z); g.color[1] = wd_unpack_hi(b.z); g.color[2] = wd_unpack_lo(b.w);
    return g;
}

// Adam on one Gaussian's 14 trained scalars (SH: DC only) followed by the fp16 re-pack of that Gaussian.
// rows_out (nullable): the re-packed row -- 6 Gaussian words, SH word 0, low half of SH word 1 -- also goes to rows_out[idx*8 ..],
// the 32-byte form in which a data-parallel rank publishes the Gaussians it owns (wdgs_comm_allgather_rows).
WD_DEV void adam_and_repack(u32 idx, bool update, const Grad14& g, const wdgs_adam_hyperparameters& h, float4* __restrict__ opt_pos,
                            float4* __restrict__ opt_rot, float4* __restrict__ opt_scale, float* __restrict__ opt_opacity,
                            float* __restrict__ dc, u32* __restrict__ gaussians, u32* __restrict__ sh_buffer, u32* __restrict__ rows_out = nullptr) {
    float4 P = opt_pos[(size_t)idx * 3];
    float4 R = opt_rot[(size_t)idx * 3];
    float4 S = opt_scale[(size_t)idx * 3];
    float op = opt_opacity[(size_t)idx * 3];
    float* d = dc + (size_t)idx * 9;
    float c0 = d[0], c1 = d[1], c2 = d[2];
    if (update) {
        {
            const float4 m = opt_pos[(size_t)idx * 3 + 1], v = opt_pos[(size_t)idx * 3 + 2];
            const Adam3 rx = adam_step(h, P.x, g.pos[0], m.x, v.x, h.lr_pos), ry = adam_step(h, P.y, g.pos[1], m.y, v.y, h.lr_pos),
                        rz = adam_step(h, P.z, g.pos[2], m.z, v.z, h.lr_pos);
            P = make_float4(rx.p, ry.p, rz.p, 1.0f);
            opt_pos[(size_t)idx * 3] = P;
            opt_pos[(size_t)idx * 3 + 1] = make_float4(rx.m, ry.m, rz.m, 0.0f);
            opt_pos[(size_t)idx * 3 + 2] = make_float4(rx.v, ry.v, rz.v, 0.0f);
        }
        {
            const float4 m = opt_rot[(size_t)idx * 3 + 1], v = opt_rot[(size_t)idx * 3 + 2];
            const Adam3 rx = adam_step(h, R.x, g.rot[0], m.x, v.x, h.lr_rot), ry = adam_step(h, R.y, g.rot[1], m.y, v.y, h.lr_rot),
                        rz = adam_step(h, R.z, g.rot[2], m.z, v.z, h.lr_rot), rw = adam_step(h, R.w, g.rot[3], m.w, v.w, h.lr_rot);
            const vec4 nr = normalize(V4(rx.p, ry.p, rz.p, rw.p));
            R = make_float4(nr.x, nr.y, nr.z, nr.w);
            opt_rot[(size_t)idx * 3] = R;
            opt_rot[(size_t)idx * 3 + 1] = make_float4(rx.m, ry.m, rz.m, rw.m);
            opt_rot[(size_t)idx * 3 + 2] = make_float4(rx.v, ry.v, rz.v, rw.v);
        }
        {
            const float4 m = opt_scale[(size_t)idx * 3 + 1], v = opt_scale[(size_t)idx * 3 + 2];
            const Adam3 rx = adam_step(h, S.x, g.scale[0], m.x, v.x, h.lr_scale), ry = adam_step(h, S.y, g.scale[1], m.y, v.y, h.lr_scale),
                        rz = adam_step(h, S.z, g.scale[2], m.z, v.z, h.lr_scale);
            S = make_float4(rx.p, ry.p, rz.p, 0.0f);
            opt_scale[(size_t)idx * 3] = S;
            opt_scale[(size_t)idx * 3 + 1] = make_float4(rx.m, ry.m, rz.m, 0.0f);
            opt_scale[(size_t)idx * 3 + 2] = make_float4(rx.v, ry.v, rz.v, 0.0f);
        }
        {
            const Adam3 r = adam_step(h, op, g.opac, opt_opacity[(size_t)idx * 3 + 1], opt_opacity[(size_t)idx * 3 + 2], h.lr_opacity);
            op = r.p;
            opt_opacity[(size_t)idx * 3] = r.p;
            opt_opacity[(size_t)idx * 3 + 1] = r.m;
            opt_opacity[(size_t)idx * 3 + 2] = r.v;
        }
        {
            const Adam3 r0 = adam_step(h, c0, g.color[0], d[3], d[6], h.lr_color), r1 = adam_step(h, c1, g.color[1], d[4], d[7], h.lr_color),
                        r2 = adam_step(h, c2, g.color[2], d[5], d[8], h.lr_color);
            c0 = r0.p; c1 = r1.p; c2 = r2.p;
            d[0] = r0.p; d[1] = r1.p; d[2] = r2.p;
            d[3] = r0.m; d[4] = r1.m; d[5] = r2.m;
            d[6] = r0.v; d[7] = r1.v; d[8] = r2.v;
        }
    }
    // re-pack (update-gaussians.wgsl:41-75): whole Gaussian, SH word 0, low half of SH word 1
    u32* gp = gaussians + (size_t)idx * 6;
    *reinterpret_cast<uint2*>(gp) = make_uint2(wd_pack2(P.x, P.y), wd_pack2(P.z, op));
    *reinterpret_cast<uint2*>(gp + 2) = make_uint2(wd_pack2(R.x, R.y), wd_pack2(R.z, R.w));
    *reinterpret_cast<uint2*>(gp + 4) = make_uint2(wd_pack2(S.x, S.y), wd_pack2(S.z, 0.0f));
    // SH word 0 and the LOW half of word 1 (the third DC coefficient) as a 4-byte and a 2-byte store.  The reference writes
    // pack2x16float(c2, unpack2x16float(old).y): the neighbouring coefficient survives that round trip bit for bit (only a NaN
    // payload could change, which WGSL leaves implementation-defined), so not touching it is the same result -- without fetching
    // the 96-byte row's cache line just to copy 2 bytes back (it was 19 % of this kernel's HBM traffic).
    u32* shp = sh_buffer + (size_t)idx * 24;
    const u32 sh0 = wd_pack2(c0, c1), sh1lo = wd_f16bits(c2) & 0xFFFFu;
    shp[0] = sh0;
    reinterpret_cast<unsigned short*>(shp)[2] = (unsigned short)sh1lo;
    if (rows_out) {
        uint4* ro = reinterpret_cast<uint4*>(rows_out + (size_t)idx * 8);
        ro[0] = make_uint4(wd_pack2(P.x, P.y), wd_pack2(P.z, op), wd_pack2(R.x, R.y), wd_pack2(R.z, R.w));
        ro[1] = make_uint4(wd_pack2(S.x, S.y), wd_pack2(S.z, 0.0f), sh0, sh1lo);
    }
}

// guard (nullable): a device word that, when non-zero at execution time, turns the whole step into a no-op -- the forward pass's
// overflow word (its tile-entry list was truncated: gradients are incomplete), so a step that is going to be reported as
// WDGS_E_CAPACITY does not first corrupt the optimizer state.
__global__ __launch_bounds__(256) void adam_repack_kernel(u32 n, wdgs_adam_hyperparameters h, const u32* __restrict__ tile_counts,
                                                           const u32* __restrict__ gradients, float4* opt_pos, float4* opt_rot, float4* opt_scale,
                                                           float* opt_opacity, float* dc, u32* gaussians, u32* sh_buffer, const u32* __restrict__ guard) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    if (guard && *guard != 0u) return;
    const bool update = tile_counts[idx] != 0u;
    Grad14 g = {};
    if (update) g = unpack_gradient(gradients, idx);
    adam_and_repack(idx, update, g, h, opt_pos, opt_rot, opt_scale, opt_opacity, dc, gaussians, sh_buffer);
}

// Gaussians [first, first + count): the slice a data-parallel rank owns (first = 0, count = n on a single GPU).
__global__ __launch_bounds__(256) void adam_repack_f32_kernel(u32 first, u32 count, wdgs_adam_hyperparameters h, const u32* __restrict__ visible,
                                                               const float* __restrict__ grad_f32, float4* opt_pos, float4* opt_rot,
                                                               float4* opt_scale, float* opt_opacity, float* dc, u32* gaussians, u32* sh_buffer,
                                                               const u32* __restrict__ guard, u32* __restrict__ guard_seen_host, u32* __restrict__ rows_out) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    if (guard && *guard != 0u) {
        if (t == 0u && guard_seen_host) *guard_seen_host = 1u;  // sticky host-visible note: this step skipped itself (deferred_checks, api.hip)
        return;
    }
    const u32 idx = first + t;
    const bool update = visible[idx] != 0u;
    Grad14 g = {};
    if (update) {
        const float* gp = grad_f32 + (size_t)idx * 14;
        g.pos[0] = gp[0]; g.pos[1] = gp[1]; g.pos[2] = gp[2]; g.opac = gp[3];
        g.rot[0] = gp[4]; g.rot[1] = gp[5]; g.rot[2] = gp[6]; g.rot[3] = gp[7];
        g.scale[0] = gp[8]; g.scale[1] = gp[9]; g.scale[2] = gp[10];
        g.color[0] = gp[11]; g.color[1] = gp[12]; g.color[2] = gp[13];
    }
    adam_and_repack(idx, update, g, h, opt_pos, opt_rot, opt_scale, opt_opacity, dc, gaussians, sh_buffer, rows_out);
}

// Rows published by the other ranks (wdgs_comm_allgather_rows) -> this replica's point cloud: every Gaussian outside
// [skip_first, skip_first + skip_count), which this rank re-packed itself.
__global__ __launch_bounds__(256) void apply_rows_kernel(u32 n, const u32* __restrict__ rows, u32 skip_first, u32 skip_count, const u32* __restrict__ guard, u32* __restrict__ guard_seen_host,
                                                          u32* __restrict__ gaussians, u32* __restrict__ sh_buffer) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0u && guard && guard_seen_host && *guard != 0u) *guard_seen_host = 1u;  // (a rank whose owned slice is empty runs no Adam kernel)
    if (idx >= n || (idx >= skip_first && idx - skip_first < skip_count)) return;
    if (guard && *guard != 0u) return;
    const uint4* ri = reinterpret_cast<const uint4*>(rows + (size_t)idx * 8);
    const uint4 a = ri[0], b = ri[1];
    u32* gp = gaussians + (size_t)idx * 6;
    *reinterpret_cast<uint2*>(gp) = make_uint2(a.x, a.y);
    *reinterpret_cast<uint2*>(gp + 2) = make_uint2(a.z, a.w);
    *reinterpret_cast<uint2*>(gp + 4) = make_uint2(b.x, b.y);
    u32* shp = sh_buffer + (size_t)idx * 24;
    shp[0] = b.z;
    reinterpret_cast<unsigned short*>(shp)[2] = (unsigned short)(b.w & 0xFFFFu);
}

// flag = (overwrite ? 0 : flag) | (src != 0): folds per-view overflow words into the one guard word of a batched step
__global__ void guard_accumulate_kernel(u32* __restrict__ flag, const u32* __restrict__ src, u32 overwrite) {
    const u32 prev = overwrite ? 0u : *flag;
    *flag = prev | (*src != 0u ? 1u : 0u);
}

// reference layout -> compact DC copy
__global__ __launch_bounds__(256) void dc_load_kernel(u32 n, const float* __restrict__ param_sh, const float2* __restrict__ state_sh, float* __restrict__ dc) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    float* d = dc + (size_t)idx * 9;
#pragma unroll
    for (u32 c = 0; c < 3u; c++) {
        const float2 mv = state_sh[(size_t)idx * 48 + c];
        d[c] = param_sh[(size_t)idx * 48 + c];
        d[3 + c] = mv.x;
        d[6 + c] = mv.y;
    }
}

// compact DC copy -> reference layout
__global__ __launch_bounds__(256) void dc_flush_kernel(u32 n, const float* __restrict__ dc, float* __restrict__ param_sh, float2* __restrict__ state_sh) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float* d = dc + (size_t)idx * 9;
#pragma unroll
    for (u32 c = 0; c < 3u; c++) {
        param_sh[(size_t)idx * 48 + c] = d[c];
        state_sh[(size_t)idx * 48 + c] = make_float2(d[3 + c], d[6 + c]);
    }
}

__global__ __launch_bounds__(256) void accumulate_gradients_kernel(u32 n, const u32* __restrict__ gradients, const u32* __restrict__ tile_counts,
                                                                    float* __restrict__ acc, u32* __restrict__ visible) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    if (tile_counts[idx] == 0u) return;
    const Grad14 g = unpack_gradient(gradients, idx);
    float* a = acc + (size_t)idx * 14;
    a[0] += g.pos[0]; a[1] += g.pos[1]; a[2] += g.pos[2]; a[3] += g.opac;
    a[4] += g.rot[0]; a[5] += g.rot[1]; a[6] += g.rot[2]; a[7] += g.rot[3];
    a[8] += g.scale[0]; a[9] += g.scale[1]; a[10] += g.scale[2];
    a[11] += g.color[0]; a[12] += g.color[1]; a[13] += g.color[2];
    visible[idx] += 1u;
}

// First view of a batch: acc = unpack(GaussianGradient) (zeros where the Gaussian touched no tile), visible = 0 or 1 -- the
// overwrite form of accumulate_gradients, so the 60 B/Gaussian block needs no clearing pass before it.
__global__ __launch_bounds__(256) void store_gradients_kernel(u32 n, const u32* __restrict__ gradients, const u32* __restrict__ tile_counts,
                                                               float* __restrict__ acc, u32* __restrict__ visible) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const bool vis = tile_counts[idx] != 0u;
    Grad14 g = {};
    if (vis) g = unpack_gradient(gradients, idx);
    float* a = acc + (size_t)idx * 14;  // 56-byte rows: 8-byte aligned
    reinterpret_cast<float2*>(a)[0] = make_float2(g.pos[0], g.pos[1]);
    reinterpret_cast<float2*>(a)[1] = make_float2(g.pos[2], g.opac);
    reinterpret_cast<float2*>(a)[2] = make_float2(g.rot[0], g.rot[1]);
    reinterpret_cast<float2*>(a)[3] = make_float2(g.rot[2], g.rot[3]);
    reinterpret_cast<float2*>(a)[4] = make_float2(g.scale[0], g.scale[1]);
    reinterpret_cast<float2*>(a)[5] = make_float2(g.scale[2], g.color[0]);
    reinterpret_cast<float2*>(a)[6] = make_float2(g.color[1], g.color[2]);
    visible[idx] = vis ? 1u : 0u;
}

__global__ __launch_bounds__(256) void unpack_kernel(u32 n, const u32* __restrict__ gaussians, const u32* __restrict__ sh_buffer, float4* opt_pos,
                                                      float4* opt_rot, float4* opt_scale, float* opt_opacity, float* param_sh) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const u32* gp = gaussians + (size_t)idx * 6;
    const uint2 w01 = *reinterpret_cast<const uint2*>(gp), w23 = *reinterpret_cast<const uint2*>(gp + 2), w45 = *reinterpret_cast<const uint2*>(gp + 4);
    opt_pos[(size_t)idx * 3] = make_float4(wd_unpack_lo(w01.x), wd_unpack_hi(w01.x), wd_unpack_lo(w01.y), 1.0f);
    opt_opacity[(size_t)idx * 3] = wd_unpack_hi(w01.y);
    opt_rot[(size_t)idx * 3] = make_float4(wd_unpack_lo(w23.x), wd_unpack_hi(w23.x), wd_unpack_lo(w23.y), wd_unpack_hi(w23.y));
    opt_scale[(size_t)idx * 3] = make_float4(wd_unpack_lo(w45.x), wd_unpack_hi(w45.x), wd_unpack_lo(w45.y), 0.0f);
    const uint4* shp = reinterpret_cast<const uint4*>(sh_buffer + (size_t)idx * 24);
    float4* out = reinterpret_cast<float4*>(param_sh + (size_t)idx * 48);
#pragma unroll
    for (u32 q = 0; q < 6u; q++) {
        const uint4 w = shp[q];
        out[q * 2] = make_float4(wd_unpack_lo(w.x), wd_unpack_hi(w.x), wd_unpack_lo(w.y), wd_unpack_hi(w.y));
        out[q * 2 + 1] = make_float4(wd_unpack_lo(w.z), wd_unpack_hi(w.z), wd_unpack_lo(w.w), wd_unpack_hi(w.w));
    }
}

}  // namespace

int launch_adam_repack(wdgs_device* dev, u32 n, const wdgs_adam_hyperparameters& h, const void* tile_counts, const void* gradients,
                       const wdgs_optimizer_state& st, void* dc, void* gaussians, void* sh, const void* guard) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "adam_repack", adam_repack_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, h, (const u32*)tile_counts, (const u32*)gradients,
                (float4*)st.opt_pos, (float4*)st.opt_rot, (float4*)st.opt_scale, (float*)st.opt_opacity, (float*)dc, (u32*)gaussians, (u32*)sh,
                (const u32*)guard);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_adam_repack_f32(wdgs_device* dev, u32 first, u32 count, const wdgs_adam_hyperparameters& h, const void* visible, const void* grad_f32,
                           const wdgs_optimizer_state& st, void* dc, void* gaussians, void* sh, const void* guard, void* guard_seen_host, void* rows_out) {
    if (count == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "adam_repack_f32", adam_repack_f32_kernel, dim3(ceil_div(count, 256)), dim3(256), 0, first, count, h, (const u32*)visible,
                (const float*)grad_f32, (float4*)st.opt_pos, (float4*)st.opt_rot, (float4*)st.opt_scale, (float*)st.opt_opacity, (float*)dc, (u32*)gaussians,
                (u32*)sh, (const u32*)guard, (u32*)guard_seen_host, (u32*)rows_out);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_apply_rows(wdgs_device* dev, u32 n, const void* rows, u32 skip_first, u32 skip_count, const void* guard, void* guard_seen_host, void* gaussians, void* sh) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "apply_repacked_rows", apply_rows_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const u32*)rows, skip_first, skip_count,
                (const u32*)guard, (u32*)guard_seen_host, (u32*)gaussians, (u32*)sh);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_guard_accumulate(wdgs_device* dev, void* flag, const void* src, u32 overwrite) {
    WDGS_LAUNCH(dev, "guard_accumulate", guard_accumulate_kernel, dim3(1), dim3(1), 0, (u32*)flag, (const u32*)src, overwrite);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_dc_load(wdgs_device* dev, u32 n, const wdgs_optimizer_state& st, void* dc) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "optimizer_dc_load", dc_load_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const float*)st.param_sh, (const float2*)st.state_sh, (float*)dc);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_dc_flush(wdgs_device* dev, u32 n, const void* dc, const wdgs_optimizer_state& st) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "optimizer_dc_flush", dc_flush_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const float*)dc, (float*)st.param_sh, (float2*)st.state_sh);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_accumulate_gradients(wdgs_device* dev, u32 n, const void* gradients, const void* tile_counts, void* acc, void* visible) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "accumulate_gradients", accumulate_gradients_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const u32*)gradients,
                (const u32*)tile_counts, (float*)acc, (u32*)visible);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_store_gradients(wdgs_device* dev, u32 n, const void* gradients, const void* tile_counts, void* acc, void* visible) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "store_gradients", store_gradients_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const u32*)gradients, (const u32*)tile_counts,
                (float*)acc, (u32*)visible);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_unpack(wdgs_device* dev, u32 n, const void* gaussians, const void* sh, const wdgs_optimizer_state& st) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "optimizer_unpack", unpack_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const u32*)gaussians, (const u32*)sh, (float4*)st.opt_pos,
                (float4*)st.opt_rot, (float4*)st.opt_scale, (float*)st.opt_opacity, (float*)st.param_sh);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}
