// Adam without bias correction (K18), fp32 -> fp16 re-pack (K19), fp16 -> fp32 master unpack (K20), and the
// data-parallel helpers (fp32 gradient accumulation across views + Adam on the reduced buffer).
//
// Replaces src/shaders/adam.wgsl:53-175, src/shaders/update-gaussians.wgsl:35-77 and the inline unpack shader of
// src/renderers/optimizer.ts:166-223.  The reference runs Adam and the re-pack as two N-wide launches; here they are
// one kernel (the re-pack only reads what Adam just wrote).
//
// HBM layout note.  The reference keeps the SH-DC parameter in param_sh[idx*48 + c] (192-byte stride) and its moments in
// state_sh[idx*48 + c] (384-byte stride): 36 useful bytes cost three extra cache lines per Gaussian per step (measured:
// 1.06 GB moved for 0.49 GB algorithmic at N = 1 M, profiles/r01b_pmc.json), and the position / log-scale structs (OptVec4 x 3) carry a
// padding lane each.  The optimizer therefore trains a compact copy "cs" = float[N][28] (adam.h: position, log-scale and SH-DC {param, m, v},
// 112 bytes per Gaussian) and the reference-layout arrays are brought up to date by cs_flush at every hand-over point (get_state /
// release_state / destroy), loaded by cs_load when state is adopted, unpacked or rewritten from outside.
// Arithmetic and visible results are unchanged.  Per visible Gaussian: 32 B gradient + 4 B visibility + 2 x (3 x 48 + 12
// + 36) B state; per Gaussian 24 B + 8 B re-pack writes.
#include "common.h"
#include "wgslm.h"
#include "adam.h"

namespace {

// guard (nullable): a device word that, when non-zero at execution time, turns the whole step into a no-op -- the forward pass's
// overflow word (its tile-entry list was truncated: gradients are incomplete), so a step that is going to be reported as
// WDGS_E_CAPACITY does not first corrupt the optimizer state.
__global__ __launch_bounds__(256) void adam_repack_kernel(u32 n, wdgs_adam_hyperparameters h, const u32* __restrict__ tile_counts,
                                                           const u32* __restrict__ gradients, float4* opt_rot, float* opt_opacity, CsView cs, u32* gaussians,
                                                           u32* sh_buffer, const u32* __restrict__ guard, u32* dc_words) {
    WD_STREAM_PRIO();
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    if (guard && *guard != 0u) return;
    const bool update = tile_counts[idx] != 0u;
    Grad14 g = {};
    if (update) g = unpack_gradient(gradients, idx);
    adam_and_repack(idx, update, g, h, opt_rot, opt_opacity, cs, gaussians, sh_buffer, nullptr, dc_words);
}

// Gaussians [first, first + count): the slice a data-parallel rank owns (first = 0, count = n on a single GPU).
__global__ __launch_bounds__(256) void adam_repack_f32_kernel(u32 first, u32 count, wdgs_adam_hyperparameters h, const u32* __restrict__ visible,
                                                               const float* __restrict__ grad_f32, float4* opt_rot, float* opt_opacity, CsView cs,
                                                               u32* gaussians, u32* sh_buffer,
                                                               const u32* __restrict__ guard, u32* __restrict__ guard_seen_host, u32* __restrict__ rows_out,
                                                               u32* dc_words) {
    WD_STREAM_PRIO();
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
    adam_and_repack(idx, update, g, h, opt_rot, opt_opacity, cs, gaussians, sh_buffer, rows_out, dc_words);
}

// Rows published by the other ranks (wdgs_comm_allgather_rows) -> this replica's point cloud: every Gaussian outside
// [skip_first, skip_first + skip_count), which this rank re-packed itself.
__global__ __launch_bounds__(256) void apply_rows_kernel(u32 n, const u32* __restrict__ rows, u32 skip_first, u32 skip_count, const u32* __restrict__ guard, u32* __restrict__ guard_seen_host,
                                                          u32* __restrict__ gaussians, u32* __restrict__ sh_buffer, u32* __restrict__ dc_words) {
    WD_STREAM_PRIO();
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
    if (dc_words) {  // deferred SH writes (adam.h)
        *reinterpret_cast<uint2*>(dc_words + (size_t)idx * 2) = make_uint2(b.z, b.w & 0xFFFFu);
    } else {
        u32* shp = sh_buffer + (size_t)idx * 24;
        shp[0] = b.z;
        reinterpret_cast<unsigned short*>(shp)[2] = (unsigned short)(b.w & 0xFFFFu);
    }
}

// SH rows -> compact DC words (when deferred writes are switched on) and back (hand-over points: a host read of the cloud, an export, the
// viewer, a densify rebuild -- anything that reads the 96-byte rows instead of going through project_count's dc_words argument)
__global__ __launch_bounds__(256) void dc_words_load_kernel(u32 n, const u32* __restrict__ sh_buffer, u32* __restrict__ dc_words) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const uint2 w = *reinterpret_cast<const uint2*>(sh_buffer + (size_t)idx * 24);
    *reinterpret_cast<uint2*>(dc_words + (size_t)idx * 2) = make_uint2(w.x, w.y & 0xFFFFu);
}
__global__ __launch_bounds__(256) void dc_words_flush_kernel(u32 n, const u32* __restrict__ dc_words, u32* __restrict__ sh_buffer) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const uint2 w = *reinterpret_cast<const uint2*>(dc_words + (size_t)idx * 2);
    u32* shp = sh_buffer + (size_t)idx * 24;
    shp[0] = w.x;
    reinterpret_cast<unsigned short*>(shp)[2] = (unsigned short)(w.y & 0xFFFFu);
}

// flag = (overwrite ? 0 : flag) | (src != 0): folds per-view overflow words into the one guard word of a batched step
__global__ void guard_accumulate_kernel(u32* __restrict__ flag, const u32* __restrict__ src, u32 overwrite) {
    const u32 prev = overwrite ? 0u : *flag;
    *flag = prev | (*src != 0u ? 1u : 0u);
}

// reference layout -> compact training copy (adam.h: CsView)
__global__ __launch_bounds__(256) void cs_load_kernel(u32 n, const float4* __restrict__ opt_pos, const float4* __restrict__ opt_scale, const float* __restrict__ param_sh,
                                                       const float2* __restrict__ state_sh, CsView cs) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float4 pp = opt_pos[(size_t)idx * 3], pm = opt_pos[(size_t)idx * 3 + 1], pv = opt_pos[(size_t)idx * 3 + 2];
    const float4 sp = opt_scale[(size_t)idx * 3], sm = opt_scale[(size_t)idx * 3 + 1], sv = opt_scale[(size_t)idx * 3 + 2];
    float cp[3], cm[3], cv[3];
#pragma unroll
    for (u32 c = 0; c < 3u; c++) {
        const float2 mv = state_sh[(size_t)idx * 48 + c];
        cp[c] = param_sh[(size_t)idx * 48 + c];
        cm[c] = mv.x;
        cv[c] = mv.y;
    }
    cs.quad(0, idx) = make_float4(pp.x, pp.y, pp.z, pm.x);
    cs.quad(1, idx) = make_float4(pm.y, pm.z, pv.x, pv.y);
    cs.quad(2, idx) = make_float4(pv.z, sp.x, sp.y, sp.z);
    cs.quad(3, idx) = make_float4(sm.x, sm.y, sm.z, sv.x);
    cs.quad(4, idx) = make_float4(sv.y, sv.z, cp[0], cp[1]);
    cs.quad(5, idx) = make_float4(cp[2], cm[0], cm[1], cm[2]);
    cs.quad(6, idx) = make_float4(cv[0], cv[1], cv[2], 0.0f);
}

// compact training copy -> reference layout.  The fourth lanes get the constants the reference's Adam writes (adam.wgsl:108, 142: position
// w = 1, scale w = 0, their moments 0) -- which is also what unpack and the densify scatter leave in Gaussians that were never updated.
__global__ __launch_bounds__(256) void cs_flush_kernel(u32 n, const CsView cs, float4* __restrict__ opt_pos, float4* __restrict__ opt_scale,
                                                        float* __restrict__ param_sh, float2* __restrict__ state_sh) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float4 q0 = cs.quad(0, idx), q1 = cs.quad(1, idx), q2 = cs.quad(2, idx), q3 = cs.quad(3, idx), q4 = cs.quad(4, idx), q5 = cs.quad(5, idx), q6 = cs.quad(6, idx);
    opt_pos[(size_t)idx * 3] = make_float4(q0.x, q0.y, q0.z, 1.0f);
    opt_pos[(size_t)idx * 3 + 1] = make_float4(q0.w, q1.x, q1.y, 0.0f);
    opt_pos[(size_t)idx * 3 + 2] = make_float4(q1.z, q1.w, q2.x, 0.0f);
    opt_scale[(size_t)idx * 3] = make_float4(q2.y, q2.z, q2.w, 0.0f);
    opt_scale[(size_t)idx * 3 + 1] = make_float4(q3.x, q3.y, q3.z, 0.0f);
    opt_scale[(size_t)idx * 3 + 2] = make_float4(q3.w, q4.x, q4.y, 0.0f);
    const float cp[3] = {q4.z, q4.w, q5.x}, cm[3] = {q5.y, q5.z, q5.w}, cv[3] = {q6.x, q6.y, q6.z};
#pragma unroll
    for (u32 c = 0; c < 3u; c++) {
        param_sh[(size_t)idx * 48 + c] = cp[c];
        state_sh[(size_t)idx * 48 + c] = make_float2(cm[c], cv[c]);
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
                       const wdgs_optimizer_state& st, const CsView& cs, void* gaussians, void* sh, const void* guard, void* dc_words) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "adam_repack", adam_repack_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, h, (const u32*)tile_counts, (const u32*)gradients,
                (float4*)st.opt_rot, (float*)st.opt_opacity, cs, (u32*)gaussians, (u32*)sh, (const u32*)guard, (u32*)dc_words);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_adam_repack_f32(wdgs_device* dev, u32 first, u32 count, const wdgs_adam_hyperparameters& h, const void* visible, const void* grad_f32,
                           const wdgs_optimizer_state& st, const CsView& cs, void* gaussians, void* sh, const void* guard, void* guard_seen_host, void* rows_out,
                           void* dc_words) {
    if (count == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "adam_repack_f32", adam_repack_f32_kernel, dim3(ceil_div(count, 256)), dim3(256), 0, first, count, h, (const u32*)visible,
                (const float*)grad_f32, (float4*)st.opt_rot, (float*)st.opt_opacity, cs, (u32*)gaussians,
                (u32*)sh, (const u32*)guard, (u32*)guard_seen_host, (u32*)rows_out, (u32*)dc_words);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_apply_rows(wdgs_device* dev, u32 n, const void* rows, u32 skip_first, u32 skip_count, const void* guard, void* guard_seen_host, void* gaussians, void* sh,
                      void* dc_words) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "apply_repacked_rows", apply_rows_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const u32*)rows, skip_first, skip_count,
                (const u32*)guard, (u32*)guard_seen_host, (u32*)gaussians, (u32*)sh, (u32*)dc_words);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_dc_words_load(wdgs_device* dev, u32 n, const void* sh, void* dc_words) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "optimizer_dc_words_load", dc_words_load_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const u32*)sh, (u32*)dc_words);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_dc_words_flush(wdgs_device* dev, u32 n, const void* dc_words, void* sh) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "optimizer_dc_words_flush", dc_words_flush_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const u32*)dc_words, (u32*)sh);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_guard_accumulate(wdgs_device* dev, void* flag, const void* src, u32 overwrite) {
    WDGS_LAUNCH(dev, "guard_accumulate", guard_accumulate_kernel, dim3(1), dim3(1), 0, (u32*)flag, (const u32*)src, overwrite);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_cs_load(wdgs_device* dev, u32 n, const wdgs_optimizer_state& st, const CsView& cs) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "optimizer_cs_load", cs_load_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, (const float4*)st.opt_pos, (const float4*)st.opt_scale,
                (const float*)st.param_sh, (const float2*)st.state_sh, cs);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int launch_cs_flush(wdgs_device* dev, u32 n, const CsView& cs, const wdgs_optimizer_state& st) {
    if (n == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "optimizer_cs_flush", cs_flush_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, n, cs, (float4*)st.opt_pos, (float4*)st.opt_scale,
                (float*)st.param_sh, (float2*)st.state_sh);
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
