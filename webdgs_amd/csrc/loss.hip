// Per-pixel loss gradient (K15): lambda_l1*sign(d) + lambda_l2*d + lambda_dssim * 0.5*(1-SSIM_5x5)*d.
//
// Replaces compute_loss_grad (src/shaders/loss.wgsl:85-115, computeSSIMGrad 30-82), which issues 2 x 25 x 2
// uncached texture loads per pixel.  Here a workgroup stages the 36x36 halo of a 32x32 tile of both images in LDS as float4
// texels (clamp-to-edge), converting rgba8unorm -> f32 ONCE per texel through a 256-entry table of i/255 (each entry one
// correctly rounded division, so values equal the per-tap f32(u8)/255 of the restatement).  HBM traffic is the
// compulsory 8 B read + 16 B write per pixel; window taps are a conflict-free ds_read_b128 + ds_read_b64, shared by four pixels per thread.
// The window sums keep the reference's order (dy outer, dx inner); the second-moment accumulations are FMAs, the contraction the
// parity oracle pins (WGSL leaves it open).
#include <cstdlib>
#include "common.h"
#include "dmath.h"

namespace {

typedef wd_pair f2;  // component-wise scalar arithmetic (dmath.h)

WD_DEV float sgn(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

constexpr u32 LT = 32;       // tile width in pixels
constexpr u32 LH = LT + 4;   // halo width in texels

// PPT = vertically adjacent pixels per thread: the workgroup's tile is 32 x 8 PPT pixels.  More pixels per thread share more window texels
// (fewer LDS reads per pixel) but make fewer and longer waves.  Measured (profiles/r06j_loss_grad_pixels_per_thread_ab.txt): at c3 40.6 us
// with 2, 43.2 with 4 (rounds 2-3), 42.4 with 1; at c2, where 32 x 32 tiles leave one or two waves per SIMD and the kernel lasts as long as one
// wave's 3 000 instructions, 11.7 us with 1, 12.4 with 2, 14.8 with 4.  The launcher picks 1 up to 640 x 480 and 2 above; the arithmetic per
// pixel is the same in every form.
template <u32 PPT>
__global__ __launch_bounds__(256) void loss_grad_kernel(u32 W, u32 H, const u32* __restrict__ pred, const u32* __restrict__ targ,
                                                         wdgs_training_config cfg, float4* __restrict__ out, int4* __restrict__ acc, u32 acc_quads,
                                                         const u32* __restrict__ acc_dirty) {
    WD_STREAM_PRIO();
    // clearBuffer x4 of the gradient accumulators (tiled-backward-pass.ts:624-627) rides on this kernel, which precedes the backward
    // rasterization anyway: the accumulators' state word (backward_raster.hip) says whether anything has to be cleared at all -- after a
    // consuming K17 nothing has -- so the clear is one scalar load here instead of a launch of its own.
    if (acc && *acc_dirty != 0u) {
        const int4 z = make_int4(0, 0, 0, 0);
        const u32 nblk = gridDim.x * gridDim.y, blk = blockIdx.y * gridDim.x + blockIdx.x;
        for (u32 i = blk * 256u + threadIdx.x; i < acc_quads; i += nblk * 256u) acc[i] = z;
    }
    __shared__ float s_lut[256];
    // A thread row is 32 lanes on 32 consecutive texels, so each 16-lane group of a ds_read_b128 covers 256 contiguous
    // bytes = every bank once, whatever the row stride.
    // Texel pair = 24 bytes: {pred.r, pred.g, pred.b, targ.r} + {targ.g, targ.b} (two float4 would carry 8 bytes of padding per texel:
    // 41.5 KB per workgroup = 3 per CU; at 31 KB five fit, the halo loads of one overlap the window sums of the others and the tail of
    // the 2040-workgroup grid is shorter).
    constexpr u32 TH = 8u * PPT, HH = TH + 4u;   // tile and halo height
    __shared__ float4 sp[HH][LH];
    __shared__ float2 st[HH][LH];
    s_lut[threadIdx.x] = wd_div((float)threadIdx.x, 255.0f);
    __syncthreads();
    const int bx = blockIdx.x * LT, by = blockIdx.y * TH;
    for (u32 t = threadIdx.x; t < HH * LH; t += 256u) {
        const int hy = (int)(t / LH), hx = (int)(t % LH);
        int gx = bx + hx - 2, gy = by + hy - 2;
        gx = gx < 0 ? 0 : (gx > (int)W - 1 ? (int)W - 1 : gx);
        gy = gy < 0 ? 0 : (gy > (int)H - 1 ? (int)H - 1 : gy);
        const u32 a = pred[(size_t)gy * W + gx], b = targ[(size_t)gy * W + gx];
        sp[hy][hx] = make_float4(s_lut[a & 0xFFu], s_lut[(a >> 8) & 0xFFu], s_lut[(a >> 16) & 0xFFu], s_lut[b & 0xFFu]);
        st[hy][hx] = make_float2(s_lut[(b >> 8) & 0xFFu], s_lut[(b >> 16) & 0xFFu]);
    }
    __syncthreads();
    // Each thread owns PPT vertically adjacent pixels: their 5x5 windows share 8 x 5 texels, so a texel is read from LDS once
    // per pass instead of up to four times (the kernel is bound by LDS read bandwidth).  Walking the shared texels row-major
    // feeds every pixel its own window in the reference's order (dy outer, dx inner), one rounding per add.
    const u32 lx = threadIdx.x & (LT - 1u), ly0 = (threadIdx.x / LT) * PPT;
    const u32 x = bx + lx, y0 = by + ly0;
    if (x >= W || y0 >= H) return;

    const bool dssim = cfg.lambda_dssim > 0.0f;
    const float n = 25.0f;
    float mx[PPT][3], my[PPT][3], sx2[PPT][3], sy2[PPT][3], sxy[PPT][3];
    if (dssim) {
        // channel pairs (r,g) and (b,-) as f2: component-wise scalar arithmetic, each the reference's own operation
        f2 mxa[PPT], mxb[PPT], mya[PPT], myb[PPT];
#pragma unroll
        for (u32 k = 0; k < PPT; k++) mxa[k] = mxb[k] = mya[k] = myb[k] = f2{0.f, 0.f};
#pragma unroll
        for (u32 r = 0; r < PPT + 4u; r++) {
#pragma unroll
            for (u32 c = 0; c < 5u; c++) {
                const float4 a = sp[ly0 + r][lx + c];
                const float2 b = st[ly0 + r][lx + c];
#pragma unroll
                for (u32 k = 0; k < PPT; k++)
                    if (r >= k && r <= k + 4u) {
                        mxa[k] += f2{a.x, a.y}; mxb[k].x += a.z;
                        mya[k] += f2{a.w, b.x}; myb[k].x += b.y;
                    }
            }
            // pin the running sums here: otherwise the adds are sunk to their use and every loaded texel stays live (500 VGPRs)
#pragma unroll
            for (u32 k = 0; k < PPT; k++)
                asm volatile("" : "+v"(mxa[k].x), "+v"(mxa[k].y), "+v"(mxb[k].x), "+v"(mya[k].x), "+v"(mya[k].y), "+v"(myb[k].x));
        }
        f2 sx2a[PPT], sx2b[PPT], sy2a[PPT], sy2b[PPT], sxya[PPT], sxyb[PPT], ma[PPT], mb[PPT], na[PPT], nb[PPT];
#pragma unroll
        for (u32 k = 0; k < PPT; k++) {
            // (window sums of values in [0, 1] over 25; below, sums of squares over 25 and a quotient whose denominator is at least
            // c1 * c2 > 0: ordinary operands, so the division without operand scaling and special-case fix-up -- dmath.h)
            mx[k][0] = wd_div_inrange(mxa[k].x, n); mx[k][1] = wd_div_inrange(mxa[k].y, n); mx[k][2] = wd_div_inrange(mxb[k].x, n);
            my[k][0] = wd_div_inrange(mya[k].x, n); my[k][1] = wd_div_inrange(mya[k].y, n); my[k][2] = wd_div_inrange(myb[k].x, n);
            ma[k] = f2{mx[k][0], mx[k][1]}; mb[k] = f2{mx[k][2], 0.f};
            na[k] = f2{my[k][0], my[k][1]}; nb[k] = f2{my[k][2], 0.f};
            sx2a[k] = sx2b[k] = sy2a[k] = sy2b[k] = sxya[k] = sxyb[k] = f2{0.f, 0.f};
        }
#pragma unroll
        for (u32 r = 0; r < PPT + 4u; r++) {
#pragma unroll
            for (u32 c = 0; c < 5u; c++) {
                const float4 a = sp[ly0 + r][lx + c];
                const float2 b = st[ly0 + r][lx + c];
#pragma unroll
                for (u32 k = 0; k < PPT; k++)
                    if (r >= k && r <= k + 4u) {
                        const f2 daa = f2{a.x, a.y} - ma[k];
                        const float dab = a.z - mb[k].x;
                        const f2 dba = f2{a.w, b.x} - na[k];
                        const float dbb = b.y - nb[k].x;
                        // the window accumulations are FMAs (the pinned contraction of the parity oracle: WGSL may fuse the multiply)
                        sx2a[k].x = __builtin_fmaf(daa.x, daa.x, sx2a[k].x); sx2a[k].y = __builtin_fmaf(daa.y, daa.y, sx2a[k].y);
                        sx2b[k].x = __builtin_fmaf(dab, dab, sx2b[k].x);
                        sy2a[k].x = __builtin_fmaf(dba.x, dba.x, sy2a[k].x); sy2a[k].y = __builtin_fmaf(dba.y, dba.y, sy2a[k].y);
                        sy2b[k].x = __builtin_fmaf(dbb, dbb, sy2b[k].x);
                        sxya[k].x = __builtin_fmaf(daa.x, dba.x, sxya[k].x); sxya[k].y = __builtin_fmaf(daa.y, dba.y, sxya[k].y);
                        sxyb[k].x = __builtin_fmaf(dab, dbb, sxyb[k].x);
                    }
            }
#pragma unroll
            for (u32 k = 0; k < PPT; k++)
                asm volatile("" : "+v"(sx2a[k].x), "+v"(sx2a[k].y), "+v"(sx2b[k].x), "+v"(sy2a[k].x), "+v"(sy2a[k].y), "+v"(sy2b[k].x), "+v"(sxya[k].x),
                             "+v"(sxya[k].y), "+v"(sxyb[k].x));
        }
#pragma unroll
        for (u32 k = 0; k < PPT; k++) {
            sx2[k][0] = sx2a[k].x; sx2[k][1] = sx2a[k].y; sx2[k][2] = sx2b[k].x;
            sy2[k][0] = sy2a[k].x; sy2[k][1] = sy2a[k].y; sy2[k][2] = sy2b[k].x;
            sxy[k][0] = sxya[k].x; sxy[k][1] = sxya[k].y; sxy[k][2] = sxyb[k].x;
        }
    }
#pragma unroll
    for (u32 k = 0; k < PPT; k++) {
        const u32 y = y0 + k;
        if (y >= H) break;
        const float4 p = sp[ly0 + k + 2u][lx + 2u];
        const float2 t = st[ly0 + k + 2u][lx + 2u];
        const float d[3] = {p.x - p.w, p.y - t.x, p.z - t.y};
        float g[3] = {0.0f, 0.0f, 0.0f};
        if (dssim) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float vx = wd_div_inrange(sx2[k][c], n), vy = wd_div_inrange(sy2[k][c], n), vxy = wd_div_inrange(sxy[k][c], n);
                const float num1 = 2.0f * mx[k][c] * my[k][c] + cfg.c1;
                const float num2 = 2.0f * vxy + cfg.c2;
                const float den1 = mx[k][c] * mx[k][c] + my[k][c] * my[k][c] + cfg.c1;
                const float den2 = vx + vy + cfg.c2;
                // (a caller may set c1 = c2 = 0: then a flat window makes the denominator 0 and the full form's special cases are needed)
                const float den = den1 * den2;
                const float ssim = (cfg.c1 > 0.0f && cfg.c2 > 0.0f) ? wd_div_inrange(num1 * num2, den) : wd_div(num1 * num2, den);
                g[c] = ((1.0f - ssim) * 0.5f) * d[c];
            }
        }
        float4 o;
        o.x = cfg.lambda_l1 * sgn(d[0]) + cfg.lambda_l2 * d[0] + cfg.lambda_dssim * g[0];
        o.y = cfg.lambda_l1 * sgn(d[1]) + cfg.lambda_l2 * d[1] + cfg.lambda_dssim * g[1];
        o.z = cfg.lambda_l1 * sgn(d[2]) + cfg.lambda_l2 * d[2] + cfg.lambda_dssim * g[2];
        o.w = 1.0f;
        out[(size_t)y * W + x] = o;
    }
}

}  // namespace

int launch_loss_grad(wdgs_device* dev, u32 W, u32 H, const void* pred, const void* targ, const wdgs_training_config& cfg, void* out, void* acc, u32 acc_rows,
                     const void* acc_dirty) {
    if (W == 0 || H == 0) return WDGS_OK;
    // WDGS_LOSS_PPT=1|2|4 forces the pixels per thread (same-box A/B)
    static const int ppt_env = std::getenv("WDGS_LOSS_PPT") ? std::atoi(std::getenv("WDGS_LOSS_PPT")) : 0;
    const u32 ppt = ppt_env == 1 || ppt_env == 2 || ppt_env == 4 ? (u32)ppt_env : ((size_t)W * H <= 640u * 480u ? 1u : 2u);
#define WDGS_LOSS_LAUNCH(PPT_)                                                                                                                                       \
    WDGS_LAUNCH(dev, "loss_grad", loss_grad_kernel<PPT_>, dim3(ceil_div(W, LT), ceil_div(H, 8u * PPT_)), dim3(256), 0, W, H, (const u32*)pred, (const u32*)targ, cfg, \
                (float4*)out, (int4*)acc, acc_rows * 3u /*12 i32 per row*/, (const u32*)acc_dirty)
    if (ppt == 1u) { WDGS_LOSS_LAUNCH(1u); } else if (ppt == 2u) { WDGS_LOSS_LAUNCH(2u); } else { WDGS_LOSS_LAUNCH(4u); }
#undef WDGS_LOSS_LAUNCH
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}
