// Per-pixel loss gradient (K15): lambda_l1*sign(d) + lambda_l2*d + lambda_dssim * 0.5*(1-SSIM_5x5)*d.
//
// Replaces compute_loss_grad (src/shaders/loss.wgsl:85-115, computeSSIMGrad 30-82), which issues 2 x 25 x 2
// uncached texture loads per pixel.  Here a 16x16 workgroup stages the 20x20 halo of both images in LDS as float4
// texels (clamp-to-edge), converting rgba8unorm -> f32 ONCE per texel through a 256-entry table of i/255 (each entry one
// correctly rounded division, so values equal the per-tap f32(u8)/255 of the restatement).  HBM traffic is the
// compulsory 8 B read + 16 B write per pixel; the 2 x 50 window taps are broadcast-free ds_read_b128.
// The window sums keep the reference's order (dy outer, dx inner, one rounding per add).
#include "common.h"
#include "dmath.h"

namespace {

WD_DEV float sgn(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

__global__ __launch_bounds__(256) void loss_grad_kernel(u32 W, u32 H, const u32* __restrict__ pred, const u32* __restrict__ targ,
                                                         wdgs_training_config cfg, float4* __restrict__ out) {
    __shared__ float s_lut[256];
    // row stride 32 float4 = 512 B: rows land on the same banks, so the four 16-lane groups of a ds_read_b128 (which mix two
    // tile rows) stay conflict-free; a 20-wide row (320 B) made 59% of the LDS cycles bank conflicts.
    __shared__ float4 sp[20][32];
    __shared__ float4 st[20][32];
    s_lut[threadIdx.x] = wd_div((float)threadIdx.x, 255.0f);
    __syncthreads();
    const int bx = blockIdx.x * 16, by = blockIdx.y * 16;
    for (u32 t = threadIdx.x; t < 400u; t += 256u) {
        const int hy = (int)(t / 20u), hx = (int)(t % 20u);
        int gx = bx + hx - 2, gy = by + hy - 2;
        gx = gx < 0 ? 0 : (gx > (int)W - 1 ? (int)W - 1 : gx);
        gy = gy < 0 ? 0 : (gy > (int)H - 1 ? (int)H - 1 : gy);
        const u32 a = pred[(size_t)gy * W + gx], b = targ[(size_t)gy * W + gx];
        sp[hy][hx] = make_float4(s_lut[a & 0xFFu], s_lut[(a >> 8) & 0xFFu], s_lut[(a >> 16) & 0xFFu], 0.0f);
        st[hy][hx] = make_float4(s_lut[b & 0xFFu], s_lut[(b >> 8) & 0xFFu], s_lut[(b >> 16) & 0xFFu], 0.0f);
    }
    __syncthreads();
    const u32 lx = threadIdx.x & 15u, ly = threadIdx.x >> 4;
    const u32 x = bx + lx, y = by + ly;
    if (x >= W || y >= H) return;

    const float4 p = sp[ly + 2][lx + 2], t = st[ly + 2][lx + 2];
    const float d[3] = {p.x - t.x, p.y - t.y, p.z - t.z};
    float g[3] = {0.0f, 0.0f, 0.0f};
    if (cfg.lambda_dssim > 0.0f) {
        float mx[3] = {0, 0, 0}, my[3] = {0, 0, 0};
#pragma unroll
        for (u32 dy = 0; dy < 5u; dy++)
#pragma unroll
            for (u32 dx = 0; dx < 5u; dx++) {
                const float4 a = sp[ly + dy][lx + dx], b = st[ly + dy][lx + dx];
                mx[0] += a.x; mx[1] += a.y; mx[2] += a.z;
                my[0] += b.x; my[1] += b.y; my[2] += b.z;
            }
        const float n = 25.0f;
#pragma unroll
        for (int c = 0; c < 3; c++) { mx[c] = wd_div(mx[c], n); my[c] = wd_div(my[c], n); }
        float sx2[3] = {0, 0, 0}, sy2[3] = {0, 0, 0}, sxy[3] = {0, 0, 0};
#pragma unroll
        for (u32 dy = 0; dy < 5u; dy++)
#pragma unroll
            for (u32 dx = 0; dx < 5u; dx++) {
                const float4 a = sp[ly + dy][lx + dx], b = st[ly + dy][lx + dx];
                const float da[3] = {a.x - mx[0], a.y - mx[1], a.z - mx[2]}, db[3] = {b.x - my[0], b.y - my[1], b.z - my[2]};
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    sx2[c] += da[c] * da[c];
                    sy2[c] += db[c] * db[c];
                    sxy[c] += da[c] * db[c];
                }
            }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float vx = wd_div(sx2[c], n), vy = wd_div(sy2[c], n), vxy = wd_div(sxy[c], n);
            const float num1 = 2.0f * mx[c] * my[c] + cfg.c1;
            const float num2 = 2.0f * vxy + cfg.c2;
            const float den1 = mx[c] * mx[c] + my[c] * my[c] + cfg.c1;
            const float den2 = vx + vy + cfg.c2;
            const float ssim = wd_div(num1 * num2, den1 * den2);
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

}  // namespace

int launch_loss_grad(wdgs_device* dev, u32 W, u32 H, const void* pred, const void* targ, const wdgs_training_config& cfg, void* out) {
    if (W == 0 || H == 0) return WDGS_OK;
    WDGS_LAUNCH(dev, "loss_grad", loss_grad_kernel, dim3(ceil_div(W, 16), ceil_div(H, 16)), dim3(256), 0, W, H, (const u32*)pred, (const u32*)targ, cfg,
                (float4*)out);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}
