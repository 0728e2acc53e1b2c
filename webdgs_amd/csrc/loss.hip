// Per-pixel loss gradient (K15): lambda_l1*sign(d) + lambda_l2*d + lambda_dssim * 0.5*(1-SSIM_5x5)*d.
//
// Replaces compute_loss_grad (src/shaders/loss.wgsl:85-115, computeSSIMGrad 30-82), which issues 2 x 25 x 2
// uncached texture loads per pixel.  Here a 16x16 workgroup stages the 20x20 rgba8 halo of both images in LDS
// (clamp-to-edge), so HBM traffic is the compulsory 8 B read + 16 B write per pixel.
#include "common.h"
#include "dmath.h"

namespace {

struct f3 { float x, y, z; };
WD_DEV f3 unorm3(u32 t) { return f3{wd_div((float)(t & 0xFFu), 255.0f), wd_div((float)((t >> 8) & 0xFFu), 255.0f), wd_div((float)((t >> 16) & 0xFFu), 255.0f)}; }
WD_DEV float sgn(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

__global__ __launch_bounds__(256) void loss_grad_kernel(u32 W, u32 H, const u32* __restrict__ pred, const u32* __restrict__ targ,
                                                         wdgs_training_config cfg, float4* __restrict__ out) {
    __shared__ u32 sp[20][20];
    __shared__ u32 st[20][20];
    const int bx = blockIdx.x * 16, by = blockIdx.y * 16;
    for (u32 t = threadIdx.x; t < 400u; t += 256u) {
        const int hy = (int)(t / 20u), hx = (int)(t % 20u);
        int gx = bx + hx - 2, gy = by + hy - 2;
        gx = gx < 0 ? 0 : (gx > (int)W - 1 ? (int)W - 1 : gx);
        gy = gy < 0 ? 0 : (gy > (int)H - 1 ? (int)H - 1 : gy);
        sp[hy][hx] = pred[(size_t)gy * W + gx];
        st[hy][hx] = targ[(size_t)gy * W + gx];
    }
    __syncthreads();
    const u32 lx = threadIdx.x & 15u, ly = threadIdx.x >> 4;
    const u32 x = bx + lx, y = by + ly;
    if (x >= W || y >= H) return;

    const f3 p = unorm3(sp[ly + 2][lx + 2]), t = unorm3(st[ly + 2][lx + 2]);
    const f3 diff = f3{p.x - t.x, p.y - t.y, p.z - t.z};
    f3 gd = f3{0.0f, 0.0f, 0.0f};
    if (cfg.lambda_dssim > 0.0f) {
        f3 mu_x = f3{0, 0, 0}, mu_y = f3{0, 0, 0};
        for (u32 dy = 0; dy < 5u; dy++)
            for (u32 dx = 0; dx < 5u; dx++) {
                const f3 a = unorm3(sp[ly + dy][lx + dx]), b = unorm3(st[ly + dy][lx + dx]);
                mu_x = f3{mu_x.x + a.x, mu_x.y + a.y, mu_x.z + a.z};
                mu_y = f3{mu_y.x + b.x, mu_y.y + b.y, mu_y.z + b.z};
            }
        const float n = 25.0f;
        mu_x = f3{wd_div(mu_x.x, n), wd_div(mu_x.y, n), wd_div(mu_x.z, n)};
        mu_y = f3{wd_div(mu_y.x, n), wd_div(mu_y.y, n), wd_div(mu_y.z, n)};
        f3 sx2 = f3{0, 0, 0}, sy2 = f3{0, 0, 0}, sxy = f3{0, 0, 0};
        for (u32 dy = 0; dy < 5u; dy++)
            for (u32 dx = 0; dx < 5u; dx++) {
                const f3 a = unorm3(sp[ly + dy][lx + dx]), b = unorm3(st[ly + dy][lx + dx]);
                const f3 da = f3{a.x - mu_x.x, a.y - mu_x.y, a.z - mu_x.z}, db = f3{b.x - mu_y.x, b.y - mu_y.y, b.z - mu_y.z};
                sx2 = f3{sx2.x + da.x * da.x, sx2.y + da.y * da.y, sx2.z + da.z * da.z};
                sy2 = f3{sy2.x + db.x * db.x, sy2.y + db.y * db.y, sy2.z + db.z * db.z};
                sxy = f3{sxy.x + da.x * db.x, sxy.y + da.y * db.y, sxy.z + da.z * db.z};
            }
        const float mx[3] = {mu_x.x, mu_x.y, mu_x.z}, my[3] = {mu_y.x, mu_y.y, mu_y.z};
        const float vx[3] = {wd_div(sx2.x, n), wd_div(sx2.y, n), wd_div(sx2.z, n)};
        const float vy[3] = {wd_div(sy2.x, n), wd_div(sy2.y, n), wd_div(sy2.z, n)};
        const float vxy[3] = {wd_div(sxy.x, n), wd_div(sxy.y, n), wd_div(sxy.z, n)};
        const float d[3] = {diff.x, diff.y, diff.z};
        float g[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float num1 = 2.0f * mx[c] * my[c] + cfg.c1;
            const float num2 = 2.0f * vxy[c] + cfg.c2;
            const float den1 = mx[c] * mx[c] + my[c] * my[c] + cfg.c1;
            const float den2 = vx[c] + vy[c] + cfg.c2;
            const float ssim = wd_div(num1 * num2, den1 * den2);
            const float dssim = (1.0f - ssim) * 0.5f;
            g[c] = dssim * d[c];
        }
        gd = f3{g[0], g[1], g[2]};
    }
    float4 o;
    o.x = cfg.lambda_l1 * sgn(diff.x) + cfg.lambda_l2 * diff.x + cfg.lambda_dssim * gd.x;
    o.y = cfg.lambda_l1 * sgn(diff.y) + cfg.lambda_l2 * diff.y + cfg.lambda_dssim * gd.y;
    o.z = cfg.lambda_l1 * sgn(diff.z) + cfg.lambda_l2 * diff.z + cfg.lambda_dssim * gd.z;
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
