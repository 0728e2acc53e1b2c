// Scalar image metrics.  The reference never reduces its loss to a scalar (it only shows a per-pixel
// gradient image, src/trainer.ts:695-768); BASELINE's north star asks for PSNR against the reference, so the library
// provides the exact integer sum of squared rgb8 differences -- order-free, hence bit-reproducible -- and the host turns
// it into PSNR = 10 log10(255^2 * 3P / SSE).  One streaming pass, 8 B read per pixel.
#include "common.h"
#include "dmath.h"

namespace {

__global__ __launch_bounds__(256) void image_sse_kernel(const u32* __restrict__ a, const u32* __restrict__ b, u32 npix, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long s_w[4];
    unsigned long long acc = 0ull;
    for (u32 p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const u32 x = a[p], y = b[p];
#pragma unroll
        for (u32 c = 0; c < 3u; c++) {
            const int d = (int)((x >> (8u * c)) & 0xFFu) - (int)((y >> (8u * c)) & 0xFFu);
            acc += (unsigned long long)(d * d);
        }
    }
#pragma unroll
    for (u32 d = 32; d >= 1; d >>= 1) acc += (unsigned long long)__shfl_xor((long long)acc, (int)d, 64);
    if ((threadIdx.x & 63u) == 0u) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, s_w[0] + s_w[1] + s_w[2] + s_w[3]);
}

// dmath self-test: evaluates one pinned primitive elementwise so the parity tests can compare the device's exp/log/fp16
// conversions/saturating casts with the oracle's over arbitrary bit patterns (DESIGN.md "bit-exact by construction").
__global__ void dmath_eval_kernel(u32 which, u32 n, const u32* __restrict__ in, u32* __restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 u = in[i];
    const float x = __uint_as_float(u);
    u32 r = 0u;
    switch (which) {
        case 0: r = __float_as_uint(wd_exp(x)); break;
        case 1: r = __float_as_uint(wd_log(x)); break;
        case 2: r = wd_f16bits(x); break;                                  // f32 -> f16 bits (RNE)
        case 3: r = __float_as_uint(wd_unpack_lo(u)); break;               // f16 bits (low half) -> f32
        case 4: r = (u32)wd_to_i32(x); break;                              // saturating f32 -> i32
        case 5: r = wd_to_u32(x); break;                                   // saturating f32 -> u32
        case 6: r = __float_as_uint(wd_sqrt(x)); break;
        case 7: r = __float_as_uint(wd_div(1.0f, x)); break;
        // the in-range forms used by the rasterization kernels, and their full forms on the same operands (second operand: the neighbour)
        case 8: r = __float_as_uint(wd_exp_inrange(x)); break;
        case 9: r = __float_as_uint(wd_div_inrange(x, __uint_as_float(in[i ^ 1u]))); break;
        case 10: r = __float_as_uint(wd_div(x, __uint_as_float(in[i ^ 1u]))); break;
        default: break;
    }
    out[i] = r;
}

}  // namespace

extern "C" int wdgs_debug_eval_math(wdgs_device* dev, uint32_t which, uint32_t count, const void* in_u32_dev, void* out_u32_dev) {
    WDGS_REQUIRE(dev && (count == 0 || (in_u32_dev && out_u32_dev)), WDGS_E_INVALID, "wdgs_debug_eval_math: null argument");
    WDGS_REQUIRE(which <= 10u, WDGS_E_INVALID, "wdgs_debug_eval_math: unknown primitive %u", which);
    if (count == 0) return WDGS_OK;
    WDGS_REQUIRE(which < 9u || (count & 1u) == 0u, WDGS_E_INVALID, "wdgs_debug_eval_math: the two-operand primitives take an even count");
    WDGS_LAUNCH(dev, "dmath_eval", dmath_eval_kernel, dim3(ceil_div(count, 256)), dim3(256), 0, which, count, (const u32*)in_u32_dev, (u32*)out_u32_dev);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

extern "C" int wdgs_image_sse_rgb8(wdgs_device* dev, const void* a_dev, const void* b_dev, uint32_t num_pixels, void* out_u64_dev) {
    WDGS_REQUIRE(dev && a_dev && b_dev && out_u64_dev, WDGS_E_INVALID, "wdgs_image_sse_rgb8: null argument");
    WDGS_CHECK_HIP(hipMemsetAsync(out_u64_dev, 0, 8, dev->stream));
    if (num_pixels == 0) return WDGS_OK;
    const u32 grid = std::min<u32>(ceil_div(num_pixels, 256), (u32)dev->num_cus * 4u);
    WDGS_LAUNCH(dev, "image_sse", image_sse_kernel, dim3(grid), dim3(256), 0, (const u32*)a_dev, (const u32*)b_dev, num_pixels, (unsigned long long*)out_u64_dev);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}
