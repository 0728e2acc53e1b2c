// Scalar image metrics.  The reference never reduces its loss to a scalar (it only shows a per-pixel
// gradient image, src/trainer.ts:695-768); BASELINE's north star asks for PSNR against the reference, so the library
// provides the exact integer sum of squared rgb8 differences -- order-free, hence bit-reproducible -- and the host turns
// it into PSNR = 10 log10(255^2 * 3P / SSE).  One streaming pass, 8 B read per pixel.
#include "common.h"

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

}  // namespace

extern "C" int wdgs_image_sse_rgb8(wdgs_device* dev, const void* a_dev, const void* b_dev, uint32_t num_pixels, void* out_u64_dev) {
    WDGS_REQUIRE(dev && a_dev && b_dev && out_u64_dev, WDGS_E_INVALID, "wdgs_image_sse_rgb8: null argument");
    WDGS_CHECK_HIP(hipMemsetAsync(out_u64_dev, 0, 8, dev->stream));
    if (num_pixels == 0) return WDGS_OK;
    const u32 grid = std::min<u32>(ceil_div(num_pixels, 256), (u32)dev->num_cus * 4u);
    WDGS_LAUNCH(dev, "image_sse", image_sse_kernel, dim3(grid), dim3(256), 0, (const u32*)a_dev, (const u32*)b_dev, num_pixels, (unsigned long long*)out_u64_dev);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}
