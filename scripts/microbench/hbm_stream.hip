// Microbenchmark (a measurement tool, not part of the product): what HBM bandwidth do plain streaming kernels reach on this box at the
// sizes and shapes of the training step's bandwidth-bound kernels?  Copy and read-only passes over 64 MB - 1 GB, 16 bytes per lane,
// one element per thread and grid-stride forms, plus a strided row form (a 112-byte row per lane, as geometry_backward_adam reads).
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/hbm_stream.hip -o /tmp/hbm_stream && /tmp/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void copy_flat(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
__global__ void copy_stride(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void copy_unroll4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    const size_t base = ((size_t)blockIdx.x * blockDim.x) * 4 + threadIdx.x;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) if (base + k * blockDim.x < n) v[k] = a[base + k * blockDim.x];
#pragma unroll
    for (int k = 0; k < 4; k++) if (base + k * blockDim.x < n) b[base + k * blockDim.x] = v[k];
}
__global__ void read_flat(const float4* __restrict__ a, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 v = make_float4(0, 0, 0, 0);
    if (i < n) v = a[i];
    if (v.x + v.y + v.z + v.w == 123.456f) out[0] = 1.0f;
}
// one 112-byte row per lane read and written back (7 x float4 at a 112-byte stride)
__global__ void rows112(float4* __restrict__ a, size_t rows) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float4* p = a + r * 7;
    float4 v[7];
#pragma unroll
    for (int k = 0; k < 7; k++) v[k] = p[k];
#pragma unroll
    for (int k = 0; k < 7; k++) { v[k].x += 1.0f; p[k] = v[k]; }
}

template <typename F> float time_ms(F f, int reps = 5) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f(); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CHECK(hipEventRecord(e0)); f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t max_bytes = 1ull << 30;
    float4 *a, *b; float* out;
    CHECK(hipMalloc(&a, max_bytes)); CHECK(hipMalloc(&b, max_bytes)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(a, 0, max_bytes)); CHECK(hipMemset(b, 0, max_bytes));
    for (size_t mb : {64, 128, 256, 512, 1024}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        const unsigned blocks = (unsigned)((n + 255) / 256);
        float t1 = time_ms([&] { hipLaunchKernelGGL(copy_flat, dim3(blocks), dim3(256), 0, 0, a, b, n); });
        float t2 = time_ms([&] { hipLaunchKernelGGL(copy_stride, dim3(256 * 16), dim3(256), 0, 0, a, b, n); });
        float t3 = time_ms([&] { hipLaunchKernelGGL(copy_unroll4, dim3((blocks + 3) / 4), dim3(256), 0, 0, a, b, n); });
        float t4 = time_ms([&] { hipLaunchKernelGGL(read_flat, dim3(blocks), dim3(256), 0, 0, a, out, n); });
        const size_t rows = bytes / 112;
        float t5 = time_ms([&] { hipLaunchKernelGGL(rows112, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, 0, a, rows); });
        printf("%5zu MB: copy flat %7.1f us %5.2f TB/s | grid-stride %7.1f us %5.2f TB/s | 4 per thread %7.1f us %5.2f TB/s | read only %7.1f us %5.2f TB/s | 112-byte rows r+w %7.1f us %5.2f TB/s\n",
               mb, t1 * 1e3, 2.0 * bytes / t1 / 1e9, t2 * 1e3, 2.0 * bytes / t2 / 1e9, t3 * 1e3, 2.0 * bytes / t3 / 1e9, t4 * 1e3, 1.0 * bytes / t4 / 1e9, t5 * 1e3,
               2.0 * rows * 112 / t5 / 1e9);
    }
    return 0;
}
