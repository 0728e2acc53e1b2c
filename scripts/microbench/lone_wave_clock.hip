// What clock does a wave run at when it is (nearly) alone on the chip?   hipcc --offload-arch=gfx950 -O3 -o /tmp/lone_wave_clock scripts/microbench/lone_wave_clock.hip
// One launch of <waves> single-wave workgroups, each 1 / 2 / 4 / 8 interleaved chains of dependent v_fma_f32 (N instructions in all): per wave the shader-clock counter (s_memtime) and the 100 MHz
// wall clock (s_memrealtime) around the chain.  cycles per instruction = shader cycles / N; shader clock = shader cycles / wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// CHAINS independent chains of dependent v_fma_f32, interleaved instruction by instruction: n instructions in all
template <int CHAINS>
__global__ void chain(float* out, unsigned long long* rec, int n) {
    float x[CHAINS];
    for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x * 1e-3f + c;
    const float a = 1.0000001f, b = 1e-7f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i += 8 * CHAINS) {
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float sum = 0.f;
    for (int c = 0; c < CHAINS; c++) sum += x[c];
    out[blockIdx.x * 64 + threadIdx.x] = sum;
    if (threadIdx.x == 0) { rec[blockIdx.x * 2] = c1 - c0; rec[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main() {
    const int n = 1 << 16;
    for (int chains : {1, 2, 4, 8})
    for (int waves : {1, 1024, 8192}) {
        float* out; unsigned long long* rec;
        hipMalloc(&out, sizeof(float) * 64 * waves); hipMalloc(&rec, 16 * waves);
        for (int rep = 0; rep < 3; rep++) {
            if (chains == 1) hipLaunchKernelGGL(chain<1>, dim3(waves), dim3(64), 0, 0, out, rec, n);
            else if (chains == 2) hipLaunchKernelGGL(chain<2>, dim3(waves), dim3(64), 0, 0, out, rec, n);
            else if (chains == 4) hipLaunchKernelGGL(chain<4>, dim3(waves), dim3(64), 0, 0, out, rec, n);
            else hipLaunchKernelGGL(chain<8>, dim3(waves), dim3(64), 0, 0, out, rec, n);
        }
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(2 * waves);
        hipMemcpy(h.data(), rec, 16 * waves, hipMemcpyDeviceToHost);
        std::vector<double> cyc, mhz;
        for (int i = 0; i < waves; i++) { cyc.push_back((double)h[2 * i] / n); mhz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 0.01)); }
        std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
        printf("%d chains, %6d waves: shader-clock cycles per v_fma_f32 of a wave (median) %.2f;  shader clock seen by a wave (median) %.0f MHz;  ns per instruction %.2f\n", chains, waves, cyc[waves / 2], mhz[waves / 2],
               cyc[waves / 2] / mhz[waves / 2] * 1e3);
        hipFree(out); hipFree(rec);
    }
    return 0;
}
