// Where does a trip of sort.hip's seg_pass_global (an oversized tile segment, one workgroup, 1 024 entries per trip) spend its time?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/seg_pass_phases scripts/microbench/seg_pass_phases.hip && /tmp/seg_pass_phases
// One workgroup of 256 threads ranks n (key, value) pairs on the low byte of the key exactly as the product loop does; wave 0's first lane reads the
// 100 MHz wall clock (s_memrealtime) at the phase boundaries and adds up the differences over the trips.  Variants: the keys all equal (the late
// regime's pile) or random; with and without the global stores; with and without the loads of the next trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32;
constexpr u32 T = 256, RADIX = 256;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool STORES, bool LOADS>
__global__ __launch_bounds__(256) void pass(const u32* __restrict__ src_k, const u32* __restrict__ src_v, u32* __restrict__ dst_k, u32* __restrict__ dst_v, u32 n, u32 shift,
                                           unsigned long long* rec) {
    __shared__ unsigned short cnt16[RADIX * 16];
    __shared__ u32 s_base[RADIX];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    s_base[threadIdx.x] = threadIdx.x * (n / RADIX);   // (any bases: the positions only have to stay inside the buffer)
    __syncthreads();
    uint4* const my_counts = reinterpret_cast<uint4*>(cnt16 + threadIdx.x * 16u);
    u32 kq[4], vq[4];
    for (u32 j = 0; j < 4u; j++) { const u32 i = j * T + threadIdx.x; kq[j] = (i < n) ? src_k[i] : 0xFFFFFFFFu; vq[j] = (i < n) ? src_v[i] : 0u; }
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = wall_clock64();
    for (u32 c0 = 0; c0 < n; c0 += 4u * T) {
        unsigned long long t0 = wall_clock64();
        u32 kn[4], vn[4];
#pragma unroll
        for (u32 j = 0; j < 4u; j++) {
            const u32 i_next = c0 + (4u + j) * T + threadIdx.x;
            kn[j] = (LOADS && i_next < n) ? src_k[i_next] : (kq[j] * 1664525u + 1013904223u);
            vn[j] = (LOADS && i_next < n) ? src_v[i_next] : vq[j] + 1u;
        }
        my_counts[0] = make_uint4(0u, 0u, 0u, 0u);
        my_counts[1] = make_uint4(0u, 0u, 0u, 0u);
        u32 below[4], group[4];
#pragma unroll
        for (u32 j = 0; j < 4u; j++) {
            const bool valid = c0 + j * T + threadIdx.x < n;
            const u32 digit = (kq[j] >> shift) & (RADIX - 1u);
            unsigned long long m = __ballot(valid);
#pragma unroll
            for (u32 b = 0; b < 8; b++) {
                const bool bit = (digit >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            below[j] = (u32)__popcll(m & lt_mask);
            group[j] = (valid && below[j] == 0u) ? (u32)__popcll(m) : 0u;
        }
        unsigned long long t1 = wall_clock64(); ph[0] += t1 - t0;
        lds_barrier();
        t0 = wall_clock64(); ph[1] += t0 - t1;
#pragma unroll
        for (u32 j = 0; j < 4u; j++)
            if (group[j] != 0u) cnt16[((kq[j] >> shift) & (RADIX - 1u)) * 16u + j * 4u + wave] = (unsigned short)group[j];
        lds_barrier();
        t1 = wall_clock64(); ph[2] += t1 - t0;
        u32 trip_total = 0u;
        {
            uint4 q[2] = {my_counts[0], my_counts[1]};
            u32* const w = reinterpret_cast<u32*>(q);
#pragma unroll
            for (u32 x = 0; x < 8u; x++) {
                const u32 lo = w[x] & 0xFFFFu, hi = w[x] >> 16u;
                w[x] = trip_total | ((trip_total + lo) << 16u);
                trip_total += lo + hi;
            }
            my_counts[0] = q[0];
            my_counts[1] = q[1];
        }
        lds_barrier();
        t0 = wall_clock64(); ph[3] += t0 - t1;
#pragma unroll
        for (u32 j = 0; j < 4u; j++) {
            if (c0 + j * T + threadIdx.x < n) {
                const u32 digit = (kq[j] >> shift) & (RADIX - 1u);
                const u32 pos = (s_base[digit] + cnt16[digit * 16u + j * 4u + wave] + below[j]) % n;
                if (STORES) { dst_k[pos] = kq[j]; dst_v[pos] = vq[j]; } else if (pos == 0xFFFFFFF0u) dst_k[0] = pos;
            }
        }
        lds_barrier();
        t1 = wall_clock64(); ph[4] += t1 - t0;
        s_base[threadIdx.x] += trip_total;
#pragma unroll
        for (u32 j = 0; j < 4u; j++) { kq[j] = kn[j]; vq[j] = vn[j]; }
        ph[5] += wall_clock64() - t1;
    }
    const unsigned long long t_end = wall_clock64();
    if (threadIdx.x == 0) { for (int i = 0; i < 6; i++) rec[i] = ph[i]; rec[6] = t_end - t_begin; }
}

int main() {
    const u32 n = 40000;
    std::vector<u32> hk(n), hv(n);
    u32 *sk, *sv, *dk, *dv; unsigned long long* rec;
    hipMalloc(&sk, 4 * n); hipMalloc(&sv, 4 * n); hipMalloc(&dk, 4 * n); hipMalloc(&dv, 4 * n); hipMalloc(&rec, 64);
    const char* names[6] = {"loads issued, counts cleared, 4 x 8 ballots", "barrier 1", "leaders' counts + barrier 2", "prefix in registers + barrier 3", "positions, stores issued + barrier 4",
                            "bases advanced, registers rotated"};
    for (int keys = 0; keys < 2; keys++) {
        for (u32 i = 0; i < n; i++) { hk[i] = keys ? (u32)rand() : 0x1234FFC0u; hv[i] = i; }
        hipMemcpy(sk, hk.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(sv, hv.data(), 4 * n, hipMemcpyHostToDevice);
        for (int variant = 0; variant < 3; variant++) {
            unsigned long long h[7];
            for (int rep = 0; rep < 3; rep++) {
                if (variant == 0) hipLaunchKernelGGL((pass<true, true>), dim3(1), dim3(256), 0, 0, sk, sv, dk, dv, n, 0u, rec);
                else if (variant == 1) hipLaunchKernelGGL((pass<false, true>), dim3(1), dim3(256), 0, 0, sk, sv, dk, dv, n, 0u, rec);
                else hipLaunchKernelGGL((pass<false, false>), dim3(1), dim3(256), 0, 0, sk, sv, dk, dv, n, 0u, rec);
                hipDeviceSynchronize();
                hipMemcpy(h, rec, 56, hipMemcpyDeviceToHost);
            }
            const double trips = (n + 1023) / 1024;
            printf("%s keys, %s: %.2f us per trip of 1 024 entries (%.1f us for the pass)\n", keys ? "random" : "all equal",
                   variant == 0 ? "as in the product" : (variant == 1 ? "no global stores" : "no global stores, no loads"), h[6] * 0.01 / trips, h[6] * 0.01);
            for (int i = 0; i < 6; i++) printf("    %-52s %.3f us per trip\n", names[i], h[i] * 0.01 / trips);
        }
    }
    return 0;
}
