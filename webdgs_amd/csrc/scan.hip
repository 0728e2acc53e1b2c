// Exclusive u32 prefix scan (replaces src/prefix/prefix.ts + prefix_sum.wgsl K2-K4: 3-phase Blelloch, <= 2 097 152
// elements).  Here: reduce -> scan of block sums -> down-sweep, 4096 elements per 256-thread block, 16-byte
// loads/stores, wave64 shuffle scans, no element cap.  HBM traffic 12 B/element (read, read, write).
#include "common.h"

namespace {

constexpr u32 SCAN_THREADS = 256;
constexpr u32 SCAN_ITEMS = 16;
constexpr u32 SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 4096

__device__ __forceinline__ u32 wave_inclusive_scan(u32 v, u32 lane) {
#pragma unroll
    for (u32 d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Block-wide exclusive scan of one value per thread (256 threads = 4 waves); returns the exclusive prefix and
// the block total through *total.
__device__ __forceinline__ u32 block_exclusive_scan(u32 v, u32* total, u32* lds /*>= 4 words*/) {
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u32 inc = wave_inclusive_scan(v, lane);
    if (lane == 63u) lds[wave] = inc;
    __syncthreads();
    u32 wave_off = 0, tot = 0;
#pragma unroll
    for (u32 w = 0; w < SCAN_THREADS / 64; w++) {
        const u32 s = lds[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return wave_off + inc - v;
}

__device__ __forceinline__ void load_items(const u32* __restrict__ in, u32 base, u32 count, u32 (&x)[SCAN_ITEMS]) {
    // thread owns SCAN_ITEMS consecutive elements: 4 x uint4
    const u32 first = base + threadIdx.x * SCAN_ITEMS;
    if (first + SCAN_ITEMS <= count && (reinterpret_cast<uintptr_t>(in + first) & 15u) == 0) {
#pragma unroll
        for (u32 j = 0; j < SCAN_ITEMS / 4; j++) {
            const uint4 q = *reinterpret_cast<const uint4*>(in + first + j * 4);
            x[j * 4 + 0] = q.x; x[j * 4 + 1] = q.y; x[j * 4 + 2] = q.z; x[j * 4 + 3] = q.w;
        }
    } else {
#pragma unroll
        for (u32 j = 0; j < SCAN_ITEMS; j++) x[j] = (first + j < count) ? in[first + j] : 0u;
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const u32* __restrict__ in, u32 count, u32* __restrict__ block_sums) {
    __shared__ u32 lds[4];
    u32 x[SCAN_ITEMS];
    load_items(in, blockIdx.x * SCAN_TILE, count, x);
    u32 s = 0;
#pragma unroll
    for (u32 j = 0; j < SCAN_ITEMS; j++) s += x[j];
    u32 total;
    (void)block_exclusive_scan(s, &total, lds);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// In-place exclusive scan of `count` values by ONE workgroup, SCAN_TILE per round with a running carry (16 consecutive values per
// thread); returns the total (uniform).
__device__ __forceinline__ u32 scan_inplace_by_block(u32* __restrict__ v, u32 count, u32* lds) {
    u32 carry = 0;
    for (u32 base = 0; base < count; base += SCAN_TILE) {
        u32 x[SCAN_ITEMS];
        load_items(v, base, count, x);
        u32 s = 0;
#pragma unroll
        for (u32 j = 0; j < SCAN_ITEMS; j++) s += x[j];
        u32 total;
        u32 run = carry + block_exclusive_scan(s, &total, lds);
        const u32 first = base + threadIdx.x * SCAN_ITEMS;
#pragma unroll
        for (u32 j = 0; j < SCAN_ITEMS; j++) {
            if (first + j < count) v[first + j] = run;
            run += x[j];
        }
        carry += total;
    }
    return carry;
}

// The optional epilogue is the reference's update_stats (src/shaders/update-stats.wgsl:19-35) for the tile-count scan of the
// forward pass: the grand total IS the number of tile entries, so the same single-block kernel publishes the stats block
// {entries (clamped to the capacity), visible splats (folded from the shard words, which it clears), overflow} and mirrors it
// into pinned host memory -- one launch less between the projection and the emit.
__device__ __forceinline__ void stats_epilogue(u32 carry, const ScanStatsEpilogue& ep) {
    u32 vis = 0u;
    if (threadIdx.x < 64u) {  // wave 0 folds the 64 visible-count shards
        vis = ep.visible_shards[threadIdx.x];
        ep.visible_shards[threadIdx.x] = 0u;
#pragma unroll
        for (u32 d = 32; d >= 1; d >>= 1) vis += (u32)__shfl_xor((int)vis, (int)d, 64);
    }
    if (threadIdx.x == 0) {
        const u32 entries = min(carry, ep.capacity), overflow = (carry > ep.capacity) ? carry : 0u;  // consumers only touch [0, capacity)
        ep.stats[0] = entries; ep.stats[1] = vis; ep.stats[2] = overflow;
        if (ep.frame) *ep.frame += 1u;   // the frame that project_count has just stamped its non-finite Splats' tiles with
        if (ep.host_mirror) {  // word 2 is STICKY: set on overflow, cleared only by the host check, so no view of a multi-view step can hide another's overflow
            ep.host_mirror[0] = entries; ep.host_mirror[1] = vis; if (overflow) ep.host_mirror[2] = overflow; ep.host_mirror[3] = 0u;
        }
    }
    if (ep.long_hdr && threadIdx.x < 8u) ep.long_hdr[threadIdx.x] = 0u;   // (longlist.h: LL_HDR_WORDS; segment_sort counts the frame's long tiles up from here)
}

// One block scans the block sums in place (exclusive), 256 x 16 at a time with a running carry (the forward pass hands over N/256 of
// them: one round up to 1 M Gaussians).
__global__ __launch_bounds__(SCAN_THREADS) void scan_block_sums_kernel(u32* __restrict__ block_sums, u32 num_blocks, u32* __restrict__ total_out,
                                                                       ScanStatsEpilogue ep) {
    WD_STREAM_PRIO();
    __shared__ u32 lds[4];
    const u32 carry = scan_inplace_by_block(block_sums, num_blocks, lds);
    if (threadIdx.x == 0 && total_out) *total_out = carry;
    if (ep.stats) stats_epilogue(carry, ep);
}

// The forward pass's scans in ONE launch: workgroup 0 is scan_block_sums above (per-workgroup entry counts -> workgroup offsets, stats
// block); workgroup 1 + c scans row c of the per-workgroup tile-COLUMN counts project_count left (column_counts[c][0 .. num_blocks)) into
// the offset of each workgroup's entries inside column c, and leaves the column's total -- what the first pass of a radix sort gets from
// its histogram and row-scan kernels, here without reading a key.
__global__ __launch_bounds__(SCAN_THREADS) void forward_scan_kernel(u32* __restrict__ block_sums, u32 num_blocks, u32* __restrict__ column_counts,
                                                                    u32* __restrict__ column_totals, ScanStatsEpilogue ep) {
    WD_STREAM_PRIO();
    __shared__ u32 lds[4];
    if (blockIdx.x == 0u) {
        const u32 carry = scan_inplace_by_block(block_sums, num_blocks, lds);
        stats_epilogue(carry, ep);
    } else {
        const u32 c = blockIdx.x - 1u;
        const u32 total = scan_inplace_by_block(column_counts + (size_t)c * num_blocks, num_blocks, lds);
        if (threadIdx.x == 0u) column_totals[c] = total;
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_downsweep_kernel(const u32* __restrict__ in, u32* __restrict__ out, u32 count,
                                                                      const u32* __restrict__ block_sums) {
    __shared__ u32 lds[4];
    u32 x[SCAN_ITEMS];
    const u32 base = blockIdx.x * SCAN_TILE;
    load_items(in, base, count, x);
    u32 s = 0;
#pragma unroll
    for (u32 j = 0; j < SCAN_ITEMS; j++) s += x[j];
    u32 total;
    u32 run = block_exclusive_scan(s, &total, lds) + block_sums[blockIdx.x];
    const u32 first = base + threadIdx.x * SCAN_ITEMS;
    u32 y[SCAN_ITEMS];
#pragma unroll
    for (u32 j = 0; j < SCAN_ITEMS; j++) { y[j] = run; run += x[j]; }
    if (first + SCAN_ITEMS <= count && (reinterpret_cast<uintptr_t>(out + first) & 15u) == 0) {
#pragma unroll
        for (u32 j = 0; j < SCAN_ITEMS / 4; j++)
            *reinterpret_cast<uint4*>(out + first + j * 4) = make_uint4(y[j * 4], y[j * 4 + 1], y[j * 4 + 2], y[j * 4 + 3]);
    } else {
#pragma unroll
        for (u32 j = 0; j < SCAN_ITEMS; j++)
            if (first + j < count) out[first + j] = y[j];
    }
}

}  // namespace

int scan_scratch_create(ScanScratch* s, u32 max_elements) {
    s->capacity_blocks = ceil_div(max_elements > 0 ? max_elements : 1, SCAN_TILE);
    return wdgs_alloc((void**)&s->block_sums, sizeof(u32) * (size_t)(s->capacity_blocks + 1), true, nullptr);
}

void scan_scratch_destroy(ScanScratch* s) {
    if (s->block_sums) wdgs_free(s->block_sums);
    s->block_sums = nullptr;
    s->capacity_blocks = 0;
}

// In-place exclusive scan of `num_blocks` per-workgroup sums by one workgroup, with the forward pass's stats epilogue: the middle
// level of a scan whose first level (the sums) and last level (the in-workgroup prefix) live in the producer and consumer kernels.
int scan_block_sums_inplace(wdgs_device* dev, u32* block_sums, u32 num_blocks, const ScanStatsEpilogue& ep) {
    WDGS_LAUNCH(dev, "scan_block_sums", scan_block_sums_kernel, dim3(1), dim3(SCAN_THREADS), 0, block_sums, num_blocks, (u32*)nullptr, ep);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int forward_scan(wdgs_device* dev, u32* block_sums, u32 num_blocks, u32* column_counts, u32* column_totals, u32 columns, const ScanStatsEpilogue& ep) {
    WDGS_LAUNCH(dev, "scan_forward", forward_scan_kernel, dim3(1u + columns), dim3(SCAN_THREADS), 0, block_sums, num_blocks, column_counts, column_totals, ep);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}

int scan_exclusive_u32(wdgs_device* dev, ScanScratch* s, const u32* in, u32* out, u32 count, u32* total_out) {
    return scan_exclusive_u32_stats(dev, s, in, out, count, total_out, ScanStatsEpilogue{nullptr, nullptr, nullptr, 0u, nullptr, nullptr});
}

int scan_exclusive_u32_stats(wdgs_device* dev, ScanScratch* s, const u32* in, u32* out, u32 count, u32* total_out, const ScanStatsEpilogue& ep) {
    if (count == 0) {
        if (total_out) WDGS_CHECK_HIP(hipMemsetAsync(total_out, 0, 4, dev->stream));
        return WDGS_OK;
    }
    const u32 blocks = ceil_div(count, SCAN_TILE);
    WDGS_REQUIRE(blocks <= s->capacity_blocks, WDGS_E_CAPACITY, "scan: %u elements exceed the scanner's capacity (%u blocks)", count, s->capacity_blocks);
    WDGS_LAUNCH(dev, "scan_reduce", scan_reduce_kernel, dim3(blocks), dim3(SCAN_THREADS), 0, in, count, s->block_sums);
    WDGS_LAUNCH(dev, "scan_block_sums", scan_block_sums_kernel, dim3(1), dim3(SCAN_THREADS), 0, s->block_sums, blocks, total_out, ep);
    WDGS_LAUNCH(dev, "scan_downsweep", scan_downsweep_kernel, dim3(blocks), dim3(SCAN_THREADS), 0, in, out, count, s->block_sums);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}
