// Stable LSD radix sort of (key u32, value u32) pairs with a device-resident element count, and the per-tile range
// table built from the sorted keys.
//
// Replaces src/sort/sort_dynamic.ts + radix_sort.wgsl (K7-K11: 4 x 8-bit passes with decoupled look-back whose rank
// loop assumes 32 lanes run in lock-step -- SURVEY section 5) and src/shaders/tile-ranges.wgsl (K12-K13: E atomicMins).
//
// Per significant 8-bit digit (reduce-then-scan; a decoupled look-back variant was measured slower here, because at
// E ~ 6 M all ~1500 partitions are co-resident and every one walks hundreds of unfinished predecessors):
//   sort_hist      per-partition digit counts: 16-byte key loads, wave-private LDS bins -> counts[digit][partition]
//   sort_scan_rows one workgroup per digit: exclusive scan of its row over the ACTIVE partitions (count read on device)
//   (the exclusive scan of the 256 row totals is done by every scatter block itself)
//   sort_scatter   ranks by wave64 ballot match (stable: order = wave, round, lane = input order) and scatters
// HBM traffic per pass: 4E (hist) + 16E (scatter); ranges: 4E + 4(T+1).  Only digits that can be non-zero are sorted.
//
// Tile-structured keys (the forward pass: key = (tile + 1) << 16 | depth16) take a shorter route, sort_segmented below:
//   1. the stable LSD passes above on the TILE bits only (2 passes for up to 65535 tiles instead of 4 over the whole key) -- entries of
//      a tile are now contiguous, in emission order (ascending Gaussian index);
//   2. tile_ranges on those keys (the high 16 bits are already in final order);
//   3. segment_sort: one workgroup per tile sorts its segment by the 16 depth bits, stably, entirely in LDS (two 8-bit counting
//      passes over <= SEG_CAP entries; larger segments run the same two passes through the ping-pong buffers, chunk by chunk).
// The result is the order a stable sort of the full 32-bit key gives -- (tile, depth16, emission order) -- for 4E + 16E per tile
// pass + 16E for the segment sort = 56E bytes instead of 84E, and 9 launches instead of 12.
#include <algorithm>

#include "common.h"
#include "longlist.h"

namespace {

constexpr u32 SORT_THREADS = 256;
// Keys per thread of a partition (ITEMS below): 16 -- partitions of 4096 keys -- or, for a sorter of small capacity, 4: c2's row pass has 88
// workgroups of 4096 keys for 256 CUs, and each lasts as long as one wave's serial ranking of ITEMS rounds; with 1024 keys it has 352.
// Same box (profiles/r06k_sort_partition_size_ab.txt, r06l_*): c2 sort 37.2 -> 32.4 us, 5 800 -> 6 010 it/s.  Large sorters keep 16: with 8, c3 is
// unchanged and c5 (41 M entries) loses 20 % of its sort (more partitions: a longer count table and row scan, shorter runs per write).
constexpr u32 SORT_ITEMS_MAX = 16;
constexpr u32 SORT_TILE_MAX = SORT_THREADS * SORT_ITEMS_MAX;  // keys per partition of the default form; capacities are multiples of it
constexpr u32 RADIX = 256;

// The digit a pass sorts on: bits [shift, shift + width) of the key ((key >> shift) & dmask) or, for the second pass of a forward-pass
// sort (sorter_sort_rows below), the tile ROW of the key's tile field: floor(((key >> 16) - 1) / num_tiles_x) by a reciprocal multiply
// (exact: tile < 2^16, num_tiles_x <= 256).  row_inv = 0 selects the bit field; dmask is the largest digit either way.
struct DigitOf {
    u32 shift, dmask, row_inv;
    // (row mode: a key whose tile field is 0 -- no emitted key has one; a corrupted list could -- would give a row far beyond the grid and index the
    // digit tables out of range: it is held to the last row, where the bounded range update of sort_scatter ignores it)
    __device__ __forceinline__ u32 operator()(u32 key) const { return row_inv ? min(__umulhi((key >> 16u) - 1u, row_inv), dmask) : ((key >> shift) & dmask); }
};

template <u32 ITEMS>
__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(const u32* __restrict__ keys, const u32* __restrict__ count_ptr, DigitOf digit_of,
                                                                 u32 num_parts, u32* __restrict__ counts /*[RADIX][num_parts]*/, u32* __restrict__ ranges_init,
                                                                 u32 total_tiles, u32* __restrict__ marks_init) {
    WD_STREAM_PRIO();
    constexpr u32 TILE = SORT_THREADS * ITEMS;  // keys per partition
    __shared__ u32 lh[SORT_THREADS / 64][RADIX];
    const u32 count = *count_ptr;
    const u32 part = blockIdx.x;
    const u32 base = part * TILE;
    const u32 dmask = digit_of.dmask;
    // ranges_init (nullable): the per-tile range table the scatter of this pass lowers (sort_scatter, ranges_mode 2) starts out "empty"
    // with its terminator ranges[T] = E -- set here, by the kernel that runs before that scatter (partition 0 runs even for an empty list)
    if (ranges_init) {
        const u32 active = max((count + TILE - 1u) / TILE, 1u);
        if (part < active)
            for (u32 t = part * SORT_THREADS + threadIdx.x; t <= total_tiles; t += active * SORT_THREADS) ranges_init[t] = (t == total_tiles) ? count : 0xFFFFFFFFu;
    }
    // marks_init (nullable): the per-tile marks of the long tile lists (longlist.h: flags) start the frame cleared -- here, among stores that are not on
    // anybody's way, rather than one store per tile in the middle of segment_sort's workgroups; that kernel then marks its long tiles only
    if (marks_init) {
        const u32 active = max((count + TILE - 1u) / TILE, 1u);
        if (part < active)
            for (u32 t = part * SORT_THREADS + threadIdx.x; t < total_tiles; t += active * SORT_THREADS) marks_init[t] = 0u;
    }
    if (base >= count) return;
    const u32 wave = threadIdx.x >> 6;
#pragma unroll
    for (u32 w = 0; w < SORT_THREADS / 64; w++) lh[w][threadIdx.x] = 0;
    __syncthreads();
    if (base + TILE <= count) {
#pragma unroll
        for (u32 j = 0; j < ITEMS / 4; j++) {
            const uint4 q = *reinterpret_cast<const uint4*>(keys + base + (j * SORT_THREADS + threadIdx.x) * 4u);
            // the four keys of a lane are neighbours in memory and, in tile-ordered data, usually share their digit: merge equal
            // digits inside the lane first (same-address LDS atomics of one wave-instruction serialise; they were 86 % of this
            // kernel's LDS cycles: profiles/r01e_pmc.json)
            const u32 d0 = digit_of(q.x), d1 = digit_of(q.y), d2 = digit_of(q.z), d3 = digit_of(q.w);
            if (d0 == d3 && d0 == d1 && d0 == d2) {
                atomicAdd(&lh[wave][d0], 4u);
            } else {
                atomicAdd(&lh[wave][d0], 1u);
                atomicAdd(&lh[wave][d1], 1u);
                atomicAdd(&lh[wave][d2], 1u);
                atomicAdd(&lh[wave][d3], 1u);
            }
        }
    } else {
        for (u32 j = 0; j < ITEMS; j++) {
            const u32 i = base + j * SORT_THREADS + threadIdx.x;
            if (i < count) atomicAdd(&lh[wave][digit_of(keys[i])], 1u);
        }
    }
    __syncthreads();
    // (only the rows of digits that exist: each count is a 4-byte store into a row of its own, i.e. a partial line write)
    if (threadIdx.x <= dmask) counts[(size_t)threadIdx.x * num_parts + part] = lh[0][threadIdx.x] + lh[1][threadIdx.x] + lh[2][threadIdx.x] + lh[3][threadIdx.x];
}

// One workgroup per digit: in-place exclusive scan of counts[digit][0 .. active_parts), row total -> totals[digit].
template <u32 ITEMS>
__global__ __launch_bounds__(256) void sort_scan_rows_kernel(u32* __restrict__ counts, const u32* __restrict__ count_ptr, u32 num_parts,
                                                              u32* __restrict__ totals) {
    WD_STREAM_PRIO();
    constexpr u32 TILE = SORT_THREADS * ITEMS;  // keys per partition
    __shared__ u32 s_w[4];
    const u32 active = (*count_ptr + TILE - 1u) / TILE;
    u32* row = counts + (size_t)blockIdx.x * num_parts;
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u32 carry = 0;
    for (u32 base = 0; base < active; base += 1024u) {
        const u32 i0 = base + threadIdx.x * 4u;
        u32 x[4];
#pragma unroll
        for (u32 j = 0; j < 4; j++) x[j] = (i0 + j < active) ? row[i0 + j] : 0u;
        const u32 tsum = x[0] + x[1] + x[2] + x[3];
        u32 inc = tsum;
#pragma unroll
        for (u32 d = 1; d < 64; d <<= 1) {
            const u32 t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63u) s_w[wave] = inc;
        __syncthreads();
        u32 woff = 0, tot = 0;
#pragma unroll
        for (u32 w = 0; w < 4; w++) { const u32 v = s_w[w]; if (w < wave) woff += v; tot += v; }
        __syncthreads();
        u32 run = carry + woff + inc - tsum;
#pragma unroll
        for (u32 j = 0; j < 4; j++) {
            if (i0 + j < active) row[i0 + j] = run;
            run += x[j];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

template <u32 ITEMS>
__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(const u32* __restrict__ keys_in, const u32* __restrict__ vals_in,
                                                                    u32* __restrict__ keys_out, u32* __restrict__ vals_out,
                                                                    const u32* __restrict__ count_ptr, DigitOf digit_of, u32 num_parts,
                                                                    const u32* __restrict__ offsets /*scanned rows*/, const u32* __restrict__ digit_totals,
                                                                    u32* __restrict__ ranges, u32 ranges_mode, u32 total_tiles) {
    WD_STREAM_PRIO();
    constexpr u32 TILE = SORT_THREADS * ITEMS;  // keys per partition
    __shared__ u32 whist[SORT_THREADS / 64][RADIX];
    const u32 dmask = digit_of.dmask;
    const u32 count = *count_ptr;
    // neighbouring partitions write neighbouring slices of every digit run: the ACTIVE ones (the grid is sized for the capacity) are
    // numbered so that neighbours share an XCD (common.h)
    const u32 active_parts = max((count + TILE - 1u) / TILE, 1u);  // partition 0 runs even for an empty list
    if (blockIdx.x >= active_parts) return;
    const u32 part = xcd_contiguous(blockIdx.x, active_parts);
    const u32 base = part * TILE;
    // The per-tile range table of a tile-structured sort (sort_segmented) is built by the two scatter passes themselves: the first one
    // sets every entry to "empty" (and the terminator ranges[T] = E), the second one -- whose output is in tile order -- lowers
    // ranges[tile] to the first position it writes for that tile (below).  ranges_mode: 0 none, 1 initialise, 2 lower.
    if (ranges_mode == 1u) {
        const u32 active = max((count + TILE - 1u) / TILE, 1u);  // partition 0 runs even for an empty list
        if (part < active)
            for (u32 t = part * SORT_THREADS + threadIdx.x; t <= total_tiles; t += active * SORT_THREADS) ranges[t] = (t == total_tiles) ? count : 0xFFFFFFFFu;
    }
    if (base >= count) return;
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // (requested here, with the keys, although they are used after the ranking: behind the barriers below they would be a second
    // dependent memory round trip in the middle of a workgroup whose whole life is ~15 us)
    const u32 tot_d_early = (threadIdx.x <= dmask) ? digit_totals[threadIdx.x] : 0u;  // (rows above the digit range are neither counted nor scanned)
    const u32 off_d_early = (threadIdx.x <= dmask) ? offsets[(size_t)threadIdx.x * num_parts + part] : 0u;
#pragma unroll
    for (u32 w = 0; w < SORT_THREADS / 64; w++) whist[w][threadIdx.x] = 0;
    __syncthreads();

    // Wave w owns keys [base + w*1024, base + (w+1)*1024) in 16 rounds of 64 consecutive keys: the order
    // (wave, round, lane) is the input order, which is what makes the pass stable.
    u32 k[ITEMS], v[ITEMS], rk[ITEMS];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (u32 j = 0; j < ITEMS; j++) {
        const u32 i = base + wave * (ITEMS * 64u) + j * 64u + lane;
        const bool valid = i < count;
        k[j] = valid ? keys_in[i] : 0xFFFFFFFFu;
        v[j] = valid ? vals_in[i] : 0u;
    }
#pragma unroll
    for (u32 j = 0; j < ITEMS; j++) {
        const u32 i = base + wave * (ITEMS * 64u) + j * 64u + lane;
        const bool valid = i < count;
        const u32 digit = valid ? digit_of(k[j]) : 0u;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (u32 b = 0; b < 8; b++) {
            const bool bit = (digit >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        // m = valid lanes of this wave holding the same digit
        const u32 pre = whist[wave][digit];
        const u32 below = (u32)__popcll(m & lt_mask);
        rk[j] = pre + below;
        if (valid && below == 0u) whist[wave][digit] = pre + (u32)__popcll(m);  // group leader bumps the wave counter
        // wave-private LDS words: the next round's read is ordered after this write within the wave
    }
    __syncthreads();
    // Reorder the partition in LDS so that it leaves in digit runs: position in the partition's sorted order =
    // (digits below) + (same digit in earlier waves) + rank in own wave.  A direct scatter writes each run 4 B at a time from 16
    // different wave-instructions (write amplification 1.9x measured); from LDS consecutive lanes write consecutive addresses.
    __shared__ u32 s_keys[TILE];
    __shared__ u32 s_vals[TILE];
    __shared__ u32 s_gdelta[RADIX];   // global base of digit d minus its start in the partition's sorted order
    __shared__ u32 s_wsum[SORT_THREADS / 64];
    {
        const u32 d = threadIdx.x;
        u32 cnt_d = 0;
#pragma unroll
        for (u32 w = 0; w < SORT_THREADS / 64; w++) cnt_d += whist[w][d];
        // exclusive scan of cnt_d over the 256 digits (4 waves x 64 lanes)
        u32 inc = cnt_d;
#pragma unroll
        for (u32 s = 1; s < 64; s <<= 1) {
            const u32 t = __shfl_up(inc, s, 64);
            if (lane >= s) inc += t;
        }
        if (lane == 63u) s_wsum[wave] = inc;
        __syncthreads();
        u32 woff = 0;
#pragma unroll
        for (u32 w = 0; w < SORT_THREADS / 64; w++) if (w < wave) woff += s_wsum[w];
        const u32 local_start = woff + inc - cnt_d;
        // global base of digit d = exclusive scan of the 256 digit totals, done here by every block (a dozen instructions)
        // rather than by a 1-block kernel between the row scan and the scatter (a 4.5 us bubble per pass)
        const u32 tot_d = tot_d_early;
        u32 tinc = tot_d;
#pragma unroll
        for (u32 s = 1; s < 64; s <<= 1) {
            const u32 t = __shfl_up(tinc, s, 64);
            if (lane >= s) tinc += t;
        }
        __syncthreads();  // s_wsum is re-used
        if (lane == 63u) s_wsum[wave] = tinc;
        __syncthreads();
        u32 tbase = tinc - tot_d;
#pragma unroll
        for (u32 w = 0; w < SORT_THREADS / 64; w++) if (w < wave) tbase += s_wsum[w];
        s_gdelta[d] = tbase + off_d_early - local_start;
        u32 run = local_start;   // per-wave start of digit d inside the partition's sorted order
#pragma unroll
        for (u32 w = 0; w < SORT_THREADS / 64; w++) {
            const u32 c = whist[w][d];
            whist[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (u32 j = 0; j < ITEMS; j++) {
        const u32 i = base + wave * (ITEMS * 64u) + j * 64u + lane;
        if (i < count) {
            const u32 digit = digit_of(k[j]);
            const u32 lpos = whist[wave][digit] + rk[j];
            s_keys[lpos] = k[j];
            s_vals[lpos] = v[j];
        }
    }
    __syncthreads();
    const u32 n_here = min(TILE, count - base);
#pragma unroll 4
    for (u32 e = threadIdx.x; e < n_here; e += SORT_THREADS) {
        const u32 key = s_keys[e];
        const u32 pos = s_gdelta[digit_of(key)] + e;
        keys_out[pos] = key;
        vals_out[pos] = s_vals[e];
        // In the partition's sorted order the entries of one tile are neighbours (same digit, and the input of this pass is ordered by
        // the tile's low bits), so a tile's first entry here is the one whose predecessor belongs to another tile: about 130 per
        // partition.  The smallest of those positions over all partitions is where the tile starts.
        // (a key whose tile field is 0 or beyond the grid cannot come out of emit; the bound keeps a corrupted list -- entries counted but never
        // written -- from turning into an address 16 GB past the table)
        if (ranges_mode == 2u && (e == 0u || (s_keys[e - 1u] >> 16u) != (key >> 16u)) && (key >> 16u) - 1u < total_tiles) atomicMin(&ranges[(key >> 16u) - 1u], pos);
    }
}

// ranges[t] = first index whose key>>16 == t+1, 0xFFFFFFFF for empty tiles, ranges[T] = E  (tile-ranges.wgsl:34-76 writes the same
// table with an init pass + one atomicMin per entry).  The keys are sorted by their high 16 bits, so each tile finds its start by a
// lower-bound search: one WAVE per tile probes 64 positions per round (a 64-ary search: 4 dependent memory round trips at 6 M entries
// instead of 23) -- one launch over T+1 waves instead of an init launch plus a pass over all E keys.
__global__ __launch_bounds__(256) void tile_ranges_kernel(const u32* __restrict__ keys, const u32* __restrict__ count_ptr, u32 total_tiles,
                                                           u32* __restrict__ ranges) {
    WD_STREAM_PRIO();
    const u32 t = blockIdx.x * 4u + (threadIdx.x >> 6);  // wave index = tile
    const u32 lane = threadIdx.x & 63u;
    if (t > total_tiles) return;
    const u32 count = *count_ptr;
    if (t == total_tiles) { if (lane == 0u) ranges[t] = count; return; }
    const u32 want = t + 1u;  // tile field of the key is 1-based
    u32 lo = 0u, hi = count;  // invariant: every index < lo has tile < want; every index >= hi has tile >= want (or hi == count)
    while (hi - lo > 64u) {
        const u32 span = hi - lo;
        // probe positions lo + ceil(span * (l + 1) / 65) - 1 ... strictly inside [lo, hi): 64 distinct, increasing with the lane
        const u32 pos = lo + (u32)(((unsigned long long)span * (lane + 1u)) / 65ull);
        const bool less = (keys[pos] >> 16u) < want;
        const unsigned long long m = __ballot(less);  // monotone: a prefix of ones
        const u32 k = (u32)__popcll(m);               // lanes [0, k) are "less"
        const u32 pos_prev = (k == 0u) ? lo : lo + (u32)(((unsigned long long)span * k) / 65ull) + 1u;       // one past the last "less" probe
        const u32 pos_next = (k == 64u) ? hi : lo + (u32)(((unsigned long long)span * (k + 1u)) / 65ull);   // the first "not less" probe
        lo = pos_prev;
        hi = pos_next;
    }
    // final window of <= 64 candidates [lo, hi): first index with tile >= want
    const u32 idx = lo + lane;
    const bool ge = idx < hi && (keys[idx] >> 16u) >= want;
    const unsigned long long m = __ballot(ge);
    const u32 first = (m == 0ull) ? hi : lo + (u32)__builtin_ctzll(m);
    if (lane == 0u) ranges[t] = (first < count && (keys[first] >> 16u) == want) ? first : 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------------------------------------------- segment sort
constexpr u32 SEG_CAP = 2048;                    // entries sorted entirely in LDS
constexpr u32 SEG_THREADS = 256;

// One stable counting pass over n <= SEG_CAP (key, value) pairs on a digit of BITS bits: the pairs arrive in REGISTERS (kk, vv: element
// index i = wave*per_wave + j*64 + lane) and leave in LDS dst at their ranked position.  digit = (((key & 0xFFFF) - sub) >> shift) &
// (2^BITS - 1).  Wave w owns the contiguous index range [w*per_wave, (w+1)*per_wave) and walks it in rounds of 64 lanes, so (wave,
// round, lane) is the index order -- which makes the ranking stable -- exactly as sort_scatter does for a global partition.  The
// caller guarantees that nobody still reads dst.
constexpr u32 SEG_ROUNDS = SEG_CAP / SEG_THREADS;
constexpr u32 SEG_WIDE_BITS = 10;              // one pass sorts a segment whose depth span is below 2^10 (z within a factor ~256)
constexpr u32 SEG_BINS = 1u << SEG_WIDE_BITS;  // bins held per wave in LDS
typedef unsigned short seg_hist_t;             // counts and positions are <= SEG_CAP: 16 bits keep the four per-wave tables at 8 KB
template <u32 BITS>
__device__ __forceinline__ void seg_pass_regs(const u32 (&kk)[SEG_ROUNDS], const u32 (&vv)[SEG_ROUNDS], u32* __restrict__ dst_k, u32* __restrict__ dst_v, u32 n,
                                              u32 per_wave, u32 sub, u32 shift, seg_hist_t (*whist)[SEG_BINS], u32* s_wsum) {
    constexpr u32 BINS = 1u << BITS, PER_THREAD = BINS / SEG_THREADS;  // digits per thread in the scan (1 or 4)
    static_assert(BINS <= SEG_BINS && BINS % SEG_THREADS == 0, "digit width");
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u32 rounds = per_wave / 64u;  // <= SEG_ROUNDS, uniform per workgroup
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    auto digit_of = [&](u32 key) { return (((key & 0xFFFFu) - sub) >> shift) & (BINS - 1u); };
#pragma unroll
    for (u32 w = 0; w < SEG_THREADS / 64; w++)
#pragma unroll
        for (u32 j = 0; j < PER_THREAD; j++) whist[w][threadIdx.x * PER_THREAD + j] = 0;
    __syncthreads();
    u32 rk[SEG_ROUNDS];
#pragma unroll
    for (u32 j = 0; j < SEG_ROUNDS; j++) {
        if (j < rounds) {
            const u32 i = wave * per_wave + j * 64u + lane;
            const bool valid = i < n;
            const u32 digit = digit_of(kk[j]);
            unsigned long long m = __ballot(valid);
#pragma unroll
            for (u32 b = 0; b < BITS; b++) {
                const bool bit = (digit >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            const u32 pre = whist[wave][digit];
            const u32 below = (u32)__popcll(m & lt_mask);
            rk[j] = pre + below;
            if (valid && below == 0u) whist[wave][digit] = (seg_hist_t)(pre + (u32)__popcll(m));
        }
    }
    __syncthreads();
    {   // exclusive scan over the digits of the per-digit totals (PER_THREAD consecutive digits per thread); per-wave starts per digit
        u32 cnt[PER_THREAD], tsum = 0;
#pragma unroll
        for (u32 j = 0; j < PER_THREAD; j++) {
            const u32 d = threadIdx.x * PER_THREAD + j;
            cnt[j] = 0;
#pragma unroll
            for (u32 w = 0; w < SEG_THREADS / 64; w++) cnt[j] += whist[w][d];
            tsum += cnt[j];
        }
        u32 inc = tsum;
#pragma unroll
        for (u32 sft = 1; sft < 64; sft <<= 1) {
            const u32 t = __shfl_up(inc, sft, 64);
            if (lane >= sft) inc += t;
        }
        if (lane == 63u) s_wsum[wave] = inc;
        __syncthreads();
        u32 run = inc - tsum;
#pragma unroll
        for (u32 w = 0; w < SEG_THREADS / 64; w++) if (w < wave) run += s_wsum[w];
#pragma unroll
        for (u32 j = 0; j < PER_THREAD; j++) {
            const u32 d = threadIdx.x * PER_THREAD + j;
#pragma unroll
            for (u32 w = 0; w < SEG_THREADS / 64; w++) {
                const u32 c = whist[w][d];
                whist[w][d] = (seg_hist_t)run;
                run += c;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (u32 j = 0; j < SEG_ROUNDS; j++) {
        if (j < rounds) {
            const u32 i = wave * per_wave + j * 64u + lane;
            if (i < n) {
                const u32 pos = (u32)whist[wave][digit_of(kk[j])] + rk[j];
                dst_k[pos] = kk[j];
                dst_v[pos] = vv[j];
            }
        }
    }
    __syncthreads();
}

// A workgroup barrier that orders LDS accesses only (s_barrier behind a wait for the wave's own LDS operations): global loads and stores stay in flight.
WD_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Segments longer than SEG_CAP: the same two stable passes, through global memory.  pass: histogram of the whole segment, exclusive scan, then the
// segment in trips of 1 024 entries -- thread t holds entries c0 + j * 256 + t, j = 0..3 (index order = (j, wave, lane)) -- each ranked stably on top
// of running digit bases.
//
// The first form of this pass ranked ONE chunk of 256 per trip: four waves, one per SIMD, each a chain of a dozen dependent LDS round trips and four
// __syncthreads (which also wait for the trip's global stores), 0.75 us per chunk -- 92 us for the 15 000 entries of the tile in which a long run of the
// reference's schedule collects its non-finite Gaussians (profiles/r08p_late_regime.txt), 293 us for a 40 000-entry tile.  What was tried on the way
// and did not matter by itself is in profiles/r09i_*, r09j_*: loads requested ahead, LDS-only barriers, a histogram free of same-address atomics.  What
// matters is the number of LDS round trips one after the other per entry.  Now: the four chunks' ballot matches are four independent chains; the
// counts of the trip's sixteen (chunk, wave) groups of a digit lie side by side, [digit][16] of 16 bits, so that thread d turns digit d's counts into
// positions with two 16-byte reads, a prefix sum in registers and two 16-byte writes; a trip has four barriers, which wait for LDS only; the next
// trip's pairs are requested at the top of this one.  (15 000 entries: 92 -> 71 us; 40 000: 293 -> 208, profiles/r10c_*.  A trip of 1 024 entries still takes 2.4-2.7 us, and 2.0 of them with every global
// access taken out (scripts/microbench/seg_pass_phases.hip, profiles/r11c_seg_pass_phases.txt): the trip is ~450 instructions of a wave that is alone
// on its SIMD, and such a wave issues one every 5-10 cycles.  More entries per second from here means more waves: several workgroups per segment.)
__device__ void seg_pass_global(const u32* __restrict__ src_k, const u32* __restrict__ src_v, u32* __restrict__ dst_k, u32* __restrict__ dst_v, u32 n,
                                u32 shift, seg_hist_t (*whist)[SEG_BINS], u32* s_base /*[RADIX]*/, u32* s_wsum) {
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    s_base[threadIdx.x] = 0u;
    __syncthreads();
    for (u32 i0 = threadIdx.x; i0 < n; i0 += 4u * SEG_THREADS) {   // (four keys per thread and trip, requested together)
        u32 kh[4];
#pragma unroll
        for (u32 j = 0; j < 4; j++) kh[j] = (i0 + j * SEG_THREADS < n) ? src_k[i0 + j * SEG_THREADS] : 0u;
#pragma unroll
        for (u32 j = 0; j < 4; j++)
            if (i0 + j * SEG_THREADS < n) atomicAdd(&s_base[(kh[j] >> shift) & (RADIX - 1u)], 1u);
    }
    __syncthreads();
    {
        const u32 cnt_d = s_base[threadIdx.x];
        u32 inc = cnt_d;
#pragma unroll
        for (u32 sft = 1; sft < 64; sft <<= 1) {
            const u32 t = __shfl_up(inc, sft, 64);
            if (lane >= sft) inc += t;
        }
        if (lane == 63u) s_wsum[wave] = inc;
        __syncthreads();
        u32 woff = 0;
#pragma unroll
        for (u32 w = 0; w < SEG_THREADS / 64; w++) if (w < wave) woff += s_wsum[w];
        s_base[threadIdx.x] = woff + inc - cnt_d;  // start of digit d in the sorted segment
    }
    __syncthreads();
    seg_hist_t* const cnt16 = &whist[0][0];                                           // [256 digits][16 groups (chunk j, wave)], 8 KB
    uint4* const my_counts = reinterpret_cast<uint4*>(cnt16 + threadIdx.x * 16u);    // digit d = threadIdx.x: its sixteen counts, 32 bytes
    u32 kq[4], vq[4];
#pragma unroll
    for (u32 j = 0; j < 4u; j++) {
        const u32 i = j * SEG_THREADS + threadIdx.x;
        kq[j] = (i < n) ? src_k[i] : 0xFFFFFFFFu;
        vq[j] = (i < n) ? src_v[i] : 0u;
    }
    for (u32 c0 = 0; c0 < n; c0 += 4u * SEG_THREADS) {
        u32 kn[4], vn[4];
#pragma unroll
        for (u32 j = 0; j < 4u; j++) {
            const u32 i_next = c0 + (4u + j) * SEG_THREADS + threadIdx.x;
            kn[j] = (i_next < n) ? src_k[i_next] : 0xFFFFFFFFu;
            vn[j] = (i_next < n) ? src_v[i_next] : 0u;
        }
        my_counts[0] = make_uint4(0u, 0u, 0u, 0u);
        my_counts[1] = make_uint4(0u, 0u, 0u, 0u);
        u32 below[4], group[4];   // lanes of this wave with the same digit in chunk j: those in front of this one; all of them (0: this lane is not their first)
#pragma unroll
        for (u32 j = 0; j < 4u; j++) {
            const bool valid = c0 + j * SEG_THREADS + threadIdx.x < n;
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
        lds_barrier();   // the counts are cleared
#pragma unroll
        for (u32 j = 0; j < 4u; j++)
            if (group[j] != 0u) cnt16[((kq[j] >> shift) & (RADIX - 1u)) * 16u + j * 4u + wave] = (seg_hist_t)group[j];
        lds_barrier();   // the counts are there
        u32 trip_total = 0u;
        {   // digit d's sixteen counts -> where each group's entries of the digit start inside the trip's run of that digit (<= 1 024: 16 bits hold it)
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
        lds_barrier();   // the positions are there
#pragma unroll
        for (u32 j = 0; j < 4u; j++) {
            if (c0 + j * SEG_THREADS + threadIdx.x < n) {
                const u32 digit = (kq[j] >> shift) & (RADIX - 1u);
                const u32 pos = s_base[digit] + cnt16[digit * 16u + j * 4u + wave] + below[j];
                dst_k[pos] = kq[j];
                dst_v[pos] = vq[j];
            }
        }
        lds_barrier();   // everybody has read the bases and the positions
        s_base[threadIdx.x] += trip_total;   // (read again behind the next trip's three barriers)
#pragma unroll
        for (u32 j = 0; j < 4u; j++) { kq[j] = kn[j]; vq[j] = vn[j]; }
    }
    __syncthreads();   // (the pass's stores are complete: the caller's next pass reads them)
}

// One workgroup per segment (= tile): stable sort of [start, end) by the low 16 key bits.  keys/vals `cur` hold the data (and receive
// the result); `alt` is the other ping-pong pair, used as scratch by oversized segments only.
// Long tile lists (longlist.h): the workgroup of a tile with more than lw.threshold entries reserves the tile's four block records and
// 4 * ceil(n / 64) item slots and marks the tile (the marks were cleared by the frame's first sort_hist).  (When a tile finds no room, the header's counts say so and ALL
// long tiles of the frame stay with the main waves: longlist.h, ll_frame_on.)
__device__ void long_list_build(const LongWork& lw, u32 t, u32 n) {
    __shared__ u32 s_first, s_lb;
    const bool want = lw.threshold != 0u && n > lw.threshold && !(lw.nf_stamp && lw.nf_stamp[t] == *lw.nf_frame);
    if (!want) return;   // (uniform per workgroup; the marks were cleared by the frame's first sort_hist)
    const u32 chunks = (n + 63u) >> 6, need = 4u * chunks;
    if (threadIdx.x == 0u) {
        s_first = atomicAdd(&lw.hdr[LL_ITEMS], need);
        s_lb = atomicAdd(&lw.hdr[LL_BLOCKS], 4u);
    }
    __syncthreads();
    const u32 first = s_first, lb = s_lb;
    const bool items_fit = first + need <= lw.max_items && first + need >= first, blocks_fit = lb + 4u <= lw.max_blocks;
    const bool ok = items_fit && blocks_fit;
    // item slots below the capacity belong to this tile either way: they name their block, or say that there is none
    for (u32 i = threadIdx.x; i < need && first + i < lw.max_items && first + i >= first; i += SEG_THREADS) lw.item_block[first + i] = ok ? lb + i / chunks : 0xFFFFFFFFu;
    if (threadIdx.x < 4u && lb + threadIdx.x < lw.max_blocks) {
        const u32 b = lb + threadIdx.x;
        lw.blocks[b] = LongBlock{t, threadIdx.x, first + threadIdx.x * chunks, ok ? chunks : 0u};
        lw.sync[b] = LongSync{0u, LL_NO_ROWS, 0u, 0u, 0u, 0u, 0u, 0u};
    }
    if (threadIdx.x == 0u && ok) lw.flags[t] = 0xFu;
}

__global__ __launch_bounds__(SEG_THREADS, 6) void segment_sort_kernel(u32* __restrict__ cur_k, u32* __restrict__ cur_v, u32* __restrict__ alt_k,
                                                                    u32* __restrict__ alt_v, const u32* __restrict__ ranges, u32 total_tiles, LongWork lw) {
    WD_STREAM_PRIO();
    __shared__ u32 a_k[SEG_CAP], a_v[SEG_CAP];  // one pair: every pass reads its input into registers before anyone scatters
    __shared__ seg_hist_t whist[SEG_THREADS / 64][SEG_BINS];
    __shared__ u32 s_base[RADIX];
    __shared__ u32 s_wsum[SEG_THREADS / 64];
    __shared__ u32 s_min[SEG_THREADS / 64], s_max[SEG_THREADS / 64];
    const u32 t = blockIdx.x;
    const u32 start = ranges[t];
    u32 end = ranges[t + 1u];          // (requested together with `start`: one round trip)
    if (start == 0xFFFFFFFFu) return;  // empty tile (uniform per workgroup)
    // end of the segment = start of the next non-empty tile (ranges[T] = E ends the walk): the successor itself unless it is empty, in
    // which case every thread walks on (uniform addresses: the same few loads for the whole workgroup, no barrier)
    for (u32 nx = t + 1u; end == 0xFFFFFFFFu && nx < total_tiles;) { nx++; end = ranges[nx]; }
    if (lw.flags) long_list_build(lw, t, (end != 0xFFFFFFFFu && end > start) ? end - start : 0u);
    if (end <= start + 1u || end == 0xFFFFFFFFu) return;
    const u32 n = end - start;
    if (n <= SEG_CAP) {
        const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        const u32 per_wave = ((n + 3u) / 4u + 63u) & ~63u;  // multiple of 64; 4*per_wave >= n
        const u32 rounds = per_wave / 64u;
        u32 kk[SEG_ROUNDS], vv[SEG_ROUNDS];
#pragma unroll
        for (u32 j = 0; j < SEG_ROUNDS; j++) {  // global -> registers, coalesced per wave-round
            const u32 i = wave * per_wave + j * 64u + lane;
            const bool valid = j < rounds && i < n;
            kk[j] = valid ? cur_k[start + i] : 0xFFFFFFFFu;
            vv[j] = valid ? cur_v[start + i] : 0u;
        }
        // depth span of the segment: when max - min < 2^10 (a tile whose splats lie within a factor ~256 in depth -- nearly always) ONE
        // pass on the 10-bit digit (depth16 - min) sorts it; otherwise the two 8-bit passes on the raw depth bytes
        u32 lo16 = 0xFFFFu, hi16 = 0u;
#pragma unroll
        for (u32 j = 0; j < SEG_ROUNDS; j++) {
            const u32 i = wave * per_wave + j * 64u + lane;
            if (j < rounds && i < n) { lo16 = min(lo16, kk[j] & 0xFFFFu); hi16 = max(hi16, kk[j] & 0xFFFFu); }
        }
#pragma unroll
        for (u32 d = 32; d >= 1; d >>= 1) {
            lo16 = min(lo16, (u32)__shfl_xor((int)lo16, (int)d, 64));
            hi16 = max(hi16, (u32)__shfl_xor((int)hi16, (int)d, 64));
        }
        if (lane == 0u) { s_min[wave] = lo16; s_max[wave] = hi16; }
        __syncthreads();
        lo16 = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
        hi16 = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        if (hi16 - lo16 < (1u << 8)) {   // (every branch is uniform per workgroup; a narrower digit = fewer bins to clear and scan, fewer ballots per round)
            seg_pass_regs<8>(kk, vv, a_k, a_v, n, per_wave, lo16, 0u, whist, s_wsum);
        } else if (hi16 - lo16 < (1u << 9)) {
            seg_pass_regs<9>(kk, vv, a_k, a_v, n, per_wave, lo16, 0u, whist, s_wsum);
        } else if (hi16 - lo16 < SEG_BINS) {
            seg_pass_regs<SEG_WIDE_BITS>(kk, vv, a_k, a_v, n, per_wave, lo16, 0u, whist, s_wsum);
        } else {
            seg_pass_regs<8>(kk, vv, a_k, a_v, n, per_wave, 0u, 0u, whist, s_wsum);   // low depth byte
#pragma unroll
            for (u32 j = 0; j < SEG_ROUNDS; j++) {  // LDS -> registers in index order again
                const u32 i = wave * per_wave + j * 64u + lane;
                const bool valid = j < rounds && i < n;
                kk[j] = valid ? a_k[i] : 0xFFFFFFFFu;
                vv[j] = valid ? a_v[i] : 0u;
            }
            __syncthreads();  // everyone holds its pairs: a_k / a_v may be overwritten
            seg_pass_regs<8>(kk, vv, a_k, a_v, n, per_wave, 0u, 8u, whist, s_wsum);   // high depth byte
        }
        for (u32 i = threadIdx.x; i < n; i += SEG_THREADS) { cur_k[start + i] = a_k[i]; cur_v[start + i] = a_v[i]; }
    } else {
        seg_pass_global(cur_k + start, cur_v + start, alt_k + start, alt_v + start, n, 0u, whist, s_base, s_wsum);
        __syncthreads();
        seg_pass_global(alt_k + start, alt_v + start, cur_k + start, cur_v + start, n, 8u, whist, s_base, s_wsum);
    }
}

}  // namespace

struct wdgs_sorter {
    wdgs_device* dev;
    u32 capacity;       // elements (multiple of SORT_TILE_MAX)
    u32 items;          // keys per thread of a partition: SORT_ITEMS_MAX, or 4 for a small sorter
    u32 num_parts;
    const u32* count_ptr;
    u32* keys[2];
    u32* vals[2];
    u32* counts;        // [RADIX][num_parts], scanned in place per pass
    u32* totals;        // [RADIX]
    int final_out_index;
};

extern "C" {

int wdgs_sorter_create(wdgs_device* dev, uint32_t max_capacity, const void* stats_dev, wdgs_sorter** out) {
    WDGS_REQUIRE(dev && out && stats_dev, WDGS_E_INVALID, "wdgs_sorter_create: null argument");
    WDGS_REQUIRE(max_capacity <= 0xFFFFF000u, WDGS_E_CAPACITY, "sorter capacity %u too large", max_capacity);
    wdgs_sorter* s = new wdgs_sorter();
    s->dev = dev;
    s->capacity = (u32)align_up(max_capacity > 0 ? max_capacity : 1, SORT_TILE_MAX);
    s->items = s->capacity <= (8u << 20) ? 4u : SORT_ITEMS_MAX;  // (c2's passes: 3.75 M entries of capacity, 0.36 M used; c3's: 37.5 M)
    s->num_parts = s->capacity / (SORT_THREADS * s->items);
    s->count_ptr = (const u32*)stats_dev;
    s->final_out_index = 0;
    for (int i = 0; i < 2; i++) { s->keys[i] = nullptr; s->vals[i] = nullptr; }
    s->counts = nullptr;
    s->totals = nullptr;
    int r = WDGS_OK;
    for (int i = 0; i < 2 && r == WDGS_OK; i++) {
        r = wdgs_alloc((void**)&s->keys[i], sizeof(u32) * (size_t)s->capacity, true, dev->stream);
        if (r == WDGS_OK) r = wdgs_alloc((void**)&s->vals[i], sizeof(u32) * (size_t)s->capacity, true, dev->stream);
    }
    if (r == WDGS_OK) r = wdgs_alloc((void**)&s->counts, sizeof(u32) * (size_t)RADIX * s->num_parts, true, dev->stream);
    if (r == WDGS_OK) r = wdgs_alloc((void**)&s->totals, sizeof(u32) * RADIX, true, dev->stream);
    if (r != WDGS_OK) { wdgs_sorter_destroy(s); return r; }
    *out = s;
    return WDGS_OK;
}

int wdgs_sorter_destroy(wdgs_sorter* s) {
    if (!s) return WDGS_OK;
    for (int i = 0; i < 2; i++) {
        if (s->keys[i]) wdgs_free(s->keys[i]);
        if (s->vals[i]) wdgs_free(s->vals[i]);
    }
    if (s->counts) wdgs_free(s->counts);
    if (s->totals) wdgs_free(s->totals);
    delete s;
    return WDGS_OK;
}

void* wdgs_sorter_keys(wdgs_sorter* s, int i) { return s ? s->keys[i & 1] : nullptr; }
void* wdgs_sorter_values(wdgs_sorter* s, int i) { return s ? s->vals[i & 1] : nullptr; }
int wdgs_sorter_final_out_index(wdgs_sorter* s) { return s ? s->final_out_index : 0; }
// (internal) the forward pass left unsorted entries in ping-pong 0 (encode(skipSort)): that is what the getters hand out
void sorter_set_final_out_index(wdgs_sorter* s, int i) { if (s) s->final_out_index = i & 1; }
uint32_t wdgs_sorter_capacity(wdgs_sorter* s) { return s ? s->capacity : 0; }

}  // extern "C"

// Stable sort of keys laid out as (segment id << 16 | 16-bit minor key): LSD passes over the segment bits, the range table of
// the segments, then one workgroup per segment for the minor key.  `ranges` = u32[num_segments + 1], written here.
int sorter_sort_segmented(wdgs_sorter* s, u32 segment_bits, u32 num_segments, u32* ranges, const LongWork* lw) {
    wdgs_device* dev = s->dev;
    const u32 bits = std::min(segment_bits, 16u);
    const u32 passes = (bits + 7u) / 8u;
    // Balanced digit widths (13 tile bits -> 6 + 7, not 8 + 5): a partition's 4096 keys leave in 2^width runs, and the scatter's
    // write efficiency follows the run length (measured at c3: 40 us for 256 runs of 16 keys, 26 us for 32 runs of 128).
    int src = 0;
    u32 shift = 16u, left = bits;
    for (u32 p = 0; p < passes; p++) {
        const u32 width = left / (passes - p);
        left -= width;
        const u32 dmask = (1u << width) - 1u;
        u32* const marks = (p == 0u && lw) ? lw->flags : nullptr;   // (the long-list marks are cleared by the frame's first histogram kernel)
        if (s->items == 4u) {
            WDGS_LAUNCH(dev, "sort_hist", sort_hist_kernel<4u>, dim3(s->num_parts), dim3(SORT_THREADS), 0, s->keys[src], s->count_ptr, (DigitOf{shift, dmask, 0u}), s->num_parts,
                        s->counts, (u32*)nullptr, num_segments, marks);
        } else {
            WDGS_LAUNCH(dev, "sort_hist", sort_hist_kernel<SORT_ITEMS_MAX>, dim3(s->num_parts), dim3(SORT_THREADS), 0, s->keys[src], s->count_ptr, (DigitOf{shift, dmask, 0u}), s->num_parts,
                        s->counts, (u32*)nullptr, num_segments, marks);
        }
        if (s->items == 4u) {
            WDGS_LAUNCH(dev, "sort_scan_rows", sort_scan_rows_kernel<4u>, dim3(dmask + 1u), dim3(256), 0, s->counts, s->count_ptr, s->num_parts, s->totals);
        } else {
            WDGS_LAUNCH(dev, "sort_scan_rows", sort_scan_rows_kernel<SORT_ITEMS_MAX>, dim3(dmask + 1u), dim3(256), 0, s->counts, s->count_ptr, s->num_parts, s->totals);
        }
        // (two passes: the first initialises the range table, the second fills it; any other pass count keeps the search kernel)
        const u32 ranges_mode = (passes == 2u) ? p + 1u : 0u;
        if (s->items == 4u) {
            WDGS_LAUNCH(dev, "sort_scatter", sort_scatter_kernel<4u>, dim3(s->num_parts), dim3(SORT_THREADS), 0, s->keys[src], s->vals[src], s->keys[src ^ 1],
                        s->vals[src ^ 1], s->count_ptr, (DigitOf{shift, dmask, 0u}), s->num_parts, s->counts, s->totals, ranges, ranges_mode, num_segments);
        } else {
            WDGS_LAUNCH(dev, "sort_scatter", sort_scatter_kernel<SORT_ITEMS_MAX>, dim3(s->num_parts), dim3(SORT_THREADS), 0, s->keys[src], s->vals[src], s->keys[src ^ 1],
                        s->vals[src ^ 1], s->count_ptr, (DigitOf{shift, dmask, 0u}), s->num_parts, s->counts, s->totals, ranges, ranges_mode, num_segments);
        }
        src ^= 1;
        shift += width;
    }
    if (passes != 2u)
        WDGS_LAUNCH(dev, "tile_ranges", tile_ranges_kernel, dim3(ceil_div(num_segments + 1, 4)), dim3(256), 0, s->keys[src], s->count_ptr, num_segments, ranges);
    if (num_segments > 0)
        WDGS_LAUNCH(dev, "sort_segments", segment_sort_kernel, dim3(num_segments), dim3(SEG_THREADS), 0, s->keys[src], s->vals[src], s->keys[src ^ 1],
                    s->vals[src ^ 1], ranges, num_segments, lw ? *lw : LongWork{});
    WDGS_CHECK_HIP(hipGetLastError());
    s->final_out_index = src;
    return WDGS_OK;
}

// The forward pass's sort when emit has already written its entries in tile-COLUMN order (project.hip: emit_scatter = the first stable
// pass, on tx): ONE stable pass on the tile ROW (ty = tile / num_tiles_x <= 255), which also builds the per-tile range table -- its
// histogram kernel initialises the table, its scatter lowers ranges[tile] to the first position it writes for the tile -- then the per-tile
// depth sort.  Input in ping-pong 0, result in ping-pong 1.  4E + 16E + 16E bytes.
int sorter_sort_rows(wdgs_sorter* s, u32 num_tiles_x, u32 num_tiles_y, u32* ranges, const LongWork* lw) {
    wdgs_device* dev = s->dev;
    const u32 tiles = num_tiles_x * num_tiles_y;
    const DigitOf rows{0u, num_tiles_y - 1u, 0xFFFFFFFFu / num_tiles_x + 1u};  // (num_tiles_x >= 2)
    if (s->items == 4u) {
        WDGS_LAUNCH(dev, "sort_hist", sort_hist_kernel<4u>, dim3(s->num_parts), dim3(SORT_THREADS), 0, s->keys[0], s->count_ptr, rows, s->num_parts, s->counts, ranges, tiles, lw ? lw->flags : (u32*)nullptr);
    } else {
        WDGS_LAUNCH(dev, "sort_hist", sort_hist_kernel<SORT_ITEMS_MAX>, dim3(s->num_parts), dim3(SORT_THREADS), 0, s->keys[0], s->count_ptr, rows, s->num_parts, s->counts, ranges, tiles, lw ? lw->flags : (u32*)nullptr);
    }
    if (s->items == 4u) {
        WDGS_LAUNCH(dev, "sort_scan_rows", sort_scan_rows_kernel<4u>, dim3(num_tiles_y), dim3(256), 0, s->counts, s->count_ptr, s->num_parts, s->totals);
    } else {
        WDGS_LAUNCH(dev, "sort_scan_rows", sort_scan_rows_kernel<SORT_ITEMS_MAX>, dim3(num_tiles_y), dim3(256), 0, s->counts, s->count_ptr, s->num_parts, s->totals);
    }
    if (s->items == 4u) {
        WDGS_LAUNCH(dev, "sort_scatter", sort_scatter_kernel<4u>, dim3(s->num_parts), dim3(SORT_THREADS), 0, s->keys[0], s->vals[0], s->keys[1], s->vals[1], s->count_ptr, rows,
                    s->num_parts, s->counts, s->totals, ranges, 2u, tiles);
    } else {
        WDGS_LAUNCH(dev, "sort_scatter", sort_scatter_kernel<SORT_ITEMS_MAX>, dim3(s->num_parts), dim3(SORT_THREADS), 0, s->keys[0], s->vals[0], s->keys[1], s->vals[1], s->count_ptr, rows,
                    s->num_parts, s->counts, s->totals, ranges, 2u, tiles);
    }
    WDGS_LAUNCH(dev, "sort_segments", segment_sort_kernel, dim3(tiles), dim3(SEG_THREADS), 0, s->keys[1], s->vals[1], s->keys[0], s->vals[0], ranges, tiles, lw ? *lw : LongWork{});
    WDGS_CHECK_HIP(hipGetLastError());
    s->final_out_index = 1;
    return WDGS_OK;
}

extern "C" {

int wdgs_sorter_sort(wdgs_sorter* s, uint32_t key_bits) {
    WDGS_REQUIRE(s, WDGS_E_INVALID, "wdgs_sorter_sort: null sorter");
    if (key_bits == 0 || key_bits > 32) key_bits = 32;
    const u32 passes = (key_bits + 7u) / 8u;
    wdgs_device* dev = s->dev;
    int src = 0;
    for (u32 p = 0; p < passes; p++) {
        const u32 shift = p * 8u;
        if (s->items == 4u) {
            WDGS_LAUNCH(dev, "sort_hist", sort_hist_kernel<4u>, dim3(s->num_parts), dim3(SORT_THREADS), 0, s->keys[src], s->count_ptr, (DigitOf{shift, RADIX - 1u, 0u}), s->num_parts,
                        s->counts, (u32*)nullptr, 0u, (u32*)nullptr);
        } else {
            WDGS_LAUNCH(dev, "sort_hist", sort_hist_kernel<SORT_ITEMS_MAX>, dim3(s->num_parts), dim3(SORT_THREADS), 0, s->keys[src], s->count_ptr, (DigitOf{shift, RADIX - 1u, 0u}), s->num_parts,
                        s->counts, (u32*)nullptr, 0u, (u32*)nullptr);
        }
        if (s->items == 4u) {
            WDGS_LAUNCH(dev, "sort_scan_rows", sort_scan_rows_kernel<4u>, dim3(RADIX), dim3(256), 0, s->counts, s->count_ptr, s->num_parts, s->totals);
        } else {
            WDGS_LAUNCH(dev, "sort_scan_rows", sort_scan_rows_kernel<SORT_ITEMS_MAX>, dim3(RADIX), dim3(256), 0, s->counts, s->count_ptr, s->num_parts, s->totals);
        }
        if (s->items == 4u) {
            WDGS_LAUNCH(dev, "sort_scatter", sort_scatter_kernel<4u>, dim3(s->num_parts), dim3(SORT_THREADS), 0, s->keys[src], s->vals[src], s->keys[src ^ 1],
                        s->vals[src ^ 1], s->count_ptr, (DigitOf{shift, RADIX - 1u, 0u}), s->num_parts, s->counts, s->totals, (u32*)nullptr, 0u, 0u);
        } else {
            WDGS_LAUNCH(dev, "sort_scatter", sort_scatter_kernel<SORT_ITEMS_MAX>, dim3(s->num_parts), dim3(SORT_THREADS), 0, s->keys[src], s->vals[src], s->keys[src ^ 1],
                        s->vals[src ^ 1], s->count_ptr, (DigitOf{shift, RADIX - 1u, 0u}), s->num_parts, s->counts, s->totals, (u32*)nullptr, 0u, 0u);
        }
        src ^= 1;
    }
    WDGS_CHECK_HIP(hipGetLastError());
    s->final_out_index = src;
    return WDGS_OK;
}

}  // extern "C"

int launch_tile_ranges(wdgs_device* dev, const void* sorted_keys, const void* count_ptr, u32 total_tiles, void* ranges) {
    WDGS_LAUNCH(dev, "tile_ranges", tile_ranges_kernel, dim3(ceil_div(total_tiles + 1, 4)), dim3(256), 0, (const u32*)sorted_keys, (const u32*)count_ptr,
                total_tiles, (u32*)ranges);
    WDGS_CHECK_HIP(hipGetLastError());
    return WDGS_OK;
}
