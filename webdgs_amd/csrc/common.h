// Internal definitions shared by the HIP translation units of libwebdgs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/webdgs.h"

typedef uint32_t u32;
typedef int32_t i32;

void wdgs_set_error(const char* fmt, ...);

#define WDGS_CHECK_HIP(expr)                                                                      \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            wdgs_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return WDGS_E_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define WDGS_REQUIRE(cond, code, ...)  \
    do {                               \
        if (!(cond)) {                 \
            wdgs_set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

#define WDGS_TRY(expr)                 \
    do {                               \
        int _r = (expr);               \
        if (_r != WDGS_OK) return _r;  \
    } while (0)

struct wdgs_device {
    int ordinal = 0;
    hipStream_t stream = nullptr;  // the CURRENT lane's stream: everything is launched, recorded and copied on it
    bool own_stream = false;
    // Lanes: lane 0 is the device's own stream (the host framework's), the others internal ones created on first use.  A batched step
    // deals consecutive views to the lanes in turn so that the bandwidth-bound kernels of one view execute beside the VALU-bound
    // rasterization kernels of another (api.hip: wdgs_device_select_lane / wdgs_device_lane_order).
    hipStream_t lanes[WDGS_MAX_LANES] = {};
    hipEvent_t lane_events[WDGS_MAX_LANES] = {};
    hipEvent_t lane_marks[WDGS_MAX_BATCH_VIEWS] = {};  // wdgs_device_lane_mark / wdgs_device_lane_wait_mark
    int lane = 0;
    // Tickets (wdgs_queue_mark / wdgs_queue_wait): a ring of events, so a host can wait for step k-1 while step k runs.
    hipEvent_t ticket_events[WDGS_TICKET_RING] = {};
    uint64_t ticket_next = 1;
    // Sticky, device-written: a guarded optimizer step found its guard word set and skipped itself (pinned host memory).
    u32* host_guard = nullptr;
    bool profiling = false;
    bool capturing = false;
    struct Pending { const char* name; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    struct Total { u32 launches = 0; float ms = 0.f; };
    std::map<std::string, Total> totals;
    std::vector<wdgs_tiled_forward*> forwards;  // live forward passes, for deferred overflow checks
    // per-tile range tables of live forward passes -> the pass (api.hip: a backward pass that is handed such a table finds the pass's long-list work)
    std::map<const void*, wdgs_tiled_forward*> range_tables;
    // forward passes whose PROJECTION the open recording consumes (wdgs_tiled_forward_encode_projected while capturing): handed to the
    // command buffer by wdgs_encoder_finish, checked and consumed by every wdgs_queue_submit of it
    std::vector<wdgs_tiled_forward*> capture_consumes;
    // command buffers whose destruction is owed (wdgs_command_buffer_destroy): hipGraphExecDestroy costs ~170 us with the ROCm 7.2 runtime, and a
    // densify event drops one per training view; wdgs_queue_submit pays one off per call, behind its launch, where the device has work to do meanwhile
    std::vector<void*> dead_command_buffers;
    int num_cus = 256;
};

// Host wait for everything submitted to the device, on either lane.
inline hipError_t wdgs_sync_lanes(wdgs_device* d) {
    for (int l = 1; l < WDGS_MAX_LANES; l++) {
        if (!d->lanes[l]) continue;
        const hipError_t e = hipStreamSynchronize(d->lanes[l]);
        if (e != hipSuccess) return e;
    }
    return hipStreamSynchronize(d->lanes[0] ? d->lanes[0] : d->stream);
}

// Brackets one kernel launch with events when profiling is on (hipEvents on the launch stream).
struct KernelScope {
    wdgs_device* d;
    const char* name;
    hipEvent_t a = nullptr, b = nullptr;
    KernelScope(wdgs_device* dev, const char* n) : d(dev), name(n) {
        if (d->profiling && !d->capturing) {
            a = take(); b = take();
            (void)hipEventRecord(a, d->stream);
        }
    }
    ~KernelScope() {
        if (a) {
            (void)hipEventRecord(b, d->stream);
            d->pending.push_back({name, a, b});
        }
    }
    hipEvent_t take() {
        if (!d->event_pool.empty()) { hipEvent_t e = d->event_pool.back(); d->event_pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
};

// Wave priority of the bandwidth- / latency-bound kernels (0..3; the rasterization kernels stay at the default 0): where a lane's streaming
// kernel shares a SIMD with another lane's rasterization waves, its few instructions are issued ahead of theirs, so it gets its memory
// requests out and leaves sooner.  Issue order only -- results cannot depend on it.  A build-time choice (make STREAM_PRIO=n) so that two
// builds can be compared on one box (WDGS_LIB_PATH).
#ifndef WDGS_STREAM_PRIO
#define WDGS_STREAM_PRIO 0
#endif
#define WD_STREAM_PRIO()                                                     \
    do {                                                                     \
        if (WDGS_STREAM_PRIO > 0) __builtin_amdgcn_s_setprio(WDGS_STREAM_PRIO); \
    } while (0)

#define WDGS_LAUNCH(dev, kname, kernel, grid, block, shmem, ...)                           \
    do {                                                                                   \
        KernelScope _ks((dev), (kname));                                                   \
        hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (dev)->stream, __VA_ARGS__);  \
    } while (0)

// The optimizer's compact training copy (adam.h): seven planes of float4, planes[k * pitch + idx].
constexpr u32 CS_PLANES = 7;
struct CsView {
    float4* planes;
    u32 pitch;  // Gaussians per plane
    __device__ __forceinline__ float4& quad(u32 k, u32 idx) const { return planes[(size_t)k * pitch + idx]; }
};

// Workgroup number for launch slot b of a grid of g.  Slots b, b + 8, b + 16, ... are observed to run on one XCD (round-robin dispatch
// over the chip's 8 XCDs); this numbering gives XCD k the CONTIGUOUS range of workgroups [start_k, start_k + count_k), so that
// workgroups writing neighbouring addresses meet in one L2.  A bijection of [0, g) for every g; a speed choice only.
__device__ __forceinline__ u32 xcd_contiguous(u32 b, u32 g) {
    const u32 k = b & 7u, j = b >> 3, q = g >> 3, r = g & 7u;
    return k * q + min(k, r) + j;
}

static inline u32 ceil_div(u32 a, u32 b) { return (a + b - 1u) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

int wdgs_alloc(void** p, size_t bytes, bool zero, hipStream_t stream);
void wdgs_free(void* p);   // the counterpart of wdgs_alloc (api.hip: freed blocks are kept by size class)
// true between wdgs_device_create and wdgs_device_destroy (api.hip): destroy functions check it before touching op->dev
bool wdgs_device_alive(const wdgs_device* d);

// ---- primitives implemented in scan.hip / sort.hip, used by the ops
struct ScanScratch { u32* block_sums = nullptr; u32 capacity_blocks = 0; };
int scan_scratch_create(ScanScratch* s, u32 max_elements);
void scan_scratch_destroy(ScanScratch* s);
// Exclusive u32 scan of `count` (host-known) elements.  If total_out != nullptr, writes the grand total there.
int scan_exclusive_u32(wdgs_device* dev, ScanScratch* s, const u32* in, u32* out, u32 count, u32* total_out);
// Same, with the forward pass's stats epilogue folded into the single-block middle kernel (count must be > 0 for it to run).
// frame (nullable): the forward pass's frame number, advanced by the scan kernel -- project.hip stamps the tiles of non-finite Splats with the number
// the frame is ABOUT to get, so a stamp never has to be cleared (raster.hip compares)
// long_hdr (nullable): the header of the pass's long-list work (longlist.h), zeroed for the frame by the same kernel
struct ScanStatsEpilogue { u32* stats; u32* visible_shards; u32* host_mirror; u32 capacity; u32* frame = nullptr; u32* long_hdr = nullptr; };
int scan_exclusive_u32_stats(wdgs_device* dev, ScanScratch* s, const u32* in, u32* out, u32 count, u32* total_out, const ScanStatsEpilogue& ep);

int scan_block_sums_inplace(wdgs_device* dev, u32* block_sums, u32 num_blocks, const ScanStatsEpilogue& ep);
// scan_block_sums_inplace + the row scans of the per-workgroup tile-column counts (column_counts[columns][num_blocks] -> offsets in place, totals)
int forward_scan(wdgs_device* dev, u32* block_sums, u32 num_blocks, u32* column_counts, u32* column_totals, u32 columns, const ScanStatsEpilogue& ep);

struct RenderSettings { float gaussian_scaling, sh_deg, viewport_x, viewport_y, point_size_px, gaussian_mode, max_splat_radius_px; };
struct TileInfo { u32 num_tiles_x, num_tiles_y, total_tiles, max_tile_entries; };
