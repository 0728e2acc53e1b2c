// C ABI of libwebdgs_hip.so (include/webdgs.h): handle types, ownership, launch sequencing.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <map>
#include <set>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "alloc_cache.h"
#include "longlist.h"

// ---- kernel launchers (project.hip, sort.hip, raster.hip, loss.hip, backward.hip, optimizer.hip)
int launch_project_count(wdgs_device*, u32, const void*, const void*, const void*, const RenderSettings&, const TileInfo&, void*, void*, void*, void*, void*, void*,
                         const void*, void*, const void*);
int launch_emit_scatter(wdgs_device*, u32, const void*, const void*, const void*, void*, const void*, const RenderSettings&, const TileInfo&, const void*, const void*, void*, void*, u32);
int sorter_sort_rows(wdgs_sorter* s, u32 num_tiles_x, u32 num_tiles_y, u32* ranges, const LongWork* lw);
extern "C" void sorter_set_final_out_index(wdgs_sorter* s, int i);
int launch_update_stats(wdgs_device*, u32, const void*, const void*, u32, void*, void*, void*);
int launch_emit(wdgs_device*, u32, const void*, const void*, const void*, void*, const void*, const RenderSettings&, const TileInfo&, void*, void*, u32);
int launch_tile_ranges(wdgs_device*, const void*, const void*, u32, void*);
int launch_rasterize(wdgs_device*, const RenderSettings&, const TileInfo&, const void*, u32, const void*, const void*, const void*, const void*, u32, void*,
                     void*, void*, const void*, const void*, const LongWork*);
int launch_loss_grad(wdgs_device*, u32, u32, const void*, const void*, const wdgs_training_config&, void*, void*, u32, const void*);
int launch_backward_rasterize(wdgs_device*, const RenderSettings&, u32, u32, const void*, const void*, const void*, const void*, const void*, const void*,
                              void*, void*, const LongWork*);
int launch_acc_clear_if_dirty(wdgs_device*, void*, u32, void*);
int launch_geometry_backward_views(wdgs_device*, u32, u32, const void* const*, const RenderSettings&, const void*, void* const*, void* const*, const void* const*,
                                   const void* const*, void* const*, void*, void*, void*, u32);
int launch_project_count_views(wdgs_device*, u32, u32, const void*, const void*, const void* const*, const RenderSettings&, const TileInfo&, void* const*, void* const*,
                               void* const*, void* const*, void* const*, void* const*, const void*, void* const*, const void* const*);
int launch_geometry_backward(wdgs_device*, u32, const void*, const RenderSettings&, const void*, void*, void*);
int launch_geometry_backward_adam(wdgs_device*, u32, const void*, const RenderSettings&, void*, void*, void*, void*, const wdgs_adam_hyperparameters&, const void*,
                                  const wdgs_optimizer_state&, const CsView&, void*, const void*, void*);
int launch_geometry_backward_accumulate(wdgs_device*, u32, const void*, const RenderSettings&, const void*, void*, void*, void*, void*, void*, const void*, void*,
                                        const void*, u32);
int launch_adam_repack(wdgs_device*, u32, const wdgs_adam_hyperparameters&, const void*, const void*, const wdgs_optimizer_state&, const CsView&, void*, void*,
                       const void*, void*);
int launch_adam_repack_f32(wdgs_device*, u32, u32, const wdgs_adam_hyperparameters&, const void*, const void*, const wdgs_optimizer_state&, const CsView&, void*,
                           void*, const void*, void*, void*, void*);
int launch_apply_rows(wdgs_device*, u32, const void*, u32, u32, const void*, void*, void*, void*, void*);
int launch_dc_words_load(wdgs_device*, u32, const void*, void*);
int launch_dc_words_flush(wdgs_device*, u32, const void*, void*);
int launch_guard_accumulate(wdgs_device*, void*, const void*, u32);
int launch_cs_load(wdgs_device*, u32, const wdgs_optimizer_state&, const CsView&);
int launch_cs_flush(wdgs_device*, u32, const CsView&, const wdgs_optimizer_state&);
int launch_accumulate_gradients(wdgs_device*, u32, const void*, const void*, void*, void*);
int launch_store_gradients(wdgs_device*, u32, const void*, const void*, void*, void*);
int launch_unpack(wdgs_device*, u32, const void*, const void*, const wdgs_optimizer_state&);
int launch_metric_map(wdgs_device*, u32, u32, const void*, const void*, float, float, void*, void*, void*, void*);
int launch_metric_count(wdgs_device*, const RenderSettings&, u32, u32, const void*, const void*, u32, const void*, u32, const void*, const void*, void*, u32);
int launch_metric_normalize(wdgs_device*, u32, u32, void*);
int launch_downsample(wdgs_device*, const void*, u32, u32, void*, u32, u32);

int sorter_sort_segmented(wdgs_sorter* s, u32 segment_bits, u32 num_segments, u32* ranges, const LongWork* lw);
extern "C" int wdgs_sorter_final_out_index(wdgs_sorter* s);
extern "C" uint32_t wdgs_sorter_capacity(wdgs_sorter* s);

static thread_local char g_last_error[512] = "";

// Devices that have been created and not yet destroyed.  Every op keeps a pointer to its device; a host that tears things down
// in the wrong order (an interpreter finalising its objects at exit, say) may call an op's destroy after wdgs_device_destroy.
// Destroy functions therefore ask here before they touch op->dev, and only release memory when the device is already gone.
static std::mutex g_live_mutex;
static std::set<const wdgs_device*> g_live_devices;
bool wdgs_device_alive(const wdgs_device* d) {
    std::lock_guard<std::mutex> lock(g_live_mutex);
    return d && g_live_devices.count(d) != 0;
}
// stream of a live device that is not in the middle of a recording, else nullptr ("nothing to wait for")
static void sync_if_alive(wdgs_device* d) {
    if (wdgs_device_alive(d) && !d->capturing) (void)wdgs_sync_lanes(d);
}

void wdgs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------- device memory
// Freed blocks are kept by size class and handed out again (alloc_cache.h: why, and the per-device epochs that say when a freed block is safe).
// WDGS_ALLOC_CACHE=0: plain hipMalloc / hipFree.  The cache holds at most a quarter of the device's memory; wdgs_device_destroy empties it.
namespace {
struct HipBackend {
    void* malloc(int device, size_t bytes) {
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) return nullptr;
        if (cur != device && hipSetDevice(device) != hipSuccess) return nullptr;
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, bytes);
        if (cur != device) (void)hipSetDevice(cur);
        if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return p;
    }
    void free(void* p) { (void)hipFree(p); }
    bool sync(int device) {   // every stream of the process on that device, a torch or RCCL side stream included
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) return false;
        if (cur != device && hipSetDevice(device) != hipSuccess) return false;
        const hipError_t e = hipDeviceSynchronize();
        if (cur != device) (void)hipSetDevice(cur);
        return e == hipSuccess;
    }
};
wdgs::AllocCache<HipBackend> g_alloc_cache;
size_t g_cache_limit = 0;
std::mutex g_cache_limit_mutex;
bool alloc_cache_enabled() {
    static const bool on = !(std::getenv("WDGS_ALLOC_CACHE") && std::getenv("WDGS_ALLOC_CACHE")[0] == '0');
    return on;
}
}  // namespace

int wdgs_alloc(void** p, size_t bytes, bool zero, hipStream_t stream) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    if (!alloc_cache_enabled()) {
        WDGS_CHECK_HIP(hipMalloc(p, bytes));
    } else {
        int device = 0;
        WDGS_CHECK_HIP(hipGetDevice(&device));
        // (an allocation while `stream` records -- relaxed capture allows hipMalloc -- cannot synchronise the device: it only takes blocks that need no wait)
        hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
        const bool recording = stream && hipStreamIsCapturing(stream, &capture) == hipSuccess && capture != hipStreamCaptureStatusNone;
        *p = g_alloc_cache.alloc(device, bytes, !recording);
        WDGS_REQUIRE(*p, WDGS_E_HIP, "device allocation of %zu bytes failed (device %d)", bytes, device);
    }
    if (zero) WDGS_CHECK_HIP(hipMemsetAsync(*p, 0, bytes, stream));
    return WDGS_OK;
}

void wdgs_free(void* p) {
    if (!p) return;
    if (alloc_cache_enabled()) {
        size_t limit;
        {
            std::lock_guard<std::mutex> lock(g_cache_limit_mutex);
            if (g_cache_limit == 0) {
                size_t free_b = 0, total_b = 0;
                g_cache_limit = (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) ? total_b / 4 : ((size_t)8 << 30);
            }
            limit = g_cache_limit;
        }
        if (g_alloc_cache.free(p, limit)) return;
    }
    (void)hipFree(p);
}
static void free_dev(void* p) { wdgs_free(p); }

struct wdgs_buffer {
    void* ptr;
    size_t size;
};

struct wdgs_prefix_scanner {
    wdgs_device* dev;
    u32 max_elements, count;
    u32 *input, *output;
    ScanScratch scratch;
};

struct wdgs_tiled_forward {
    wdgs_device* dev;
    wdgs_tiled_forward_config cfg;
    RenderSettings settings;
    TileInfo tile_info;
    u32* stats;   // {total_tile_entries, visible_gaussians, overflow (0 or requested total), pad} + 64 visible-count shards + the frame number (FRAME_WORD)
    // u32[tiles]: a tile whose entry equals the frame number holds a Splat with a NaN or an infinity among its fp16 fields (project.hip stamps,
    // scan.hip advances the number, raster.hip takes such tiles in the oracle's own forms)
    u32* nf_stamp;
    u32 nf_capacity;
    // Long tile lists (longlist.h): tables, scratch and the per-tile marks; `long_lists.hdr == nullptr`: switched off.  Built by this pass's sort, used by
    // the rasterizer that composites with this pass's range table and by a backward pass that is handed that table (found through dev->range_tables).
    LongWork long_lists;
    u32 long_flags_capacity;
    u32* host_stats;  // pinned, device-visible copy of stats[0..3] written by update_stats: the per-step overflow check reads host memory
    u32* splats;
    u32* depths;
    u32* block_counts;  // u32[ceil(N/256)]: tile entries per project_count workgroup, scanned in place into workgroup offsets
    // u32[256][ceil(N/256)] + u32[256]: tile entries per workgroup and tile COLUMN (project_count), scanned per column into the workgroup's
    // offset inside the column, and the column totals -- the digit counts of the sort's first pass, which emit_scatter performs (project.hip)
    u32* column_counts;
    u32* column_totals;
    const void* dc_source;  // nullable: the optimizer's compact SH-DC words (wdgs_tiled_forward_set_dc_source), read by project_count in place of the rows' first 6 bytes
    u32 points_capacity;  // Gaussians the per-Gaussian buffers above and the scanner hold (>= cfg.num_points: wdgs_tiled_forward_resize)
    wdgs_prefix_scanner* scanner;  // input = tile counts, output = per-Gaussian offsets
    wdgs_sorter* sorter;
    // Per-tile range table u32[tiles + 1].  The sort of tile-structured keys needs it half way (sort.hip, sort_segmented), so the
    // forward pass builds it and the rasterizer -- which owns K12-K13 in the reference (tiled-rasterizer.ts:213-230) -- takes it over
    // instead of searching the keys a second time.  Not valid after encode(skip_sort) or under compat_caps (plain 4-pass sort).
    u32* ranges;
    u32 ranges_capacity;
    bool ranges_valid;
    bool encoded;
    bool projected;          // K1 of the current frame ran through wdgs_tiled_forward_project_views
    bool projected_columns;  // ... and counted per tile column
};

constexpr u32 FRAME_WORD = 4u + 64u;   // index of the frame number in wdgs_tiled_forward::stats
constexpr size_t FORWARD_STATS_BYTES = 16 + 64 * 4 + 16;

struct wdgs_tiled_rasterizer {
    wdgs_device* dev;
    wdgs_tiled_forward* fwd;
    u32 compat_caps;
    u32 width, height;       // allocated image size
    u32 ranges_capacity;     // tiles + 1
    u32* ranges;             // own table (used when the forward pass did not build one)
    u32* ranges_used;        // the table the last encode composited with: own or the forward pass's
    u32* rgba8;
    float* alpha;
    u32* n_contrib;
    bool encoded;
};

struct wdgs_tiled_backward {
    wdgs_device* dev;
    wdgs_tiled_backward_config cfg;
    RenderSettings settings;
    int* acc;            // i32[N*12], followed by the state word below
    u32* acc_dirty;      // device word behind the accumulators: 0 = all rows are zero (a consuming K17 left them so), 1 = they hold sums
    u32* gradients;      // GaussianGradient[N]
    bool gradient_output;  // the fused K17 + Adam step also writes the packed gradient (wdgs_tiled_backward_set_gradient_output; default on)
    float* loss_image;   // rgba32f
    u32* metric_counts;  // u32[N]
    u32* metric_counts_into;  // nullable: computeMetricCounts adds into THIS array instead (wdgs_tiled_backward_set_metric_counts_target)
    u32* metric_err;     // u32[W*H]
    u32* metric_flags;   // u32[W*H]
    u32* metric_minmax;  // u32[2] + scratch
    u32 img_capacity;    // pixels allocated
    u32 points_capacity; // Gaussians acc / gradients / metric_counts hold (>= cfg.num_points: wdgs_tiled_backward_resize)
};

struct wdgs_optimizer {
    wdgs_device* dev;
    u32 num_points;
    wdgs_adam_hyperparameters params;
    wdgs_optimizer_state state;
    bool owns_state;
    u32 iteration;
    float* dc;        // compact training copy, float4[7][max(N, 1)] planes: position, log-scale and SH-DC {param, m, v} (adam.h; optimizer.hip "HBM layout note"); always owned
    static u32 cs_pitch(u32 n) { return (std::max(n, 1u) + 15u) & ~15u; }  // planes start on 256-byte boundaries
    CsView cs() const { return CsView{reinterpret_cast<float4*>(dc), cs_pitch(num_points)}; }
    bool dc_dirty;    // it is ahead of state.opt_pos / opt_scale / param_sh / state_sh
    const void* guard;  // device word: non-zero at execution time turns step / step_f32 into a no-op (wdgs_optimizer_set_guard)
    // Deferred SH writes (wdgs_optimizer_set_deferred_sh): the steps write the trained DC halves to dc_words (u32[N][2]) instead of the
    // 96-byte rows; sh_stale says the rows are behind until wdgs_optimizer_flush_sh.
    u32* dc_words;
    bool deferred_sh;
    bool sh_stale;
};

// Brings the reference-layout arrays (position, log-scale, SH) up to date with the compact training copy (no-op when nothing was trained since).
static int optimizer_flush_dc(wdgs_optimizer* op) {
    if (!op->dc_dirty) return WDGS_OK;
    WDGS_TRY(launch_cs_flush(op->dev, op->num_points, op->cs(), op->state));
    op->dc_dirty = false;
    return WDGS_OK;
}

extern "C" {

const char* wdgs_last_error(void) { return g_last_error; }
int wdgs_abi_version(void) { return 1; }

// ---------------------------------------------------------------- device
int wdgs_device_create(int ordinal, void* external_stream, wdgs_device** out) {
    WDGS_REQUIRE(out, WDGS_E_INVALID, "wdgs_device_create: out is null");
    int count = 0;
    WDGS_CHECK_HIP(hipGetDeviceCount(&count));
    WDGS_REQUIRE(ordinal >= 0 && ordinal < count, WDGS_E_INVALID, "wdgs_device_create: ordinal %d out of range (%d devices)", ordinal, count);
    WDGS_CHECK_HIP(hipSetDevice(ordinal));
    wdgs_device* d = new wdgs_device();
    d->ordinal = ordinal;
    if (external_stream) {
        d->stream = (hipStream_t)external_stream;
        d->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete d; wdgs_set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); return WDGS_E_HIP; }
        d->own_stream = true;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ordinal) == hipSuccess) d->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    d->lanes[0] = d->stream;
    if (hipHostMalloc((void**)&d->host_guard, 16, hipHostMallocDefault) != hipSuccess) {
        wdgs_set_error("hipHostMalloc(16) failed");
        if (d->own_stream) (void)hipStreamDestroy(d->stream);
        delete d;
        return WDGS_E_HIP;
    }
    std::memset(d->host_guard, 0, 16);
    { std::lock_guard<std::mutex> lock(g_live_mutex); g_live_devices.insert(d); }
    *out = d;
    return WDGS_OK;
}

// ---- lanes: in-order streams of one device, ordered against each other only where the host says so ----
int wdgs_device_select_lane(wdgs_device* d, int lane) {
    WDGS_REQUIRE(d, WDGS_E_INVALID, "wdgs_device_select_lane: null device");
    WDGS_REQUIRE(lane >= 0 && lane < WDGS_MAX_LANES, WDGS_E_INVALID, "wdgs_device_select_lane: lane %d (0..%d)", lane, WDGS_MAX_LANES - 1);
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_device_select_lane while recording a command buffer");
    WDGS_CHECK_HIP(hipSetDevice(d->ordinal));
    if (!d->lanes[lane]) WDGS_CHECK_HIP(hipStreamCreateWithFlags(&d->lanes[lane], hipStreamNonBlocking));
    d->stream = d->lanes[lane];
    d->lane = lane;
    return WDGS_OK;
}

int wdgs_device_lane_order(wdgs_device* d, int waiter, int signal) {
    WDGS_REQUIRE(d, WDGS_E_INVALID, "wdgs_device_lane_order: null device");
    WDGS_REQUIRE(waiter >= 0 && waiter < WDGS_MAX_LANES && signal >= 0 && signal < WDGS_MAX_LANES, WDGS_E_INVALID, "wdgs_device_lane_order: lanes %d, %d (0..%d)", waiter,
                 signal, WDGS_MAX_LANES - 1);
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_device_lane_order while recording a command buffer");
    if (waiter == signal || !d->lanes[signal]) return WDGS_OK;  // a lane is in order with itself; a lane never used has nothing pending
    WDGS_CHECK_HIP(hipSetDevice(d->ordinal));
    if (!d->lanes[waiter]) WDGS_CHECK_HIP(hipStreamCreateWithFlags(&d->lanes[waiter], hipStreamNonBlocking));
    // one event per signalling lane: a wait refers to the record that precedes it, so re-recording the event later is safe
    if (!d->lane_events[signal]) WDGS_CHECK_HIP(hipEventCreateWithFlags(&d->lane_events[signal], hipEventDisableTiming));
    WDGS_CHECK_HIP(hipEventRecord(d->lane_events[signal], d->lanes[signal]));
    WDGS_CHECK_HIP(hipStreamWaitEvent(d->lanes[waiter], d->lane_events[signal], 0));
    return WDGS_OK;
}

// A position on a lane, kept in one of WDGS_MAX_BATCH_VIEWS numbered marks, that other lanes can be made to wait for LATER (lane_order
// records and waits in one call): a batched step projects its view groups one after the other on one lane and lets each view wait only
// for its own group's projection.
int wdgs_device_lane_mark(wdgs_device* d, int lane, int mark) {
    WDGS_REQUIRE(d && lane >= 0 && lane < WDGS_MAX_LANES && mark >= 0 && mark < WDGS_MAX_BATCH_VIEWS, WDGS_E_INVALID, "wdgs_device_lane_mark: lane %d, mark %d", lane, mark);
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_device_lane_mark while recording a command buffer");
    WDGS_CHECK_HIP(hipSetDevice(d->ordinal));
    if (!d->lanes[lane]) WDGS_CHECK_HIP(hipStreamCreateWithFlags(&d->lanes[lane], hipStreamNonBlocking));
    if (!d->lane_marks[mark]) WDGS_CHECK_HIP(hipEventCreateWithFlags(&d->lane_marks[mark], hipEventDisableTiming));
    WDGS_CHECK_HIP(hipEventRecord(d->lane_marks[mark], d->lanes[lane]));
    return WDGS_OK;
}
int wdgs_device_lane_wait_mark(wdgs_device* d, int lane, int mark) {
    WDGS_REQUIRE(d && lane >= 0 && lane < WDGS_MAX_LANES && mark >= 0 && mark < WDGS_MAX_BATCH_VIEWS, WDGS_E_INVALID, "wdgs_device_lane_wait_mark: lane %d, mark %d", lane, mark);
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_device_lane_wait_mark while recording a command buffer");
    WDGS_REQUIRE(d->lane_marks[mark], WDGS_E_STATE, "wdgs_device_lane_wait_mark: mark %d was never set", mark);
    WDGS_CHECK_HIP(hipSetDevice(d->ordinal));
    if (!d->lanes[lane]) WDGS_CHECK_HIP(hipStreamCreateWithFlags(&d->lanes[lane], hipStreamNonBlocking));
    WDGS_CHECK_HIP(hipStreamWaitEvent(d->lanes[lane], d->lane_marks[mark], 0));
    return WDGS_OK;
}

// Folds the finished event pairs into the per-kernel totals.  Pairs whose kernels have not finished yet (a wait for an earlier ticket,
// a query between steps) stay pending: events of one queue complete in order, so the scan stops at the first unfinished pair.
static int collect_profile(wdgs_device* d) {
    size_t done = 0;
    for (auto& p : d->pending) {
        float ms = 0.f;
        const hipError_t e = hipEventElapsedTime(&ms, p.a, p.b);
        if (e == hipErrorNotReady) { (void)hipGetLastError(); break; }
        if (e == hipSuccess) {
            auto& t = d->totals[p.name];
            t.launches += 1;
            t.ms += ms;
        }
        d->event_pool.push_back(p.a);
        d->event_pool.push_back(p.b);
        done++;
    }
    d->pending.erase(d->pending.begin(), d->pending.begin() + (long)done);
    return WDGS_OK;
}

// Device-side conditions that are reported at the next host wait: a truncated tile-entry list (per forward pass, sticky since its last
// check) and a guarded optimizer step that skipped itself (sticky; set when the guard word of a batched / data-parallel step was
// non-zero -- on this rank or, after the exchange summed it, on any other).  Reading consumes them.
static int deferred_checks(wdgs_device* d) {
    // The words are consumed with an atomic exchange: a wait for step k's ticket runs while step k+1 is in flight (pipeline depth 2),
    // and a read followed by a plain store of 0 would lose a note the device writes between the two.  The device writes
    // whole aligned words into pinned host memory; the exchange either sees the note (and reports it now) or leaves it for the next check.
    const u32 skipped = __atomic_exchange_n(d->host_guard, 0u, __ATOMIC_ACQ_REL);
    // every pass's word is consumed; the report names every pass that overflowed (up to four), so that a host whose passes share the device with
    // another owner's -- a Trainer beside a Viewer -- can tell its own
    char report[384];
    size_t at = 0;
    u32 found = 0u;
    for (wdgs_tiled_forward* f : d->forwards) {
        if (!f->encoded) continue;
        // written by update_stats before the stream drained (no device round trip here); sticky across the encodes since the last
        // check: the exchange consumes it
        const u32 v = __atomic_exchange_n(f->host_stats + 2, 0u, __ATOMIC_ACQ_REL);
        if (v != 0u && found < 4u && at < sizeof(report)) {
            const int w = std::snprintf(report + at, sizeof(report) - at, "%s%u entries needed, max_tile_entries = %u (forward pass %p)", found ? "; " : "", v,
                                        f->tile_info.max_tile_entries, (void*)f);
            if (w > 0) at += (size_t)w;
            found++;
        }
    }
    WDGS_REQUIRE(found == 0u, WDGS_E_CAPACITY, "tile entries overflow: %s (raise wdgs_tiled_forward_config.max_tile_entries)", report);
    WDGS_REQUIRE(skipped == 0u, WDGS_E_CAPACITY, "an optimizer step was skipped on every rank: tile entries overflowed on another rank (its own error names the size)");
    return WDGS_OK;
}

int wdgs_device_synchronize(wdgs_device* d) {
    WDGS_REQUIRE(d, WDGS_E_INVALID, "wdgs_device_synchronize: null device");
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_device_synchronize while recording a command buffer");
    WDGS_CHECK_HIP(hipSetDevice(d->ordinal));
    WDGS_CHECK_HIP(wdgs_sync_lanes(d));
    collect_profile(d);
    return deferred_checks(d);
}

int wdgs_queue_mark(wdgs_device* d, uint64_t* ticket) {
    WDGS_REQUIRE(d && ticket, WDGS_E_INVALID, "wdgs_queue_mark: null argument");
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_queue_mark while recording a command buffer");
    WDGS_CHECK_HIP(hipSetDevice(d->ordinal));
    const uint64_t t = d->ticket_next;
    hipEvent_t& e = d->ticket_events[t % WDGS_TICKET_RING];
    if (!e) WDGS_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    WDGS_CHECK_HIP(hipEventRecord(e, d->stream));
    d->ticket_next = t + 1;
    *ticket = t;
    return WDGS_OK;
}

int wdgs_queue_wait(wdgs_device* d, uint64_t ticket) {
    WDGS_REQUIRE(d, WDGS_E_INVALID, "wdgs_queue_wait: null device");
    WDGS_REQUIRE(ticket != 0 && ticket < d->ticket_next, WDGS_E_INVALID, "wdgs_queue_wait: ticket %llu was never issued", (unsigned long long)ticket);
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_queue_wait while recording a command buffer");
    // (a ticket older than the ring waits on the mark that took its slot: later in the same queue, so the wait still holds)
    WDGS_CHECK_HIP(hipEventSynchronize(d->ticket_events[ticket % WDGS_TICKET_RING]));
    if (!d->pending.empty()) collect_profile(d);
    return deferred_checks(d);
}

int wdgs_device_memory_info(wdgs_device* d, size_t* free_bytes, size_t* total_bytes, size_t* cached_bytes) {
    WDGS_REQUIRE(d, WDGS_E_INVALID, "wdgs_device_memory_info: null device");
    size_t f = 0, t = 0;
    WDGS_CHECK_HIP(hipSetDevice(d->ordinal));
    WDGS_CHECK_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    if (cached_bytes) *cached_bytes = g_alloc_cache.held(d->ordinal);
    return WDGS_OK;
}
static void reap_command_buffers(wdgs_device* d, size_t at_most);   // (defined with the command buffers, below)
int wdgs_device_destroy(wdgs_device* d) {
    if (!d) return WDGS_OK;
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        if (!g_live_devices.erase(d)) return WDGS_OK;  // already destroyed (or never ours): idempotent
    }
    (void)hipSetDevice(d->ordinal);
    if (d->capturing) {  // an abandoned recording: end it so the stream is usable by its owner again
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(d->stream, &g);
        if (g) (void)hipGraphDestroy(g);
        d->capturing = false;
    }
    (void)wdgs_sync_lanes(d);
    collect_profile(d);
    reap_command_buffers(d, (size_t)-1);
    g_alloc_cache.release(d->ordinal);
    for (hipEvent_t e : d->event_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : d->lane_events) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : d->lane_marks) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : d->ticket_events) if (e) (void)hipEventDestroy(e);
    if (d->host_guard) (void)hipHostFree(d->host_guard);
    for (int l = 1; l < WDGS_MAX_LANES; l++)
        if (d->lanes[l]) (void)hipStreamDestroy(d->lanes[l]);
    if (d->own_stream) (void)hipStreamDestroy(d->lanes[0]);
    delete d;
    return WDGS_OK;
}

int wdgs_device_set_profiling(wdgs_device* d, int enabled) {
    WDGS_REQUIRE(d, WDGS_E_INVALID, "null device");
    d->profiling = enabled != 0;
    return WDGS_OK;
}

int wdgs_device_get_kernel_times(wdgs_device* d, wdgs_kernel_time* out, uint32_t cap, uint32_t* count) {
    WDGS_REQUIRE(d && count, WDGS_E_INVALID, "null argument");
    if (!d->pending.empty() && !d->capturing) collect_profile(d);  // (whatever has finished by now; a synchronize before the call makes that everything)
    u32 i = 0;
    for (auto& kv : d->totals) {
        if (out && i < cap) {
            std::memset(&out[i], 0, sizeof(out[i]));
            std::strncpy(out[i].name, kv.first.c_str(), sizeof(out[i].name) - 1);
            out[i].launches = kv.second.launches;
            out[i].total_ms = kv.second.ms;
        }
        i++;
    }
    *count = i;
    return WDGS_OK;
}

int wdgs_device_reset_kernel_times(wdgs_device* d) {
    WDGS_REQUIRE(d, WDGS_E_INVALID, "null device");
    d->totals.clear();
    return WDGS_OK;
}

// ---------------------------------------------------------------- recorded command buffers
static bool forward_holds_projection(const wdgs_tiled_forward* f);
static void forward_consume_projection(wdgs_tiled_forward* f);
// `consumes`: the forward passes whose projection (wdgs_tiled_forward_project_views) a replay of this recording uses up -- its scan works in
// place on K1's workgroup sums.  The validity flag of a pass is a HOST fact checked when an encode is made; a replay makes no encode call, so
// the command buffer carries the check to wdgs_queue_submit.  Without it a recording that starts at the scan could be replayed on a pass whose
// projection another encode had already consumed: offsets scanned twice, a tile-entry count far beyond the entries really written, stale keys
// with a zero tile field among them, and sort_scatter's atomicMin(&ranges[(key >> 16) - 1]) landing 16 GB past the table -- the device fault
// behind the abort of gpurun_out/r04o_tests.log.
struct wdgs_command_buffer_impl { hipGraph_t graph; hipGraphExec_t exec; std::vector<wdgs_tiled_forward*> consumes; wdgs_device* dev; };
static void command_buffer_release(wdgs_command_buffer_impl* c) {
    (void)hipGraphExecDestroy(c->exec);
    (void)hipGraphDestroy(c->graph);
    delete c;
}
static void reap_command_buffers(wdgs_device* d, size_t at_most) {
    while (at_most-- && !d->dead_command_buffers.empty()) {
        command_buffer_release(static_cast<wdgs_command_buffer_impl*>(d->dead_command_buffers.back()));
        d->dead_command_buffers.pop_back();
    }
}

int wdgs_encoder_begin(wdgs_device* d) {
    WDGS_REQUIRE(d, WDGS_E_INVALID, "null device");
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_encoder_begin: a recording is already open on this device");
    WDGS_CHECK_HIP(hipStreamBeginCapture(d->stream, hipStreamCaptureModeRelaxed));
    d->capturing = true;
    d->capture_consumes.clear();
    return WDGS_OK;
}
int wdgs_encoder_finish(wdgs_device* d, wdgs_command_buffer** out) {
    WDGS_REQUIRE(d && out, WDGS_E_INVALID, "null argument");
    WDGS_REQUIRE(d->capturing, WDGS_E_STATE, "wdgs_encoder_finish without wdgs_encoder_begin");
    d->capturing = false;
    hipGraph_t graph = nullptr;
    WDGS_CHECK_HIP(hipStreamEndCapture(d->stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(graph); d->capture_consumes.clear(); wdgs_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e)); return WDGS_E_HIP; }
    auto* c = new wdgs_command_buffer_impl{graph, exec, std::move(d->capture_consumes), d};
    d->capture_consumes.clear();
    *out = reinterpret_cast<wdgs_command_buffer*>(c);
    return WDGS_OK;
}
int wdgs_encoder_abort(wdgs_device* d) {
    WDGS_REQUIRE(d, WDGS_E_INVALID, "null device");
    if (!d->capturing) return WDGS_OK;  // nothing open: harmless, so error paths may call it unconditionally
    d->capturing = false;
    d->capture_consumes.clear();
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(d->stream, &graph);  // an invalidated capture reports an error here and yields no graph
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();  // the failed encode's sticky error, if any, belongs to the aborted recording
    (void)e;
    return WDGS_OK;
}
int wdgs_queue_submit(wdgs_device* d, wdgs_command_buffer* cmd) {
    WDGS_REQUIRE(d && cmd, WDGS_E_INVALID, "null argument");
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_queue_submit while recording");
    auto* c = reinterpret_cast<wdgs_command_buffer_impl*>(cmd);
    // a recording that contains wdgs_tiled_forward_encode_projected uses up the projection of its pass on EVERY replay: all must be there
    // (checked before anything is consumed, so a refused submit changes nothing)
    for (wdgs_tiled_forward* f : c->consumes) {
        WDGS_REQUIRE(std::find(d->forwards.begin(), d->forwards.end(), f) != d->forwards.end(), WDGS_E_STATE,
                     "wdgs_queue_submit: the command buffer was recorded against a forward pass that has been destroyed");
        WDGS_REQUIRE(forward_holds_projection(f), WDGS_E_STATE,
                     "wdgs_queue_submit: the command buffer starts at the scan of a projected forward pass (wdgs_tiled_forward_encode_projected), but the pass holds no "
                     "projection: run wdgs_tiled_forward_project_views before every submit, and no other encode of the pass in between");
    }
    WDGS_CHECK_HIP(hipGraphLaunch(c->exec, d->stream));
    for (wdgs_tiled_forward* f : c->consumes) forward_consume_projection(f);   // (only a replay that was really enqueued uses the projections up)
    reap_command_buffers(d, 1);   // (behind the launch: the device is busy with it while the host pays)
    return WDGS_OK;
}
namespace {
struct DoneThunk { wdgs_done_callback fn; void* user; };
void done_trampoline(void* p) {
    DoneThunk* t = static_cast<DoneThunk*>(p);
    t->fn(t->user);
    delete t;
}
}  // namespace
int wdgs_queue_on_done(wdgs_device* d, wdgs_done_callback fn, void* user) {
    WDGS_REQUIRE(d && fn, WDGS_E_INVALID, "wdgs_queue_on_done: null argument");
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_queue_on_done while recording a command buffer");
    DoneThunk* t = new DoneThunk{fn, user};
    hipError_t e = hipLaunchHostFunc(d->stream, done_trampoline, t);
    if (e != hipSuccess) { delete t; wdgs_set_error("hipLaunchHostFunc failed: %s", hipGetErrorString(e)); return WDGS_E_HIP; }
    return WDGS_OK;
}
int wdgs_command_buffer_destroy(wdgs_command_buffer* cmd) {
    if (!cmd) return WDGS_OK;
    auto* c = reinterpret_cast<wdgs_command_buffer_impl*>(cmd);
    // WDGS_LAZY_GRAPH_DESTROY=0: destroy here and now
    static const bool lazy = !(std::getenv("WDGS_LAZY_GRAPH_DESTROY") && std::getenv("WDGS_LAZY_GRAPH_DESTROY")[0] == '0');
    if (lazy && wdgs_device_alive(c->dev)) {
        c->dev->dead_command_buffers.push_back(c);
        if (c->dev->dead_command_buffers.size() > 64u) reap_command_buffers(c->dev, 32u);   // (a host that never submits again must not pile them up)
    } else command_buffer_release(c);
    return WDGS_OK;
}

int wdgs_copy_to_host(wdgs_device* d, void* dst, const void* src, size_t bytes) {
    WDGS_REQUIRE(d && (bytes == 0 || (dst && src)), WDGS_E_INVALID, "wdgs_copy_to_host: null argument");
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_copy_to_host while recording a command buffer");
    if (bytes == 0) return WDGS_OK;
    WDGS_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, d->stream));
    WDGS_CHECK_HIP(wdgs_sync_lanes(d));
    return WDGS_OK;
}

int wdgs_copy_to_device(wdgs_device* d, void* dst, const void* src, size_t bytes) {
    WDGS_REQUIRE(d && (bytes == 0 || (dst && src)), WDGS_E_INVALID, "wdgs_copy_to_device: null argument");
    // a recording would bake the (pageable, soon reused) host pointer into the graph: uploads belong before wdgs_encoder_begin
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_copy_to_device while recording a command buffer");
    if (bytes == 0) return WDGS_OK;
    // pageable source: hipMemcpyAsync stages the copy before returning, so `src` may be reused immediately
    WDGS_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, d->stream));
    return WDGS_OK;
}

int wdgs_memset(wdgs_device* d, void* dst, int value, size_t bytes) {
    WDGS_REQUIRE(d && (bytes == 0 || dst), WDGS_E_INVALID, "wdgs_memset: null argument");
    if (bytes == 0) return WDGS_OK;
    // allowed inside a recording: it becomes a memset node of the command buffer (encoder.clearBuffer is recorded in the reference too)
    WDGS_CHECK_HIP(hipMemsetAsync(dst, value, bytes, d->stream));
    return WDGS_OK;
}

int wdgs_copy_buffer_to_buffer(wdgs_device* d, void* dst, const void* src, size_t bytes) {
    WDGS_REQUIRE(d && (bytes == 0 || (dst && src)), WDGS_E_INVALID, "wdgs_copy_buffer_to_buffer: null argument");
    if (bytes == 0) return WDGS_OK;
    // device to device, stream-ordered; inside a recording it becomes a copy node (encoder.copyBufferToBuffer is recorded in the reference too)
    WDGS_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, d->stream));
    return WDGS_OK;
}

// ---------------------------------------------------------------- buffers
int wdgs_buffer_create(wdgs_device* d, size_t bytes, wdgs_buffer** out) {
    WDGS_REQUIRE(d && out, WDGS_E_INVALID, "wdgs_buffer_create: null argument");
    wdgs_buffer* b = new wdgs_buffer{nullptr, bytes};
    int r = wdgs_alloc(&b->ptr, bytes, true, d->stream);
    if (r != WDGS_OK) { delete b; return r; }
    *out = b;
    return WDGS_OK;
}
int wdgs_buffer_destroy(wdgs_buffer* b) {
    if (!b) return WDGS_OK;
    free_dev(b->ptr);
    delete b;
    return WDGS_OK;
}
void* wdgs_buffer_ptr(const wdgs_buffer* b) { return b ? b->ptr : nullptr; }
size_t wdgs_buffer_size(const wdgs_buffer* b) { return b ? b->size : 0; }
int wdgs_buffer_write(wdgs_device* d, wdgs_buffer* b, size_t off, const void* src, size_t bytes) {
    WDGS_REQUIRE(d && b, WDGS_E_INVALID, "wdgs_buffer_write: null argument");
    WDGS_REQUIRE(off + bytes <= b->size, WDGS_E_INVALID, "wdgs_buffer_write: range [%zu, %zu) exceeds buffer size %zu", off, off + bytes, b->size);
    return wdgs_copy_to_device(d, (char*)b->ptr + off, src, bytes);
}
int wdgs_buffer_read(wdgs_device* d, const wdgs_buffer* b, size_t off, void* dst, size_t bytes) {
    WDGS_REQUIRE(d && b, WDGS_E_INVALID, "wdgs_buffer_read: null argument");
    WDGS_REQUIRE(off + bytes <= b->size, WDGS_E_INVALID, "wdgs_buffer_read: range [%zu, %zu) exceeds buffer size %zu", off, off + bytes, b->size);
    return wdgs_copy_to_host(d, dst, (const char*)b->ptr + off, bytes);
}
int wdgs_buffer_read_async(wdgs_device* d, const wdgs_buffer* b, size_t off, void* dst, size_t bytes) {
    WDGS_REQUIRE(d && b && (bytes == 0 || dst), WDGS_E_INVALID, "wdgs_buffer_read_async: null argument");
    WDGS_REQUIRE(off + bytes <= b->size, WDGS_E_INVALID, "wdgs_buffer_read_async: range [%zu, %zu) exceeds buffer size %zu", off, off + bytes, b->size);
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_buffer_read_async while recording a command buffer");
    if (bytes == 0) return WDGS_OK;
    WDGS_CHECK_HIP(hipMemcpyAsync(dst, (const char*)b->ptr + off, bytes, hipMemcpyDeviceToHost, d->stream));
    return WDGS_OK;
}
int wdgs_host_alloc(size_t bytes, void** out) {
    WDGS_REQUIRE(out, WDGS_E_INVALID, "wdgs_host_alloc: null argument");
    WDGS_CHECK_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return WDGS_OK;
}
int wdgs_host_free(void* p) {
    if (p) WDGS_CHECK_HIP(hipHostFree(p));
    return WDGS_OK;
}

// ---------------------------------------------------------------- prefix scanner
int wdgs_prefix_scanner_create(wdgs_device* d, uint32_t max_elements, wdgs_prefix_scanner** out) {
    WDGS_REQUIRE(d && out, WDGS_E_INVALID, "wdgs_prefix_scanner_create: null argument");
    wdgs_prefix_scanner* s = new wdgs_prefix_scanner();
    s->dev = d;
    s->max_elements = max_elements > 0 ? max_elements : 1;
    s->count = s->max_elements;
    s->input = s->output = nullptr;
    int r = wdgs_alloc((void**)&s->input, sizeof(u32) * (size_t)s->max_elements, true, d->stream);
    if (r == WDGS_OK) r = wdgs_alloc((void**)&s->output, sizeof(u32) * (size_t)s->max_elements, true, d->stream);
    if (r == WDGS_OK) r = scan_scratch_create(&s->scratch, s->max_elements);
    if (r != WDGS_OK) { wdgs_prefix_scanner_destroy(s); return r; }
    *out = s;
    return WDGS_OK;
}
int wdgs_prefix_scanner_destroy(wdgs_prefix_scanner* s) {
    if (!s) return WDGS_OK;
    free_dev(s->input);
    free_dev(s->output);
    scan_scratch_destroy(&s->scratch);
    delete s;
    return WDGS_OK;
}
void* wdgs_prefix_scanner_input(wdgs_prefix_scanner* s) { return s ? s->input : nullptr; }
void* wdgs_prefix_scanner_output(wdgs_prefix_scanner* s) { return s ? s->output : nullptr; }
int wdgs_prefix_scanner_set_count(wdgs_prefix_scanner* s, uint32_t count) {
    WDGS_REQUIRE(s, WDGS_E_INVALID, "null scanner");
    WDGS_REQUIRE(count <= s->max_elements, WDGS_E_CAPACITY, "scan count %u exceeds max_elements %u", count, s->max_elements);
    s->count = count;
    return WDGS_OK;
}
int wdgs_prefix_scanner_scan(wdgs_prefix_scanner* s) {
    WDGS_REQUIRE(s, WDGS_E_INVALID, "null scanner");
    return scan_exclusive_u32(s->dev, &s->scratch, s->input, s->output, s->count, nullptr);
}
int wdgs_prefix_scanner_scan_ptr(wdgs_prefix_scanner* s, const void* in, void* out, uint32_t count) {
    WDGS_REQUIRE(s && in && out, WDGS_E_INVALID, "null argument");
    WDGS_REQUIRE(count <= s->max_elements, WDGS_E_CAPACITY, "scan count %u exceeds max_elements %u", count, s->max_elements);
    return scan_exclusive_u32(s->dev, &s->scratch, (const u32*)in, (u32*)out, count, nullptr);
}

// ---------------------------------------------------------------- TiledForwardPass
// Tile entries a pass over n Gaussians may produce before it reports WDGS_E_CAPACITY.
static uint64_t forward_tile_entry_cap(const wdgs_tiled_forward_config& cfg, u32 n) {
    uint64_t cap;
    if (cfg.compat_caps) {  // tiled-forward-pass.ts:137-154
        uint64_t base = std::min<uint64_t>((uint64_t)n * 30, (uint64_t)n * 2048);
        cap = std::min<uint64_t>(std::min<uint64_t>(base, 32ull * 1024 * 1024), 2097152ull);
        cap = (cap + 3839) / 3840 * 3840;
    } else if (cfg.max_tile_entries) {
        cap = cfg.max_tile_entries;
    } else {
        cap = std::max<uint64_t>((uint64_t)n * 30, 1ull << 20);
    }
    return std::min<uint64_t>(align_up(cap, 4096), 0xFFFFF000ull);
}
static int forward_alloc_per_point(wdgs_tiled_forward* op, u32 capacity) {
    wdgs_device* d = op->dev;
    int r = wdgs_alloc((void**)&op->splats, (size_t)24 * capacity, true, d->stream);
    if (r == WDGS_OK) r = wdgs_alloc((void**)&op->depths, (size_t)4 * capacity, true, d->stream);
    if (r == WDGS_OK) r = wdgs_alloc((void**)&op->block_counts, (size_t)4 * (ceil_div(capacity, 256) + 1), true, d->stream);
    if (r == WDGS_OK) r = wdgs_alloc((void**)&op->column_counts, (size_t)4 * 256 * (ceil_div(capacity, 256) + 1), true, d->stream);
    if (r == WDGS_OK && !op->column_totals) r = wdgs_alloc((void**)&op->column_totals, (size_t)4 * 256, true, d->stream);
    if (r == WDGS_OK) r = wdgs_prefix_scanner_create(d, capacity, &op->scanner);
    if (r == WDGS_OK) op->points_capacity = capacity;
    return r;
}
static void forward_set_viewport(wdgs_tiled_forward* op, u32 w, u32 h) {
    op->cfg.viewport_width = w;
    op->cfg.viewport_height = h;
    op->settings.viewport_x = (float)w;
    op->settings.viewport_y = (float)h;
    op->tile_info.num_tiles_x = ceil_div(w, 16);
    op->tile_info.num_tiles_y = ceil_div(h, 16);
    op->tile_info.total_tiles = op->tile_info.num_tiles_x * op->tile_info.num_tiles_y;
}
static int forward_alloc_nf_stamp(wdgs_tiled_forward* op) {
    if (op->tile_info.total_tiles <= op->nf_capacity) return WDGS_OK;
    free_dev(op->nf_stamp);
    op->nf_stamp = nullptr;
    op->nf_capacity = 0;
    WDGS_TRY(wdgs_alloc((void**)&op->nf_stamp, sizeof(u32) * (size_t)op->tile_info.total_tiles, true, op->dev->stream));
    op->nf_capacity = op->tile_info.total_tiles;
    return WDGS_OK;
}

// The long-list work of a pass lives in ONE device allocation (twelve tables carved out of it: a pass is built inside a densify event, where every
// hipMalloc counts) plus the per-tile marks, which follow the viewport.
static void forward_free_long_lists(wdgs_tiled_forward* op) {
    LongWork& lw = op->long_lists;
    free_dev(lw.hdr);     // (the base of the allocation)
    free_dev(lw.flags);
    lw = LongWork{};
    op->long_flags_capacity = 0;
}
// (Re)allocates the long-list work of a pass: `threshold` entries (0 = off), room for `items` (block, chunk) slots and `rows` list rows.
static int forward_alloc_long_lists(wdgs_tiled_forward* op, u32 threshold, u32 items, u32 rows) {
    wdgs_device* d = op->dev;
    forward_free_long_lists(op);
    if (threshold == 0u) return WDGS_OK;
    LongWork lw{};
    lw.threshold = threshold;
    lw.max_items = std::max(items, 64u);
    lw.max_blocks = std::max(lw.max_items / 8u, 16u) & ~3u;   // (four per long tile; a tile of `threshold` > 128 entries takes more than 8 items per block)
    lw.max_rows = std::max(rows, 64u);
    const u32 tiles = std::max(op->tile_info.total_tiles, 1u);
    const size_t I = lw.max_items, B = lw.max_blocks, R = lw.max_rows;
    // the small tables first (they are zeroed), the three large ones behind them (written before they are read)
    const size_t sizes[12] = {256, sizeof(LongBlock) * B, sizeof(LongSync) * B, sizeof(u32) * I, sizeof(u32) * I, sizeof(u32) * 64 * B, sizeof(u32) * 64 * B,
                              sizeof(u32) * 64 * I, sizeof(u32) * 64 * I, sizeof(float4) * 192 * I, sizeof(u32) * 256 * R, 0};
    size_t at[12], total = 0, zeroed = 0;
    for (int i = 0; i < 11; i++) { at[i] = total; total += align_up(sizes[i], 256); if (i == 6) zeroed = total; }
    char* base = nullptr;
    WDGS_TRY(wdgs_alloc((void**)&base, total, false, d->stream));
    if (hipMemsetAsync(base, 0, zeroed, d->stream) != hipSuccess) { free_dev(base); wdgs_set_error("hipMemsetAsync failed"); return WDGS_E_HIP; }
    lw.hdr = (u32*)(base + at[0]); lw.blocks = (LongBlock*)(base + at[1]); lw.sync = (LongSync*)(base + at[2]); lw.item_block = (u32*)(base + at[3]); lw.nlist = (u32*)(base + at[4]);
    lw.total = (u32*)(base + at[5]); lw.jlast = (u32*)(base + at[6]); lw.cnt = (u32*)(base + at[7]); lw.off = (u32*)(base + at[8]); lw.records = (float4*)(base + at[9]);
    lw.rows = (u32*)(base + at[10]);
    lw.nf_stamp = op->nf_stamp;
    lw.nf_frame = op->stats + FRAME_WORD;
    op->long_lists = lw;   // (what has been allocated is freed with the pass, also after a failure)
    const int r = wdgs_alloc((void**)&op->long_lists.flags, sizeof(u32) * (size_t)tiles, true, d->stream);
    if (r != WDGS_OK) { forward_free_long_lists(op); return r; }
    op->long_flags_capacity = tiles;
    return WDGS_OK;
}
constexpr u32 LONG_LIST_THRESHOLD = 2048u, LONG_LIST_ITEMS = 1024u, LONG_LIST_ROWS = 8192u;   // defaults (include/webdgs.h: wdgs_tiled_forward_set_long_lists)

int wdgs_tiled_forward_create(wdgs_device* d, const wdgs_tiled_forward_config* cfg, wdgs_tiled_forward** out) {
    WDGS_REQUIRE(d && cfg && out, WDGS_E_INVALID, "wdgs_tiled_forward_create: null argument");
    WDGS_REQUIRE(cfg->viewport_width > 0 && cfg->viewport_height > 0, WDGS_E_INVALID, "viewport must be non-empty");
    WDGS_REQUIRE(cfg->sh_deg <= 3, WDGS_E_INVALID, "sh_deg %u > 3", cfg->sh_deg);
    // the sort key keeps tile_id + 1 in 16 bits (tiled-forward.wgsl:121-136)
    const uint64_t tiles = (uint64_t)ceil_div(cfg->viewport_width, 16) * ceil_div(cfg->viewport_height, 16);
    WDGS_REQUIRE(tiles + 1 <= 0xFFFFu, WDGS_E_CAPACITY, "%llu tiles do not fit the 16-bit tile field of the sort key", (unsigned long long)tiles);
    wdgs_tiled_forward* op = new wdgs_tiled_forward();
    op->dev = d;
    op->cfg = *cfg;
    op->stats = op->splats = op->depths = op->block_counts = op->column_counts = op->column_totals = nullptr;
    op->dc_source = nullptr;
    op->nf_stamp = nullptr;
    op->nf_capacity = 0;
    op->long_lists = LongWork{};
    op->long_flags_capacity = 0;
    op->host_stats = nullptr;
    op->scanner = nullptr;
    op->sorter = nullptr;
    op->ranges = nullptr;
    op->ranges_capacity = 0;
    op->ranges_valid = false;
    op->encoded = false;
    op->projected = op->projected_columns = false;
    const u32 n = cfg->num_points;
    const uint64_t cap = forward_tile_entry_cap(*cfg, n);
    op->points_capacity = std::max(n, 1u);
    op->settings = RenderSettings{cfg->gaussian_scale != 0.f ? cfg->gaussian_scale : 1.0f, (float)cfg->sh_deg, 0.f, 0.f,
                                  cfg->point_size_px != 0.f ? cfg->point_size_px : 3.0f, cfg->render_mode ? 1.0f : 0.0f,
                                  cfg->max_splat_radius_px != 0.f ? cfg->max_splat_radius_px : 128.0f};
    op->tile_info.max_tile_entries = (u32)cap;
    forward_set_viewport(op, cfg->viewport_width, cfg->viewport_height);
    int r = wdgs_alloc((void**)&op->stats, FORWARD_STATS_BYTES, true, d->stream);
    // (frame numbers start at 1: a zeroed stamp table marks nothing)
    if (r == WDGS_OK && hipMemsetD32Async((hipDeviceptr_t)(op->stats + FRAME_WORD), 1, 1, d->stream) != hipSuccess) { wdgs_set_error("hipMemsetD32Async failed"); r = WDGS_E_HIP; }
    if (r == WDGS_OK) r = forward_alloc_nf_stamp(op);
    // (WDGS_LONG_LISTS=0: off for every pass of the process; a threshold otherwise -- same-box comparisons)
    static const char* const ll_env = std::getenv("WDGS_LONG_LISTS");
    if (r == WDGS_OK && !cfg->compat_caps) r = forward_alloc_long_lists(op, ll_env ? (u32)std::atoi(ll_env) : LONG_LIST_THRESHOLD, LONG_LIST_ITEMS, LONG_LIST_ROWS);
    if (r == WDGS_OK && hipHostMalloc((void**)&op->host_stats, 16, hipHostMallocDefault) != hipSuccess) { wdgs_set_error("hipHostMalloc(16) failed"); r = WDGS_E_HIP; }
    if (r == WDGS_OK) std::memset(op->host_stats, 0, 16);
    if (r == WDGS_OK) r = forward_alloc_per_point(op, std::max(n, 1u));
    if (r == WDGS_OK) r = wdgs_sorter_create(d, (u32)cap, op->stats, &op->sorter);
    if (r != WDGS_OK) { wdgs_tiled_forward_destroy(op); return r; }
    d->forwards.push_back(op);
    *out = op;
    return WDGS_OK;
}

int wdgs_tiled_forward_destroy(wdgs_tiled_forward* op) {
    if (!op) return WDGS_OK;
    if (wdgs_device_alive(op->dev)) {
        auto& v = op->dev->forwards;
        v.erase(std::remove(v.begin(), v.end(), op), v.end());
    }
    sync_if_alive(op->dev);
    if (op->ranges && wdgs_device_alive(op->dev)) op->dev->range_tables.erase(op->ranges);
    free_dev(op->stats);
    free_dev(op->nf_stamp);
    forward_free_long_lists(op);
    if (op->host_stats) (void)hipHostFree(op->host_stats);
    free_dev(op->splats);
    free_dev(op->depths);
    free_dev(op->block_counts);
    free_dev(op->column_counts);
    free_dev(op->column_totals);
    free_dev(op->ranges);
    wdgs_prefix_scanner_destroy(op->scanner);
    wdgs_sorter_destroy(op->sorter);
    delete op;
    return WDGS_OK;
}

// The point cloud changed size (densify / prune: applyPointCloudSwap, trainer.ts:201-237, which destroys the pass and builds a new
// one).  Here the pass is kept: its per-Gaussian buffers and its sort buffers are reused when they are large enough and re-allocated
// with 25 % headroom when they are not, so a training run re-allocates a few times at most.  After the call the pass is in the state
// of a freshly created one for n points (zeroed buffers, nothing encoded); max_tile_entries follows the same formula as at creation.
int wdgs_tiled_forward_resize(wdgs_tiled_forward* op, uint32_t n) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "wdgs_tiled_forward_resize: null op");
    wdgs_device* d = op->dev;
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_tiled_forward_resize while recording a command buffer");
    WDGS_CHECK_HIP(hipSetDevice(d->ordinal));
    WDGS_CHECK_HIP(wdgs_sync_lanes(d));
    const u32 need = std::max(n, 1u);
    if (need > op->points_capacity) {
        free_dev(op->splats); free_dev(op->depths); free_dev(op->block_counts); free_dev(op->column_counts);
        op->splats = op->depths = op->block_counts = op->column_counts = nullptr;
        wdgs_prefix_scanner_destroy(op->scanner);
        op->scanner = nullptr;
        op->points_capacity = 0;
        WDGS_TRY(forward_alloc_per_point(op, (u32)std::min<uint64_t>((uint64_t)need + need / 4, 0xFFFFFFFFull)));
    } else {
        WDGS_CHECK_HIP(hipMemsetAsync(op->splats, 0, (size_t)24 * need, d->stream));
        WDGS_CHECK_HIP(hipMemsetAsync(op->depths, 0, (size_t)4 * need, d->stream));
        WDGS_CHECK_HIP(hipMemsetAsync(op->block_counts, 0, (size_t)4 * (ceil_div(need, 256) + 1), d->stream));
        WDGS_CHECK_HIP(hipMemsetAsync(op->scanner->input, 0, (size_t)4 * need, d->stream));
        WDGS_CHECK_HIP(hipMemsetAsync(op->scanner->output, 0, (size_t)4 * need, d->stream));
    }
    const uint64_t cap = forward_tile_entry_cap(op->cfg, n);
    if (cap > wdgs_sorter_capacity(op->sorter)) {
        wdgs_sorter_destroy(op->sorter);
        op->sorter = nullptr;
        const uint64_t roomy = op->cfg.compat_caps || op->cfg.max_tile_entries ? cap : std::min<uint64_t>(align_up(cap + cap / 4, 4096), 0xFFFFF000ull);
        WDGS_TRY(wdgs_sorter_create(d, (u32)roomy, op->stats, &op->sorter));
    }
    op->tile_info.max_tile_entries = (u32)cap;
    op->cfg.num_points = n;
    WDGS_CHECK_HIP(hipMemsetAsync(op->stats, 0, 16 + 64 * 4, d->stream));   // (the frame number behind the shards goes on counting)
    std::memset(op->host_stats, 0, 16);
    op->encoded = false;
    op->ranges_valid = false;
    op->projected = false;
    return WDGS_OK;
}

static u32 bits_for(u32 v) {  // number of bits needed to represent v
    u32 b = 0;
    while (v) { b++; v >>= 1; }
    return b;
}

// The tile sort's first pass is folded into its neighbours when the grid allows it (2..256 tile columns, <= 256 tile rows: any
// viewport up to 4096 x 4096): project_count also counts its workgroups' entries per tile COLUMN, the scan kernel turns those into
// per-column offsets, and emit writes its entries straight into column order (project.hip: emit_scatter) -- the keys are never
// written in emission order, histogrammed and scattered.  encode(skipSort) and compat_caps keep the reference's emission order.
static bool forward_uses_columns(const wdgs_tiled_forward* op, int skip_sort) {
    // (WDGS_FORWARD_COLUMNS=0: the separate emit + two-pass tile sort of round 2, for same-box A/B timing; results are identical)
    static const bool columns_enabled = !(std::getenv("WDGS_FORWARD_COLUMNS") && std::getenv("WDGS_FORWARD_COLUMNS")[0] == '0');
    const TileInfo& ti = op->tile_info;
    return columns_enabled && !skip_sort && !op->cfg.compat_caps && op->cfg.num_points > 0 && ti.num_tiles_x >= 2u && ti.num_tiles_x <= 256u && ti.num_tiles_y <= 256u;
}

// Everything of TiledForwardPass.encode behind K1: scan (+ stats), emit, sort.
static int forward_encode_rest(wdgs_tiled_forward* op, int skip_sort, bool columns) {
    wdgs_device* d = op->dev;
    const u32 n = op->cfg.num_points;
    const TileInfo& ti = op->tile_info;
    const ScanStatsEpilogue ep{op->stats, op->stats + 4, op->host_stats, op->tile_info.max_tile_entries, op->stats + FRAME_WORD, op->long_lists.hdr};
    if (columns) {
        WDGS_TRY(forward_scan(d, op->block_counts, ceil_div(n, 256), op->column_counts, op->column_totals, ti.num_tiles_x, ep));
    } else if (n > 0) {
        WDGS_TRY(scan_block_sums_inplace(d, op->block_counts, ceil_div(n, 256), ep));
    } else {
        WDGS_TRY(launch_update_stats(d, n, op->scanner->output, op->scanner->input, op->tile_info.max_tile_entries, op->stats, op->stats + 4, op->host_stats));
    }
    op->ranges_valid = false;
    if (!skip_sort && !op->cfg.compat_caps) {
        const u32 tiles = op->tile_info.total_tiles;
        if (tiles + 1 > op->ranges_capacity) {
            WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "TiledForwardPass.encode allocates its range table on first use: run one eager encode before recording");
            (void)wdgs_sync_lanes(d);
            if (op->ranges) d->range_tables.erase(op->ranges);
            free_dev(op->ranges);
            op->ranges = nullptr;
            WDGS_TRY(wdgs_alloc((void**)&op->ranges, sizeof(u32) * (size_t)(tiles + 1), true, d->stream));
            op->ranges_capacity = tiles + 1;
            d->range_tables[op->ranges] = op;   // (a backward pass handed this table finds the pass's long-list work through it)
        }
    }
    if (columns) {
        WDGS_TRY(launch_emit_scatter(d, n, op->splats, op->depths, op->scanner->input, op->scanner->output, op->block_counts, op->settings, op->tile_info,
                                     op->column_counts, op->column_totals, wdgs_sorter_keys(op->sorter, 0), wdgs_sorter_values(op->sorter, 0), op->tile_info.max_tile_entries));
        // one stable pass on the tile row (it builds the range table), then the per-tile depth sort: the order of a stable sort of the full key
        WDGS_TRY(sorter_sort_rows(op->sorter, ti.num_tiles_x, ti.num_tiles_y, op->ranges, op->long_lists.hdr ? &op->long_lists : nullptr));
        op->ranges_valid = true;
    } else {
        WDGS_TRY(launch_emit(d, n, op->splats, op->depths, op->scanner->input, op->scanner->output, op->block_counts, op->settings, op->tile_info,
                             wdgs_sorter_keys(op->sorter, 0), wdgs_sorter_values(op->sorter, 0), op->tile_info.max_tile_entries));
        if (skip_sort) sorter_set_final_out_index(op->sorter, 0);  // the unsorted entries are in ping-pong 0
        if (!skip_sort) {
            // key = (tile_id + 1) << 16 | depth16: only 16 + bits(total_tiles) bits are ever set
            if (op->cfg.compat_caps) {
                WDGS_TRY(wdgs_sorter_sort(op->sorter, 32u));  // the reference's four 8-bit passes over the whole key
            } else {
                // tile passes, range table, per-tile depth sort (sort.hip): the order of a stable sort of the full key
                WDGS_TRY(sorter_sort_segmented(op->sorter, bits_for(op->tile_info.total_tiles), op->tile_info.total_tiles, op->ranges, op->long_lists.hdr ? &op->long_lists : nullptr));
                op->ranges_valid = true;
            }
        }
    }
    op->encoded = true;
    return WDGS_OK;
}

int wdgs_tiled_forward_encode(wdgs_tiled_forward* op, const void* gaussians, const void* sh, const void* camera, int skip_sort) {
    WDGS_REQUIRE(op && gaussians && sh && camera, WDGS_E_INVALID, "wdgs_tiled_forward_encode: null argument");
    wdgs_device* d = op->dev;
    const u32 n = op->cfg.num_points;
    // clearBuffer(pipelineStatsBuffer) (tiled-forward-pass.ts:345) needs no launch here: update_stats overwrites words 0..2, word 3
    // stays 0, and the visible count is accumulated in shard words that update_stats clears after folding them.
    // The offsets scan (K2-K4) is spread over its neighbours: project_count leaves the entry count of each of its workgroups, one
    // single-workgroup kernel scans those N/256 sums (and publishes the stats block: update_stats, K5, as its epilogue), and emit adds
    // its own in-workgroup prefix -- writing the per-Gaussian offsets table on the way.  Three launches instead of five.
    const bool columns = forward_uses_columns(op, skip_sort);
    op->projected = false;  // (K1 overwrites whatever projection the buffers held)
    WDGS_TRY(launch_project_count(d, n, gaussians, sh, camera, op->settings, op->tile_info, op->splats, op->depths, op->scanner->input, op->stats + 4,
                                  op->block_counts, columns ? op->column_counts : nullptr, op->dc_source, op->nf_stamp, op->stats + FRAME_WORD));
    return forward_encode_rest(op, skip_sort, columns);
}

// ---- view-batched K1 (include/webdgs.h)
int wdgs_tiled_forward_project_views(wdgs_tiled_forward* const* ops, const void* const* cameras, uint32_t count, const void* gaussians, const void* sh) {
    WDGS_REQUIRE(ops && cameras && gaussians && sh && count > 0 && count <= WDGS_MAX_BATCH_VIEWS, WDGS_E_INVALID, "wdgs_tiled_forward_project_views: invalid argument");
    wdgs_tiled_forward* f0 = ops[0];
    WDGS_REQUIRE(f0, WDGS_E_INVALID, "wdgs_tiled_forward_project_views: null pass");
    const bool columns = forward_uses_columns(f0, 0);
    void *splats[WDGS_MAX_BATCH_VIEWS], *depths[WDGS_MAX_BATCH_VIEWS], *counts[WDGS_MAX_BATCH_VIEWS], *shards[WDGS_MAX_BATCH_VIEWS], *blocks[WDGS_MAX_BATCH_VIEWS],
        *cols[WDGS_MAX_BATCH_VIEWS], *stamps[WDGS_MAX_BATCH_VIEWS];
    const void* frames[WDGS_MAX_BATCH_VIEWS];
    for (u32 v = 0; v < count; v++) {
        wdgs_tiled_forward* f = ops[v];
        WDGS_REQUIRE(f && cameras[v], WDGS_E_INVALID, "wdgs_tiled_forward_project_views: null pass or camera %u", v);
        for (u32 u = 0; u < v; u++) WDGS_REQUIRE(ops[u] != f, WDGS_E_INVALID, "wdgs_tiled_forward_project_views: pass %u is given twice (every view needs buffers of its own)", v);
        WDGS_REQUIRE(f->dev == f0->dev && f->cfg.num_points == f0->cfg.num_points && f->cfg.sh_deg == f0->cfg.sh_deg && f->cfg.compat_caps == f0->cfg.compat_caps &&
                         std::memcmp(&f->settings, &f0->settings, sizeof(RenderSettings)) == 0 && f->tile_info.num_tiles_x == f0->tile_info.num_tiles_x &&
                         f->tile_info.num_tiles_y == f0->tile_info.num_tiles_y && f->dc_source == f0->dc_source,
                     WDGS_E_STATE, "wdgs_tiled_forward_project_views: pass %u differs from pass 0 (cloud size, SH degree, viewport, settings or dc source)", v);
        splats[v] = f->splats; depths[v] = f->depths; counts[v] = f->scanner->input; shards[v] = f->stats + 4; blocks[v] = f->block_counts; cols[v] = f->column_counts;
        stamps[v] = f->nf_stamp; frames[v] = f->stats + FRAME_WORD;
    }
    WDGS_TRY(launch_project_count_views(f0->dev, f0->cfg.num_points, count, gaussians, sh, cameras, f0->settings, f0->tile_info, splats, depths, counts, shards, blocks,
                                        columns ? cols : nullptr, f0->dc_source, stamps, frames));
    for (u32 v = 0; v < count; v++) { ops[v]->projected = true; ops[v]->projected_columns = columns; }
    return WDGS_OK;
}
int wdgs_tiled_forward_encode_projected(wdgs_tiled_forward* op) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "wdgs_tiled_forward_encode_projected: null op");
    if (op->dev->capturing) {
        // recorded: nothing runs now, so nothing is consumed now -- every submit of the command buffer will need (and use up) a projection
        // of this pass (wdgs_queue_submit).  Whether the columns were counted is a property of the pass's configuration, not of one projection.
        op->dev->capture_consumes.push_back(op);
        return forward_encode_rest(op, 0, forward_uses_columns(op, 0));
    }
    WDGS_REQUIRE(op->projected, WDGS_E_STATE,
                 "wdgs_tiled_forward_encode_projected: the pass holds no projection (wdgs_tiled_forward_project_views), or a later encode has consumed or overwritten it");
    op->projected = false;  // the scan consumes K1's workgroup sums in place: the rest of the pass can run ONCE per projection
    return forward_encode_rest(op, 0, op->projected_columns);
}
static bool forward_holds_projection(const wdgs_tiled_forward* f) { return f->projected; }
static void forward_consume_projection(wdgs_tiled_forward* f) { f->projected = false; }

int wdgs_tiled_forward_is_projected(const wdgs_tiled_forward* op) { return (op && op->projected) ? 1 : 0; }

int wdgs_tiled_forward_set_viewport(wdgs_tiled_forward* op, uint32_t w, uint32_t h) {
    WDGS_REQUIRE(op && w > 0 && h > 0, WDGS_E_INVALID, "wdgs_tiled_forward_set_viewport: invalid argument");
    WDGS_REQUIRE((uint64_t)ceil_div(w, 16) * ceil_div(h, 16) + 1 <= 0xFFFFu, WDGS_E_CAPACITY, "viewport %ux%u has too many tiles for the 16-bit tile field", w, h);
    WDGS_REQUIRE(!op->dev->capturing, WDGS_E_STATE, "wdgs_tiled_forward_set_viewport while recording a command buffer");
    forward_set_viewport(op, w, h);
    if (op->tile_info.total_tiles > op->nf_capacity) { (void)wdgs_sync_lanes(op->dev); WDGS_TRY(forward_alloc_nf_stamp(op)); op->long_lists.nf_stamp = op->long_lists.hdr ? op->nf_stamp : nullptr; }
    if (op->long_lists.hdr && op->tile_info.total_tiles > op->long_flags_capacity) {
        (void)wdgs_sync_lanes(op->dev);
        free_dev(op->long_lists.flags);
        op->long_lists.flags = nullptr;
        WDGS_TRY(wdgs_alloc((void**)&op->long_lists.flags, sizeof(u32) * (size_t)op->tile_info.total_tiles, true, op->dev->stream));
        op->long_flags_capacity = op->tile_info.total_tiles;
    }
    return WDGS_OK;
}
int wdgs_tiled_forward_set_render_mode(wdgs_tiled_forward* op, uint32_t mode) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "null op");
    op->cfg.render_mode = mode;
    op->settings.gaussian_mode = mode ? 1.0f : 0.0f;
    return WDGS_OK;
}
int wdgs_tiled_forward_set_point_size(wdgs_tiled_forward* op, float v) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "null op");
    op->settings.point_size_px = v;
    return WDGS_OK;
}
int wdgs_tiled_forward_set_gaussian_scale(wdgs_tiled_forward* op, float v) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "null op");
    op->settings.gaussian_scaling = v;
    return WDGS_OK;
}

int wdgs_tiled_forward_get_resources(wdgs_tiled_forward* op, wdgs_tiled_forward_resources* out) {
    WDGS_REQUIRE(op && out, WDGS_E_INVALID, "null argument");
    const int fo = wdgs_sorter_final_out_index(op->sorter);
    out->splat_buffer = op->splats;
    out->depths_buffer = op->depths;
    out->tile_keys_buffer = wdgs_sorter_keys(op->sorter, fo);
    out->tile_indices_buffer = wdgs_sorter_values(op->sorter, fo);
    out->tile_offsets_buffer = op->scanner->output;
    out->tile_counts_buffer = op->scanner->input;
    out->stats_buffer = op->stats;
    out->num_tiles_x = op->tile_info.num_tiles_x;
    out->num_tiles_y = op->tile_info.num_tiles_y;
    out->total_tiles = op->tile_info.total_tiles;
    out->max_tile_entries = op->tile_info.max_tile_entries;
    std::memcpy(out->settings, &op->settings, sizeof(float) * 7);
    return WDGS_OK;
}

int wdgs_tiled_forward_check(wdgs_tiled_forward* op, uint32_t* stats_out) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "null op");
    WDGS_REQUIRE(!op->dev->capturing, WDGS_E_STATE, "wdgs_tiled_forward_check while recording a command buffer");
    WDGS_CHECK_HIP(wdgs_sync_lanes(op->dev));
    u32 st[4];
    for (int i = 0; i < 4; i++) st[i] = ((const volatile u32*)op->host_stats)[i];
    st[2] = __atomic_exchange_n(op->host_stats + 2, 0u, __ATOMIC_ACQ_REL);  // the overflow word is sticky (set by any encode since the last check): consumed here, atomically
    if (stats_out) std::memcpy(stats_out, st, sizeof(st));
    WDGS_REQUIRE(st[2] == 0u, WDGS_E_CAPACITY, "tile entries overflow: %u entries needed, max_tile_entries = %u (forward pass %p)", st[2], op->tile_info.max_tile_entries, (void*)op);
    return WDGS_OK;
}

int wdgs_tiled_forward_set_long_lists(wdgs_tiled_forward* op, uint32_t threshold, uint32_t max_items, uint32_t max_rows) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "wdgs_tiled_forward_set_long_lists: null op");
    wdgs_device* d = op->dev;
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_tiled_forward_set_long_lists while recording a command buffer");
    WDGS_REQUIRE(!op->cfg.compat_caps || threshold == 0u, WDGS_E_STATE, "wdgs_tiled_forward_set_long_lists: a pass with compat_caps truncates its lists as the reference does");
    WDGS_CHECK_HIP(hipSetDevice(d->ordinal));
    WDGS_CHECK_HIP(wdgs_sync_lanes(d));
    const LongWork& now = op->long_lists;
    return forward_alloc_long_lists(op, threshold, max_items ? max_items : (now.hdr ? now.max_items : LONG_LIST_ITEMS), max_rows ? max_rows : (now.hdr ? now.max_rows : LONG_LIST_ROWS));
}
int wdgs_tiled_forward_long_list_stats(wdgs_tiled_forward* op, uint32_t stats_out[12]) {
    WDGS_REQUIRE(op && stats_out, WDGS_E_INVALID, "wdgs_tiled_forward_long_list_stats: null argument");
    std::memset(stats_out, 0, sizeof(uint32_t) * 12);
    const LongWork& lw = op->long_lists;
    if (!lw.hdr) return WDGS_OK;
    WDGS_TRY(wdgs_copy_to_host(op->dev, stats_out, lw.hdr, sizeof(u32) * LL_HDR_WORDS));
    stats_out[8] = lw.max_items; stats_out[9] = lw.max_blocks; stats_out[10] = lw.max_rows; stats_out[11] = lw.threshold;
    if (std::getenv("WDGS_LL_DEBUG")) {   // (dev aid: the block records of the frame)
        const u32 nb = std::min(stats_out[0], lw.max_blocks);
        std::vector<LongBlock> b(nb);
        std::vector<LongSync> sy(nb);
        if (nb) { WDGS_TRY(wdgs_copy_to_host(op->dev, b.data(), lw.blocks, sizeof(LongBlock) * nb)); WDGS_TRY(wdgs_copy_to_host(op->dev, sy.data(), lw.sync, sizeof(LongSync) * nb)); }
        for (u32 i = 0; i < nb; i++)
            std::fprintf(stderr, "[long lists] block %u: tile %u sub %u first_item %u chunks %u | counted %u row_base %u scanned %u filled %u rows %u walked %u\n", i, b[i].tile, b[i].sub,
                         b[i].first_item, b[i].chunks, sy[i].counted, sy[i].row_base, sy[i].scanned, sy[i].filled, sy[i].rows, sy[i].walked);
    }
    return WDGS_OK;
}

// ---------------------------------------------------------------- TiledRasterizer
int wdgs_tiled_rasterizer_create(wdgs_device* d, wdgs_tiled_forward* fwd, uint32_t compat_caps, wdgs_tiled_rasterizer** out) {
    WDGS_REQUIRE(d && fwd && out, WDGS_E_INVALID, "wdgs_tiled_rasterizer_create: null argument");
    wdgs_tiled_rasterizer* op = new wdgs_tiled_rasterizer();
    std::memset(op, 0, sizeof(*op));
    op->dev = d;
    op->fwd = fwd;
    op->compat_caps = compat_caps;
    *out = op;
    return WDGS_OK;
}
int wdgs_tiled_rasterizer_destroy(wdgs_tiled_rasterizer* op) {
    if (!op) return WDGS_OK;
    sync_if_alive(op->dev);
    free_dev(op->ranges);
    free_dev(op->rgba8);
    free_dev(op->alpha);
    free_dev(op->n_contrib);
    delete op;
    return WDGS_OK;
}
int wdgs_tiled_rasterizer_encode(wdgs_tiled_rasterizer* op, uint32_t width, uint32_t height) {
    WDGS_REQUIRE(op && width > 0 && height > 0, WDGS_E_INVALID, "wdgs_tiled_rasterizer_encode: invalid argument");
    wdgs_tiled_forward* f = op->fwd;
    WDGS_REQUIRE(f->encoded, WDGS_E_STATE, "rasterizer encode before the forward pass was encoded");
    WDGS_REQUIRE(width == f->cfg.viewport_width && height == f->cfg.viewport_height, WDGS_E_STATE,
                 "rasterizer size %ux%u differs from the forward pass viewport %ux%u", width, height, f->cfg.viewport_width, f->cfg.viewport_height);
    wdgs_device* d = op->dev;
    const TileInfo& ti = f->tile_info;
    if (width != op->width || height != op->height) {  // ensureTextures (tiled-rasterizer.ts:244-306)
        WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "TiledRasterizer.encode allocates its textures on first use: run one eager encode before recording");
        (void)wdgs_sync_lanes(d);
        free_dev(op->rgba8); free_dev(op->alpha); free_dev(op->n_contrib);
        op->rgba8 = nullptr; op->alpha = nullptr; op->n_contrib = nullptr;
        const size_t px = (size_t)width * height;
        WDGS_TRY(wdgs_alloc((void**)&op->rgba8, px * 4, true, d->stream));
        WDGS_TRY(wdgs_alloc((void**)&op->alpha, px * 4, true, d->stream));
        WDGS_TRY(wdgs_alloc((void**)&op->n_contrib, px * 4, true, d->stream));
        op->width = width;
        op->height = height;
    }
    const int fo = wdgs_sorter_final_out_index(f->sorter);
    const void* keys = wdgs_sorter_keys(f->sorter, fo);
    const void* vals = wdgs_sorter_values(f->sorter, fo);
    if (f->ranges_valid) {
        op->ranges_used = f->ranges;  // built by the forward pass's sort
    } else {
        if (ti.total_tiles + 1 > op->ranges_capacity) {
            WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "TiledRasterizer.encode allocates its range table on first use: run one eager encode before recording");
            (void)wdgs_sync_lanes(d);
            free_dev(op->ranges);
            op->ranges = nullptr;
            WDGS_TRY(wdgs_alloc((void**)&op->ranges, sizeof(u32) * (size_t)(ti.total_tiles + 1), true, d->stream));
            op->ranges_capacity = ti.total_tiles + 1;
        }
        WDGS_TRY(launch_tile_ranges(d, keys, f->stats, ti.total_tiles, op->ranges));
        op->ranges_used = op->ranges;
    }
    WDGS_TRY(launch_rasterize(d, f->settings, ti, f->splats, f->cfg.num_points, op->ranges_used, keys, vals, f->stats, op->compat_caps ? 32u : 0u, op->rgba8,
                              op->alpha, op->n_contrib, f->nf_stamp, f->stats + FRAME_WORD,
                              (f->ranges_valid && f->long_lists.hdr && !op->compat_caps) ? &f->long_lists : nullptr));   // (the table this pass's sort built and marked)
    op->encoded = true;
    return WDGS_OK;
}
#define RASTER_GETTER(fn, field, what)                                                                            \
    int fn(wdgs_tiled_rasterizer* op, void** out) {                                                               \
        WDGS_REQUIRE(op && out, WDGS_E_INVALID, #fn ": null argument");                                           \
        WDGS_REQUIRE(op->encoded && op->field, WDGS_E_STATE, "TiledRasterizer: " what " not created yet (call encode first)"); \
        *out = op->field;                                                                                         \
        return WDGS_OK;                                                                                           \
    }
RASTER_GETTER(wdgs_tiled_rasterizer_get_output, rgba8, "output texture")
RASTER_GETTER(wdgs_tiled_rasterizer_get_alpha, alpha, "alpha texture")
RASTER_GETTER(wdgs_tiled_rasterizer_get_n_contrib, n_contrib, "n_contrib texture")
RASTER_GETTER(wdgs_tiled_rasterizer_get_tile_offsets, ranges_used, "tile offsets")
#undef RASTER_GETTER

int wdgs_tiled_rasterizer_blit(wdgs_tiled_rasterizer* op, void* target, uint32_t tw, uint32_t th) {
    WDGS_REQUIRE(op && target, WDGS_E_INVALID, "wdgs_tiled_rasterizer_blit: null argument");
    WDGS_REQUIRE(op->encoded && op->rgba8, WDGS_E_STATE, "TiledRasterizer: no output texture to blit from. Call encode() first.");
    WDGS_REQUIRE(tw > 0 && th > 0, WDGS_E_INVALID, "wdgs_tiled_rasterizer_blit: empty target %ux%u", tw, th);
    return launch_downsample(op->dev, op->rgba8, op->width, op->height, target, tw, th);
}

// ---------------------------------------------------------------- TiledBackwardPass
static int backward_alloc_images(wdgs_tiled_backward* op, u32 w, u32 h) {
    const u32 px = w * h;
    if (px <= op->img_capacity) return WDGS_OK;
    (void)wdgs_sync_lanes(op->dev);
    free_dev(op->loss_image); free_dev(op->metric_err); free_dev(op->metric_flags);
    op->loss_image = nullptr; op->metric_err = nullptr; op->metric_flags = nullptr;
    WDGS_TRY(wdgs_alloc((void**)&op->loss_image, (size_t)px * 16, true, op->dev->stream));
    WDGS_TRY(wdgs_alloc((void**)&op->metric_err, (size_t)px * 4, true, op->dev->stream));
    WDGS_TRY(wdgs_alloc((void**)&op->metric_flags, (size_t)px * 4, true, op->dev->stream));
    op->img_capacity = px;
    return WDGS_OK;
}

static int backward_alloc_per_point(wdgs_tiled_backward* op, u32 capacity) {
    wdgs_device* d = op->dev;
    int r = wdgs_alloc((void**)&op->acc, (size_t)capacity * 48 + 16, true, d->stream);
    if (r == WDGS_OK) op->acc_dirty = reinterpret_cast<u32*>(reinterpret_cast<char*>(op->acc) + (size_t)capacity * 48);
    if (r == WDGS_OK) r = wdgs_alloc((void**)&op->gradients, (size_t)capacity * 32, true, d->stream);
    if (r == WDGS_OK) r = wdgs_alloc((void**)&op->metric_counts, (size_t)capacity * 4, true, d->stream);
    if (r == WDGS_OK) op->points_capacity = capacity;
    return r;
}
// Counterpart of wdgs_tiled_forward_resize for the backward pass's per-Gaussian buffers (accumulators, packed gradients, metric counts).
int wdgs_tiled_backward_resize(wdgs_tiled_backward* op, uint32_t n) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "wdgs_tiled_backward_resize: null op");
    wdgs_device* d = op->dev;
    WDGS_REQUIRE(!d->capturing, WDGS_E_STATE, "wdgs_tiled_backward_resize while recording a command buffer");
    WDGS_CHECK_HIP(hipSetDevice(d->ordinal));
    WDGS_CHECK_HIP(wdgs_sync_lanes(d));
    const u32 need = std::max(n, 1u);
    if (need > op->points_capacity) {
        free_dev(op->acc); free_dev(op->gradients); free_dev(op->metric_counts);
        op->acc = nullptr; op->gradients = nullptr; op->metric_counts = nullptr;
        op->points_capacity = 0;
        WDGS_TRY(backward_alloc_per_point(op, (u32)std::min<uint64_t>((uint64_t)need + need / 4, 0xFFFFFFFFull)));
    } else {
        WDGS_CHECK_HIP(hipMemsetAsync(op->acc, 0, (size_t)need * 48, d->stream));
        WDGS_CHECK_HIP(hipMemsetAsync(op->acc_dirty, 0, 16, d->stream));
        WDGS_CHECK_HIP(hipMemsetAsync(op->gradients, 0, (size_t)need * 32, d->stream));
        WDGS_CHECK_HIP(hipMemsetAsync(op->metric_counts, 0, (size_t)need * 4, d->stream));
    }
    op->cfg.num_points = n;
    return WDGS_OK;
}
int wdgs_tiled_backward_create(wdgs_device* d, const wdgs_tiled_backward_config* cfg, wdgs_tiled_backward** out) {
    WDGS_REQUIRE(d && cfg && out, WDGS_E_INVALID, "wdgs_tiled_backward_create: null argument");
    WDGS_REQUIRE(cfg->viewport_width > 0 && cfg->viewport_height > 0, WDGS_E_INVALID, "viewport must be non-empty");
    wdgs_tiled_backward* op = new wdgs_tiled_backward();
    std::memset(op, 0, sizeof(*op));
    op->dev = d;
    op->cfg = *cfg;
    op->gradient_output = true;
    if (op->cfg.training.c1 == 0.f) op->cfg.training.c1 = 0.01f * 0.01f;  // tiled-backward-pass.ts:172-173
    if (op->cfg.training.c2 == 0.f) op->cfg.training.c2 = 0.03f * 0.03f;
    op->settings = RenderSettings{cfg->gaussian_scale != 0.f ? cfg->gaussian_scale : 1.0f, (float)cfg->sh_deg, (float)cfg->viewport_width,
                                  (float)cfg->viewport_height, cfg->point_size_px != 0.f ? cfg->point_size_px : 3.0f, 0.0f,
                                  cfg->max_splat_radius_px != 0.f ? cfg->max_splat_radius_px : 128.0f};
    int r = backward_alloc_per_point(op, std::max(cfg->num_points, 1u));
    if (r == WDGS_OK) r = wdgs_alloc((void**)&op->metric_minmax, 4096, true, d->stream);
    if (r == WDGS_OK) r = backward_alloc_images(op, cfg->viewport_width, cfg->viewport_height);
    if (r != WDGS_OK) { wdgs_tiled_backward_destroy(op); return r; }
    *out = op;
    return WDGS_OK;
}
int wdgs_tiled_backward_destroy(wdgs_tiled_backward* op) {
    if (!op) return WDGS_OK;
    sync_if_alive(op->dev);
    free_dev(op->acc); free_dev(op->gradients); free_dev(op->loss_image); free_dev(op->metric_counts);
    free_dev(op->metric_err); free_dev(op->metric_flags); free_dev(op->metric_minmax);
    delete op;
    return WDGS_OK;
}
int wdgs_tiled_backward_set_gradient_output(wdgs_tiled_backward* op, int enabled) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "wdgs_tiled_backward_set_gradient_output: null op");
    op->gradient_output = enabled != 0;
    return WDGS_OK;
}
int wdgs_tiled_backward_compute_loss_only(wdgs_tiled_backward* op, const void* pred, const void* targ) {
    WDGS_REQUIRE(op && pred && targ, WDGS_E_INVALID, "wdgs_tiled_backward_compute_loss_only: null argument");
    return launch_loss_grad(op->dev, op->cfg.viewport_width, op->cfg.viewport_height, pred, targ, op->cfg.training, op->loss_image, nullptr, 0u, nullptr);
}
static int backward_encode_raster(wdgs_tiled_backward* op, const void* pred, const void* targ, const wdgs_tiled_backward_resources* res) {
    wdgs_device* d = op->dev;
    const u32 w = op->cfg.viewport_width, h = op->cfg.viewport_height, n = op->cfg.num_points;
    // K15 + clearBuffer x4 (tiled-backward-pass.ts:624-627): the clear rides on the loss kernel and is a no-op behind a consuming K17
    if (w > 0 && h > 0) WDGS_TRY(launch_loss_grad(d, w, h, pred, targ, op->cfg.training, op->loss_image, op->acc, std::max(n, 1u), op->acc_dirty));
    else WDGS_TRY(launch_acc_clear_if_dirty(d, op->acc, n, op->acc_dirty));
    // long tile lists (longlist.h): when the range table is one a forward pass of this device built, that pass's lists serve the backward walk too
    const LongWork* lw = nullptr;
    auto owner = d->range_tables.find(res->tile_offsets_buffer);
    if (owner != d->range_tables.end() && owner->second->ranges_valid && owner->second->long_lists.hdr) lw = &owner->second->long_lists;
    return launch_backward_rasterize(d, op->settings, ceil_div(w, 16), ceil_div(h, 16), res->tile_offsets_buffer, res->tile_indices_buffer, res->splat_buffer,
                                     res->alpha_texture, res->n_contrib_texture, op->loss_image, op->acc, op->acc_dirty, lw);
}
int wdgs_tiled_backward_encode(wdgs_tiled_backward* op, const void* pred, const void* targ, const wdgs_tiled_backward_resources* res,
                               const void* gaussians) {
    WDGS_REQUIRE(op && pred && targ && res && gaussians, WDGS_E_INVALID, "wdgs_tiled_backward_encode: null argument");
    WDGS_REQUIRE(res->splat_buffer && res->tile_offsets_buffer && res->tile_indices_buffer && res->camera_buffer && res->alpha_texture && res->n_contrib_texture,
                 WDGS_E_INVALID, "wdgs_tiled_backward_encode: incomplete resources");
    WDGS_TRY(backward_encode_raster(op, pred, targ, res));
    return launch_geometry_backward(op->dev, op->cfg.num_points, res->camera_buffer, op->settings, gaussians, op->acc, op->gradients);
}
// The two halves of wdgs_tiled_backward_encode, for a batched step: K15 + clear + K16 may run (and be recorded) on any lane at any
// time; K17 of view k also adds the view's gradient to the step's fp32 block, which has to follow view k-1's K17.
int wdgs_tiled_backward_encode_raster(wdgs_tiled_backward* op, const void* pred, const void* targ, const wdgs_tiled_backward_resources* res) {
    WDGS_REQUIRE(op && pred && targ && res, WDGS_E_INVALID, "wdgs_tiled_backward_encode_raster: null argument");
    WDGS_REQUIRE(res->splat_buffer && res->tile_offsets_buffer && res->tile_indices_buffer && res->alpha_texture && res->n_contrib_texture, WDGS_E_INVALID,
                 "wdgs_tiled_backward_encode_raster: incomplete resources");
    return backward_encode_raster(op, pred, targ, res);
}
int wdgs_tiled_backward_encode_geometry(wdgs_tiled_backward* op, const void* camera, const void* gaussians, const wdgs_view_accumulate* into) {
    WDGS_REQUIRE(op && camera && gaussians, WDGS_E_INVALID, "wdgs_tiled_backward_encode_geometry: null argument");
    if (!into) return launch_geometry_backward(op->dev, op->cfg.num_points, camera, op->settings, gaussians, op->acc, op->gradients);
    WDGS_REQUIRE(into->sums && into->visible && into->tile_counts && into->guard && into->overflow_word, WDGS_E_INVALID,
                 "wdgs_tiled_backward_encode_geometry: incomplete accumulate target");
    return launch_geometry_backward_accumulate(op->dev, op->cfg.num_points, camera, op->settings, gaussians, op->acc, op->acc_dirty, op->gradients, into->sums, into->visible,
                                               into->tile_counts, into->guard, into->overflow_word, into->first ? 1u : 2u);
}
// ---- view-batched K17 (include/webdgs.h)
int wdgs_tiled_backward_encode_geometry_views(wdgs_tiled_backward* const* ops, const void* const* cameras, const void* const* tile_counts, const void* const* overflow_words,
                                              uint32_t count, const void* gaussians, void* sums, void* visible, void* guard, int write_gradients, int continues) {
    WDGS_REQUIRE(ops && cameras && tile_counts && overflow_words && gaussians && sums && visible && guard && count > 0 && count <= WDGS_MAX_BATCH_VIEWS, WDGS_E_INVALID,
                 "wdgs_tiled_backward_encode_geometry_views: invalid argument");
    wdgs_tiled_backward* b0 = ops[0];
    WDGS_REQUIRE(b0, WDGS_E_INVALID, "wdgs_tiled_backward_encode_geometry_views: null pass");
    void *accs[WDGS_MAX_BATCH_VIEWS], *dirty[WDGS_MAX_BATCH_VIEWS], *grads[WDGS_MAX_BATCH_VIEWS];
    for (u32 v = 0; v < count; v++) {
        wdgs_tiled_backward* b = ops[v];
        WDGS_REQUIRE(b && cameras[v] && tile_counts[v] && overflow_words[v], WDGS_E_INVALID, "wdgs_tiled_backward_encode_geometry_views: null argument for view %u", v);
        for (u32 u = 0; u < v; u++) WDGS_REQUIRE(ops[u] != b, WDGS_E_INVALID, "wdgs_tiled_backward_encode_geometry_views: pass %u is given twice (every view has accumulators of its own)", v);
        WDGS_REQUIRE(b->dev == b0->dev && b->cfg.num_points == b0->cfg.num_points && std::memcmp(&b->settings, &b0->settings, sizeof(RenderSettings)) == 0, WDGS_E_STATE,
                     "wdgs_tiled_backward_encode_geometry_views: pass %u differs from pass 0 (cloud size, viewport or settings)", v);
        accs[v] = b->acc; dirty[v] = b->acc_dirty; grads[v] = b->gradients;
    }
    return launch_geometry_backward_views(b0->dev, b0->cfg.num_points, count, cameras, b0->settings, gaussians, accs, dirty, tile_counts, overflow_words,
                                          write_gradients ? grads : nullptr, sums, visible, guard, continues ? 1u : 0u);
}
int wdgs_tiled_backward_compute_metric_map(wdgs_tiled_backward* op, const void* pred, const void* targ, float threshold) {
    WDGS_REQUIRE(op && pred && targ, WDGS_E_INVALID, "wdgs_tiled_backward_compute_metric_map: null argument");
    return launch_metric_map(op->dev, op->cfg.viewport_width, op->cfg.viewport_height, pred, targ, 1000000.0f, threshold, op->metric_err, op->metric_minmax,
                             op->metric_minmax + 2, op->metric_flags);
}
int wdgs_tiled_backward_compute_metric_counts(wdgs_tiled_backward* op, const wdgs_tiled_backward_resources* res, uint32_t num_instances, int clear) {
    WDGS_REQUIRE(op && res && res->splat_buffer && res->tile_offsets_buffer && res->tile_indices_buffer && res->n_contrib_texture, WDGS_E_INVALID,
                 "wdgs_tiled_backward_compute_metric_counts: incomplete resources");
    const u32 n = op->cfg.num_points;
    u32* const counts = op->metric_counts_into ? op->metric_counts_into : op->metric_counts;
    if (clear) WDGS_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)std::max(n, 1u) * 4, op->dev->stream));
    return launch_metric_count(op->dev, op->settings, ceil_div(op->cfg.viewport_width, 16), ceil_div(op->cfg.viewport_height, 16), res->tile_offsets_buffer,
                               res->tile_indices_buffer, num_instances, res->splat_buffer, n, op->metric_flags, res->n_contrib_texture, counts, n);
}
int wdgs_tiled_backward_normalize_metric_counts(wdgs_tiled_backward* op, uint32_t divisor) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "null op");
    return launch_metric_normalize(op->dev, op->cfg.num_points, divisor, op->metric_counts);
}
int wdgs_tiled_backward_set_viewport(wdgs_tiled_backward* op, uint32_t w, uint32_t h) {
    WDGS_REQUIRE(op && w > 0 && h > 0, WDGS_E_INVALID, "wdgs_tiled_backward_set_viewport: invalid argument");
    op->cfg.viewport_width = w;
    op->cfg.viewport_height = h;
    op->settings.viewport_x = (float)w;
    op->settings.viewport_y = (float)h;
    return backward_alloc_images(op, w, h);
}
int wdgs_tiled_backward_set_training_config(wdgs_tiled_backward* op, const wdgs_training_config* cfg) {
    WDGS_REQUIRE(op && cfg, WDGS_E_INVALID, "null argument");
    op->cfg.training = *cfg;
    if (op->cfg.training.c1 == 0.f) op->cfg.training.c1 = 0.01f * 0.01f;
    if (op->cfg.training.c2 == 0.f) op->cfg.training.c2 = 0.03f * 0.03f;
    return WDGS_OK;
}
void* wdgs_tiled_backward_gradients(wdgs_tiled_backward* op) { return op ? op->gradients : nullptr; }
void* wdgs_tiled_backward_metric_counts(wdgs_tiled_backward* op) { return op ? op->metric_counts : nullptr; }
int wdgs_tiled_backward_set_metric_counts_target(wdgs_tiled_backward* op, void* counts_dev) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "wdgs_tiled_backward_set_metric_counts_target: null op");
    op->metric_counts_into = (u32*)counts_dev;
    return WDGS_OK;
}
void* wdgs_tiled_backward_loss_image(wdgs_tiled_backward* op) { return op ? op->loss_image : nullptr; }
void* wdgs_tiled_backward_metric_map(wdgs_tiled_backward* op) { return op ? op->metric_flags : nullptr; }
void* wdgs_tiled_backward_accumulators(wdgs_tiled_backward* op) { return op ? op->acc : nullptr; }
void* wdgs_tiled_backward_metric_minmax(wdgs_tiled_backward* op) { return op ? op->metric_minmax : nullptr; }

int wdgs_downsample_rgba8(wdgs_device* d, const void* src, uint32_t sw, uint32_t sh, void* dst, uint32_t dw, uint32_t dh) {
    WDGS_REQUIRE(d && src && dst && sw && sh && dw && dh, WDGS_E_INVALID, "wdgs_downsample_rgba8: invalid argument");
    return launch_downsample(d, src, sw, sh, dst, dw, dh);
}

// ---------------------------------------------------------------- Optimizer
int wdgs_optimizer_state_sizes(uint32_t n, size_t sizes[6]) {
    WDGS_REQUIRE(sizes, WDGS_E_INVALID, "null argument");
    const size_t N = std::max(n, 1u);  // allocateOptimizerStateBuffers: Math.max(1, ...) (optimizer.ts:28)
    sizes[0] = sizes[1] = sizes[2] = N * 48;
    sizes[3] = N * 12;
    sizes[4] = N * 48 * 4;
    sizes[5] = N * 48 * 2 * 4;
    return WDGS_OK;
}

static void optimizer_free_state(wdgs_optimizer* op) {
    free_dev(op->state.opt_pos); free_dev(op->state.opt_rot); free_dev(op->state.opt_scale);
    free_dev(op->state.opt_opacity); free_dev(op->state.param_sh); free_dev(op->state.state_sh);
    std::memset(&op->state, 0, sizeof(op->state));
}

int wdgs_optimizer_create(wdgs_device* d, uint32_t n, const wdgs_adam_hyperparameters* params, const void* gaussians, const void* sh,
                          const wdgs_optimizer_state* initial, int owns_state, uint32_t initial_iteration, wdgs_optimizer** out) {
    WDGS_REQUIRE(d && out, WDGS_E_INVALID, "wdgs_optimizer_create: null argument");
    wdgs_optimizer* op = new wdgs_optimizer();
    std::memset(op, 0, sizeof(*op));
    op->dev = d;
    op->num_points = n;
    const wdgs_adam_hyperparameters defaults = {0.00016f, 0.0025f, 0.05f, 0.005f, 0.001f, 0.9f, 0.999f, 1e-8f};  // adam-config.ts:12-21
    op->params = params ? *params : defaults;
    if (initial) {
        WDGS_REQUIRE(initial->opt_pos && initial->opt_rot && initial->opt_scale && initial->opt_opacity && initial->param_sh && initial->state_sh,
                     WDGS_E_INVALID, "wdgs_optimizer_create: incomplete initial state");
        op->state = *initial;
        op->owns_state = owns_state != 0;
        op->iteration = initial_iteration;
    } else {
        if (!(gaussians && sh)) { delete op; wdgs_set_error("wdgs_optimizer_create: point cloud required when no initial state is given"); return WDGS_E_INVALID; }
        size_t sz[6];
        wdgs_optimizer_state_sizes(n, sz);
        void** fields[6] = {&op->state.opt_pos, &op->state.opt_rot, &op->state.opt_scale, &op->state.opt_opacity, &op->state.param_sh, &op->state.state_sh};
        op->owns_state = true;
        for (int i = 0; i < 6; i++) {
            int r = wdgs_alloc(fields[i], sz[i], true, d->stream);
            if (r != WDGS_OK) { optimizer_free_state(op); delete op; return r; }
        }
        int r = launch_unpack(d, n, gaussians, sh, op->state);
        if (r != WDGS_OK) { optimizer_free_state(op); delete op; return r; }
        op->iteration = 0;
    }
    int r = wdgs_alloc((void**)&op->dc, sizeof(float4) * CS_PLANES * (size_t)wdgs_optimizer::cs_pitch(n), true, d->stream);
    if (r == WDGS_OK) r = launch_cs_load(d, n, op->state, op->cs());
    if (r != WDGS_OK) { free_dev(op->dc); if (op->owns_state) optimizer_free_state(op); delete op; return r; }
    op->dc_dirty = false;
    *out = op;
    return WDGS_OK;
}
int wdgs_optimizer_destroy(wdgs_optimizer* op) {
    if (!op) return WDGS_OK;
    const bool live = wdgs_device_alive(op->dev) && !op->dev->capturing;
    if (!op->owns_state && live) (void)optimizer_flush_dc(op);  // adopted buffers outlive the optimizer: leave them current
    sync_if_alive(op->dev);
    if (op->owns_state) optimizer_free_state(op);
    free_dev(op->dc);
    free_dev(op->dc_words);
    delete op;
    return WDGS_OK;
}
int wdgs_optimizer_init_from_point_cloud(wdgs_optimizer* op, const void* gaussians, const void* sh) {
    WDGS_REQUIRE(op && gaussians && sh, WDGS_E_INVALID, "wdgs_optimizer_init_from_point_cloud: null argument");
    WDGS_TRY(launch_unpack(op->dev, op->num_points, gaussians, sh, op->state));
    op->dc_dirty = false;
    return launch_cs_load(op->dev, op->num_points, op->state, op->cs());
}
int wdgs_optimizer_step(wdgs_optimizer* op, void* gaussians, void* sh, const void* gradients, const void* tile_counts) {
    WDGS_REQUIRE(op && gaussians && sh && gradients && tile_counts, WDGS_E_INVALID, "wdgs_optimizer_step: null argument");
    op->iteration++;  // optimizer.ts:301
    op->dc_dirty = true;
    if (op->deferred_sh) op->sh_stale = true;
    return launch_adam_repack(op->dev, op->num_points, op->params, tile_counts, gradients, op->state, op->cs(), gaussians, sh, op->guard, op->deferred_sh ? op->dc_words : nullptr);
}
// K17 + K18 + K19 in one pass over the Gaussians (the single-view step): `bwd` must have run wdgs_tiled_backward_encode_raster for this view.
int wdgs_optimizer_step_with_geometry(wdgs_optimizer* op, wdgs_tiled_backward* bwd, const void* camera, void* gaussians, void* sh, const void* tile_counts) {
    WDGS_REQUIRE(op && bwd && camera && gaussians && sh && tile_counts, WDGS_E_INVALID, "wdgs_optimizer_step_with_geometry: null argument");
    WDGS_REQUIRE(bwd->cfg.num_points == op->num_points, WDGS_E_STATE, "wdgs_optimizer_step_with_geometry: the backward pass holds %u Gaussians, the optimizer %u",
                 bwd->cfg.num_points, op->num_points);
    op->iteration++;  // optimizer.ts:301
    op->dc_dirty = true;
    if (op->deferred_sh) op->sh_stale = true;
    return launch_geometry_backward_adam(op->dev, op->num_points, camera, bwd->settings, gaussians, bwd->acc, bwd->acc_dirty, bwd->gradient_output ? bwd->gradients : nullptr,
                                         op->params, tile_counts, op->state,
                                         op->cs(), sh, op->guard, op->deferred_sh ? op->dc_words : nullptr);
}
int wdgs_optimizer_step_f32(wdgs_optimizer* op, void* gaussians, void* sh, const void* grad_f32, const void* visible) {
    WDGS_REQUIRE(op && gaussians && sh && grad_f32 && visible, WDGS_E_INVALID, "wdgs_optimizer_step_f32: null argument");
    op->iteration++;
    op->dc_dirty = true;
    if (op->deferred_sh) op->sh_stale = true;
    return launch_adam_repack_f32(op->dev, 0, op->num_points, op->params, visible, grad_f32, op->state, op->cs(), gaussians, sh, op->guard, op->dev->host_guard, nullptr,
                                  op->deferred_sh ? op->dc_words : nullptr);
}
int wdgs_optimizer_step_f32_range(wdgs_optimizer* op, void* gaussians, void* sh, const void* grad_f32, const void* visible, uint32_t first, uint32_t count,
                                  void* rows_out) {
    WDGS_REQUIRE(op && gaussians && sh && grad_f32 && visible, WDGS_E_INVALID, "wdgs_optimizer_step_f32_range: null argument");
    WDGS_REQUIRE((uint64_t)first + count <= op->num_points, WDGS_E_INVALID, "wdgs_optimizer_step_f32_range: [%u, %u) exceeds %u points", first, first + count,
                 op->num_points);
    op->iteration++;
    op->dc_dirty = true;
    if (op->deferred_sh) op->sh_stale = true;
    return launch_adam_repack_f32(op->dev, first, count, op->params, visible, grad_f32, op->state, op->cs(), gaussians, sh, op->guard, op->dev->host_guard, rows_out,
                                  op->deferred_sh ? op->dc_words : nullptr);
}
int wdgs_optimizer_set_guard(wdgs_optimizer* op, const void* flag) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "null op");
    op->guard = flag;
    return WDGS_OK;
}
int wdgs_optimizer_state_changed(wdgs_optimizer* op) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "null op");
    op->dc_dirty = false;
    return launch_cs_load(op->dev, op->num_points, op->state, op->cs());
}
int wdgs_apply_repacked_rows(wdgs_device* d, uint32_t n, const void* rows, uint32_t skip_first, uint32_t skip_count, const void* guard, void* gaussians,
                             void* sh) {
    WDGS_REQUIRE(d && rows && gaussians && sh, WDGS_E_INVALID, "wdgs_apply_repacked_rows: null argument");
    return launch_apply_rows(d, n, rows, skip_first, skip_count, guard, d->host_guard, gaussians, sh, nullptr);
}
// The same for a replica whose optimizer defers its SH writes: the gathered DC halves go to its compact words as well.
int wdgs_optimizer_apply_repacked_rows(wdgs_optimizer* op, const void* rows, uint32_t skip_first, uint32_t skip_count, const void* guard, void* gaussians, void* sh) {
    WDGS_REQUIRE(op && rows && gaussians && sh, WDGS_E_INVALID, "wdgs_optimizer_apply_repacked_rows: null argument");
    if (op->deferred_sh) op->sh_stale = true;
    return launch_apply_rows(op->dev, op->num_points, rows, skip_first, skip_count, guard, op->dev->host_guard, gaussians, sh, op->deferred_sh ? op->dc_words : nullptr);
}
// ---- deferred SH writes (no reference counterpart)
int wdgs_optimizer_set_deferred_sh(wdgs_optimizer* op, void* sh, int enabled) {
    WDGS_REQUIRE(op && sh, WDGS_E_INVALID, "wdgs_optimizer_set_deferred_sh: null argument");
    WDGS_REQUIRE(!op->dev->capturing, WDGS_E_STATE, "wdgs_optimizer_set_deferred_sh while recording a command buffer");
    if (!enabled) {
        if (op->deferred_sh && op->sh_stale) WDGS_TRY(launch_dc_words_flush(op->dev, op->num_points, op->dc_words, sh));
        op->deferred_sh = false;
        op->sh_stale = false;
        return WDGS_OK;
    }
    if (!op->dc_words) WDGS_TRY(wdgs_alloc((void**)&op->dc_words, sizeof(u32) * 2 * (size_t)std::max(op->num_points, 1u), true, op->dev->stream));
    if (!op->deferred_sh || !op->sh_stale) WDGS_TRY(launch_dc_words_load(op->dev, op->num_points, sh, op->dc_words));  // the rows are current: take their DC halves
    op->deferred_sh = true;
    return WDGS_OK;
}
void* wdgs_optimizer_dc_words(wdgs_optimizer* op) { return (op && op->deferred_sh) ? op->dc_words : nullptr; }
int wdgs_optimizer_flush_sh(wdgs_optimizer* op, void* sh) {
    WDGS_REQUIRE(op && sh, WDGS_E_INVALID, "wdgs_optimizer_flush_sh: null argument");
    if (!op->deferred_sh || !op->sh_stale) return WDGS_OK;
    WDGS_REQUIRE(!op->dev->capturing, WDGS_E_STATE, "wdgs_optimizer_flush_sh while recording a command buffer");
    WDGS_TRY(launch_dc_words_flush(op->dev, op->num_points, op->dc_words, sh));  // stream-ordered: later kernels and copies on this device see current rows
    op->sh_stale = false;
    return WDGS_OK;
}
int wdgs_tiled_forward_set_dc_source(wdgs_tiled_forward* op, const void* dc_words) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "wdgs_tiled_forward_set_dc_source: null op");
    op->dc_source = dc_words;
    return WDGS_OK;
}
int wdgs_guard_accumulate(wdgs_device* d, void* flag, const void* src, int overwrite) {
    WDGS_REQUIRE(d && flag && src, WDGS_E_INVALID, "wdgs_guard_accumulate: null argument");
    return launch_guard_accumulate(d, flag, src, overwrite ? 1u : 0u);
}
int wdgs_accumulate_gradients(wdgs_device* d, uint32_t n, const void* gradients, const void* tile_counts, void* acc, void* visible) {
    WDGS_REQUIRE(d && gradients && tile_counts && acc && visible, WDGS_E_INVALID, "wdgs_accumulate_gradients: null argument");
    return launch_accumulate_gradients(d, n, gradients, tile_counts, acc, visible);
}
int wdgs_store_gradients(wdgs_device* d, uint32_t n, const void* gradients, const void* tile_counts, void* acc, void* visible) {
    WDGS_REQUIRE(d && gradients && tile_counts && acc && visible, WDGS_E_INVALID, "wdgs_store_gradients: null argument");
    return launch_store_gradients(d, n, gradients, tile_counts, acc, visible);
}
uint32_t wdgs_optimizer_get_iteration(const wdgs_optimizer* op) { return op ? op->iteration : 0; }
int wdgs_optimizer_advance_iteration(wdgs_optimizer* op, uint32_t count) {
    WDGS_REQUIRE(op, WDGS_E_INVALID, "null op");
    op->iteration += count;
    if (count) op->dc_dirty = true;  // a recorded step() was re-submitted
    if (count && op->deferred_sh) op->sh_stale = true;
    return WDGS_OK;
}
int wdgs_optimizer_get_hyperparameters(const wdgs_optimizer* op, wdgs_adam_hyperparameters* out) {
    WDGS_REQUIRE(op && out, WDGS_E_INVALID, "null argument");
    *out = op->params;
    return WDGS_OK;
}
int wdgs_optimizer_set_hyperparameters(wdgs_optimizer* op, const wdgs_adam_hyperparameters* p) {
    WDGS_REQUIRE(op && p, WDGS_E_INVALID, "null argument");
    op->params = *p;
    return WDGS_OK;
}
int wdgs_optimizer_get_state(wdgs_optimizer* op, wdgs_optimizer_state* out) {
    WDGS_REQUIRE(op && out, WDGS_E_INVALID, "null argument");
    WDGS_TRY(optimizer_flush_dc(op));  // stream-ordered: later kernels and copies on this device see current SH rows
    *out = op->state;
    return WDGS_OK;
}
int wdgs_optimizer_release_state(wdgs_optimizer* op, wdgs_optimizer_state* out) {
    WDGS_REQUIRE(op && out, WDGS_E_INVALID, "null argument");
    WDGS_TRY(optimizer_flush_dc(op));
    *out = op->state;
    op->owns_state = false;
    return WDGS_OK;
}

}  // extern "C"
