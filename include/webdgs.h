/*
 * webdgs.h -- C ABI of libwebdgs_hip.so: the MI355X (gfx950) hot path of krispy-kenay/WebDGS.
 *
 * The reference has no FFI: its boundary is the set of TypeScript operator classes that
 * src/trainer.ts and src/viewer.ts construct.  Every entry point below names the
 * reference interface it replaces (paths relative to the reference's src/).  Conventions:
 *
 *   - every function returns int: 0 = WDGS_OK, <0 = WDGS_E_*; wdgs_last_error() gives the text
 *     (the reference throws `Error`, e.g. renderers/tiled-rasterizer.ts:308-330);
 *   - `void*` arguments named *_dev are DEVICE pointers (hipMalloc / torch / wdgs_buffer_ptr); no torch or
 *     HIP types appear in signatures;
 *   - ops own the buffers they create and free them in *_destroy (idempotent on NULL), as the reference's
 *     destroy() methods do (renderers/tiled-forward-pass.ts:518-533); borrowed pointers are never freed;
 *   - "encode" = the reference's encode(encoder, ...): work is queued on the device's HIP stream in call
 *     order and runs asynchronously; wdgs_device_synchronize() = queue.onSubmittedWorkDone()
 *     (trainer.ts:639-645).  wdgs_buffer_write() is stream-ordered (queue.writeBuffer, camera/camera.ts:194);
 *   - sizes that the reference derives on the GPU (total tile entries) stay on the GPU: no host read-back
 *     inside a step.  Capacity overflow is a hard error reported by *_check / wdgs_device_synchronize
 *     (the reference silently overruns: SURVEY Q1, Q2);
 *   - handles are not thread-safe; use one host thread per wdgs_device;
 *   - teardown order: destroy command buffers and ops, then wdgs_comm, then the device.  An op destroyed AFTER its device
 *     (a host finalising objects in arbitrary order at exit) only releases its memory and never touches the dead device;
 *     wdgs_device_destroy is idempotent.
 *
 * Data layouts (byte-exact with the reference unless marked INTERNAL):
 *   Gaussian        6 x u32 = 12 fp16: x y z opacity_raw | rot w x y z | log-sigma x y z, pad   shaders/common.wgsl:20-24
 *   SH              24 x u32 = 48 fp16, [k][rgb], 16 coefficient slots                          shaders/tiled-forward.wgsl:64-86
 *   Splat           6 x u32 fp16 pairs: ndc.xy | extent.xy px | conic.xy | conic.z,0 | r,g | b,opacity   shaders/common.wgsl:26-33
 *   CameraUniforms  68 f32: view, view_inv, proj, proj_inv (column-major), viewport.xy, focal.xy  shaders/common.wgsl:1-8
 *   GaussianGradient 8 x u32 = 16 fp16: dpos.xyz dopacity | drot.wxyz | dlogsigma.xyz 0 | drgb 0  shaders/tiled-backward.wgsl:18-23
 *   OptVec4 {param,m,v: vec4f} 48 B; OptFloat {param,m,v} 12 B; param_sh 48 f32; state_sh 48 x (m,v)   renderers/optimizer.ts:7-11
 *   images          rgba8unorm, row-major, W*H*4 bytes (textures in the reference)
 *   grad accumulators INTERNAL: i32 x 12 per Gaussian {mean.x, mean.y, conic.x, conic.y, conic.z, opacity, r, g, b as four partial sums}
 *                   (the reference keeps four arrays, renderers/tiled-backward-pass.ts:249-252; same fixed-point x1e6 values)
 */
#ifndef WEBDGS_H
#define WEBDGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WDGS_OK 0
#define WDGS_E_INVALID (-1)  /* bad argument / API misuse */
#define WDGS_E_HIP (-2)      /* HIP runtime error */
#define WDGS_E_CAPACITY (-3) /* a device-side capacity was exceeded (tile entries, densify output) */
#define WDGS_E_STATE (-4)    /* getter before first encode, mismatched sizes, ... */

typedef struct wdgs_device wdgs_device;
typedef struct wdgs_buffer wdgs_buffer;
typedef struct wdgs_prefix_scanner wdgs_prefix_scanner;
typedef struct wdgs_sorter wdgs_sorter;
typedef struct wdgs_tiled_forward wdgs_tiled_forward;
typedef struct wdgs_tiled_rasterizer wdgs_tiled_rasterizer;
typedef struct wdgs_tiled_backward wdgs_tiled_backward;
typedef struct wdgs_optimizer wdgs_optimizer;
typedef struct wdgs_densify_prune wdgs_densify_prune;
typedef struct wdgs_command_buffer wdgs_command_buffer;

const char* wdgs_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int wdgs_abi_version(void);

/* ---------------------------------------------------------------- device / queue
 * Replaces GPUDevice + GPUQueue (main.ts:186-201).  external_hip_stream may be NULL (the device creates its own
 * stream) or a hipStream_t the caller owns (e.g. torch.cuda.current_stream().cuda_stream). */
int wdgs_device_create(int hip_ordinal, void* external_hip_stream, wdgs_device** out);
int wdgs_device_destroy(wdgs_device* dev);
/* queue.onSubmittedWorkDone(): waits for the stream, then reports deferred device-side errors. */
int wdgs_device_synchronize(wdgs_device* dev);
/* device.limits (trainer.ts:147 reads maxStorageBufferBindingSize to bound the densify rebuild): the device's memory in bytes.  free_bytes counts what
 * the library's allocation cache holds (cached_bytes: freed blocks kept for the next allocation of their size class) as used.  Any pointer may be NULL. */
int wdgs_device_memory_info(wdgs_device* dev, size_t* free_bytes, size_t* total_bytes, size_t* cached_bytes);
/* Per-kernel hipEvent timing (the reference only has a wall-clock meter, trainer.ts:570,647-651). */
int wdgs_device_set_profiling(wdgs_device* dev, int enabled);
/* After a synchronize: copies up to `cap` records; returns the number of distinct kernels through *count. */
typedef struct wdgs_kernel_time {
    char name[48];
    uint32_t launches;
    float total_ms;
} wdgs_kernel_time;
int wdgs_device_get_kernel_times(wdgs_device* dev, wdgs_kernel_time* out, uint32_t cap, uint32_t* count);
int wdgs_device_reset_kernel_times(wdgs_device* dev);
/* Lanes.  A device has WDGS_MAX_LANES in-order queues: lane 0 is its stream (the one given to wdgs_device_create), the others are
 * internal and created on first use.  wdgs_device_select_lane makes `lane` the target of every later encode / submit / copy; work
 * on different lanes may run concurrently and is ordered only where wdgs_device_lane_order says so: everything submitted to lane
 * `waiter` after the call runs after everything submitted to lane `signal` before it (an event, no host wait).
 * wdgs_device_synchronize waits for all of them.  The reference has one GPUQueue and no counterpart; a batched step
 * (views_per_rank > 1, BASELINE config c4) uses the lanes to run the bandwidth-bound stages of one view beside the rasterization
 * kernels of another.  Each lane needs its own forward / rasterizer / backward ops (their intermediate buffers are per op); the
 * point cloud is shared read-only until the lanes are joined in front of the optimizer step.  Neither call is allowed while
 * recording. */
#define WDGS_MAX_LANES 4
#define WDGS_MAX_BATCH_VIEWS 16 /* views one view-batched K1 / K17 launch covers (wdgs_tiled_forward_project_views) */
int wdgs_device_select_lane(wdgs_device* dev, int lane);
int wdgs_device_lane_order(wdgs_device* dev, int waiter_lane, int signal_lane);
/* A position on `lane`, remembered in one of WDGS_MAX_BATCH_VIEWS numbered marks, for other lanes to wait for later
 * (wdgs_device_lane_order records and waits in one call). */
int wdgs_device_lane_mark(wdgs_device* dev, int lane, int mark);
int wdgs_device_lane_wait_mark(wdgs_device* dev, int lane, int mark);

/* ---------------------------------------------------------------- recorded command buffers (hipGraph)
 * Replaces device.createCommandEncoder() ... encoder.finish() -> GPUCommandBuffer -> queue.submit([cmd]) (trainer.ts:603-645).
 * Between begin and end every encode call is captured into a HIP graph instead of running; wdgs_queue_submit replays it on
 * the device's stream.  Unlike a GPUCommandBuffer the result may be submitted any number of times, as long as the buffers it
 * was recorded against are alive (all sizes the kernels need are read on the device, so a replay adapts to new data).
 * Calls that synchronise or allocate (copy_to_host, *_check, first-use allocations inside encode) are not allowed while
 * recording: run one eager step first.  Per-kernel profiling is suspended inside a recording. */
int wdgs_encoder_begin(wdgs_device* dev);
int wdgs_encoder_finish(wdgs_device* dev, wdgs_command_buffer** out);
/* Drops an open recording (an encode between begin and finish failed, or the host threw): ends the capture, discards the partial
 * graph and returns the device to eager mode.  A no-op when nothing is being recorded, so error paths may call it unconditionally.
 * The reference has no counterpart: an encoder that is never finished is simply garbage-collected. */
int wdgs_encoder_abort(wdgs_device* dev);
int wdgs_queue_submit(wdgs_device* dev, wdgs_command_buffer* cmd);
int wdgs_command_buffer_destroy(wdgs_command_buffer* cmd);
/* queue.onSubmittedWorkDone() as a completion callback (trainer.ts:639-645): `fn(user)` runs on a runtime thread once everything
 * submitted to the device's stream before this call has finished.  It must not call back into this library or HIP; an N-API
 * host resolves its Promise from it through a thread-safe function.  Deferred device-side checks (tile-entry overflow) are
 * still reported by wdgs_device_synchronize / wdgs_tiled_forward_check. */
typedef void (*wdgs_done_callback)(void* user);
int wdgs_queue_on_done(wdgs_device* dev, wdgs_done_callback fn, void* user);
/* The same promise, kept and awaited later: wdgs_queue_mark returns a ticket for "everything submitted to the current lane so far",
 * wdgs_queue_wait blocks the host until that work has finished and then reports the deferred device-side errors exactly as
 * wdgs_device_synchronize does.  A host that waits for step k-1's ticket after submitting step k (`const p = queue.onSubmittedWorkDone();
 * ...; await previous`) keeps the device busy across the step boundary; trainer.ts:639-645 awaits at once, which is mark + wait.
 * At most WDGS_TICKET_RING tickets stay distinct: an older one waits for the mark that took its slot (later in the same queue). */
#define WDGS_TICKET_RING 8
int wdgs_queue_mark(wdgs_device* dev, uint64_t* ticket);
int wdgs_queue_wait(wdgs_device* dev, uint64_t ticket);

/* Raw device<->host copies on the device's stream (copy_to_host synchronises): mapAsync/getMappedRange
 * (trainer.ts:455-458) and queue.writeBuffer. */
int wdgs_copy_to_host(wdgs_device* dev, void* dst_host, const void* src_dev, size_t bytes);
int wdgs_copy_to_device(wdgs_device* dev, void* dst_dev, const void* src_host, size_t bytes);
int wdgs_memset(wdgs_device* dev, void* dst_dev, int value, size_t bytes); /* encoder.clearBuffer */
/* encoder.copyBufferToBuffer(src, srcOffset, dst, dstOffset, size) (the staging copy of trainer.ts:445): device to device on the
 * current lane, stream-ordered; recordable.  The ranges must not overlap. */
int wdgs_copy_buffer_to_buffer(wdgs_device* dev, void* dst_dev, const void* src_dev, size_t bytes);

/* ---------------------------------------------------------------- buffers (for hosts without their own allocator)
 * Replaces device.createBuffer / GPUBuffer.destroy; contents are zero-filled like WebGPU buffers. */
int wdgs_buffer_create(wdgs_device* dev, size_t bytes, wdgs_buffer** out);
int wdgs_buffer_destroy(wdgs_buffer* buf);
void* wdgs_buffer_ptr(const wdgs_buffer* buf);
size_t wdgs_buffer_size(const wdgs_buffer* buf);
int wdgs_buffer_write(wdgs_device* dev, wdgs_buffer* buf, size_t offset, const void* src_host, size_t bytes);
int wdgs_buffer_read(wdgs_device* dev, const wdgs_buffer* buf, size_t offset, void* dst_host, size_t bytes);
/* mapAsync(READ) counterpart (trainer.ts:455-458): queues the copy on the device's stream and returns; dst_host is valid after
 * a later wdgs_queue_on_done callback or wdgs_device_synchronize.  dst_host should come from wdgs_host_alloc (pinned) for the
 * copy to overlap with the host. */
int wdgs_buffer_read_async(wdgs_device* dev, const wdgs_buffer* buf, size_t offset, void* dst_host, size_t bytes);
int wdgs_host_alloc(size_t bytes, void** out_host); /* pinned host memory */
int wdgs_host_free(void* host);

/* ---------------------------------------------------------------- prefix scanner
 * Replaces get_prefix_scanner(maxElements, device): PrefixScanner (prefix/prefix.ts:140, interface 26-43).
 * Exclusive u32 scan.  No 2 097 152-element cap. */
int wdgs_prefix_scanner_create(wdgs_device* dev, uint32_t max_elements, wdgs_prefix_scanner** out);
int wdgs_prefix_scanner_destroy(wdgs_prefix_scanner* s);
void* wdgs_prefix_scanner_input(wdgs_prefix_scanner* s);  /* input_buffer  (u32[max_elements]) */
void* wdgs_prefix_scanner_output(wdgs_prefix_scanner* s); /* output_buffer (u32[max_elements]) */
int wdgs_prefix_scanner_set_count(wdgs_prefix_scanner* s, uint32_t count);
int wdgs_prefix_scanner_scan(wdgs_prefix_scanner* s);
/* Scan arbitrary device arrays with this scanner's scratch (count <= max_elements). */
int wdgs_prefix_scanner_scan_ptr(wdgs_prefix_scanner* s, const void* in_dev, void* out_dev, uint32_t count);

/* ---------------------------------------------------------------- dynamic sorter
 * Replaces get_dynamic_sorter(maxCapacity, device, statsBuffer): DynamicSortStuff (sort/sort_dynamic.ts:252,
 * interface 9-24).  Stable ascending LSD radix sort of (key u32, value u32) pairs; the element count is read on
 * the device from stats_dev[0] (TilePipelineStats.total_tile_entries).  Results land in
 * ping_pong[final_out_index] (sort_dynamic.ts:24, 386).  key_bits limits the passes to the significant digits
 * (the reference always runs 4). */
int wdgs_sorter_create(wdgs_device* dev, uint32_t max_capacity, const void* stats_dev, wdgs_sorter** out);
int wdgs_sorter_destroy(wdgs_sorter* s);
void* wdgs_sorter_keys(wdgs_sorter* s, int ping_pong_index);   /* sort_depths_buffer  */
void* wdgs_sorter_values(wdgs_sorter* s, int ping_pong_index); /* sort_indices_buffer */
int wdgs_sorter_sort(wdgs_sorter* s, uint32_t key_bits);
int wdgs_sorter_final_out_index(wdgs_sorter* s);  /* DynamicSortStuff.final_out_index */
uint32_t wdgs_sorter_capacity(wdgs_sorter* s);

/* ---------------------------------------------------------------- TiledForwardPass
 * Replaces `new TiledForwardPass(device, pointCloud, cameraBuffer, config)` (renderers/tiled-forward-pass.ts:120-125,
 * config 24-31), .encode (341-387), setters (389-425), .getResources (428-443), getters (445-459), .destroy (518). */
typedef struct wdgs_tiled_forward_config {
    uint32_t num_points;         /* pointCloud.num_points */
    uint32_t sh_deg;             /* pointCloud.sh_deg */
    uint32_t viewport_width;
    uint32_t viewport_height;
    float gaussian_scale;        /* default 1.0 (declared, unused by the tiled path: SURVEY A3) */
    float point_size_px;         /* default 3.0 */
    float max_splat_radius_px;   /* default 128.0 */
    uint32_t render_mode;        /* 1 = 'gaussian', 0 = 'pointcloud' */
    uint32_t max_tile_entries;   /* 0 = auto: max(30*N, 1<<20) rounded up to 4096 */
    uint32_t compat_caps;        /* 1 = reproduce the reference's capacity caps (min(30N, 32Mi, 2097152) -> x3840) */
} wdgs_tiled_forward_config;

typedef struct wdgs_tiled_forward_resources { /* TiledForwardResources (tiled-forward-pass.ts:33-46) */
    void* splat_buffer;        /* Splat[num_points] */
    void* depths_buffer;       /* u32[num_points] ordered-uint view z */
    void* tile_keys_buffer;    /* u32[max_tile_entries]  (sorted after encode) */
    void* tile_indices_buffer; /* u32[max_tile_entries]  (sorted after encode) */
    void* tile_offsets_buffer; /* u32[num_points] per-GAUSSIAN exclusive scan */
    void* tile_counts_buffer;  /* u32[num_points] */
    void* stats_buffer;        /* u32[4] {total_tile_entries, visible_gaussians, overflow_flag, pad} */
    uint32_t num_tiles_x, num_tiles_y, total_tiles, max_tile_entries;
    float settings[7];         /* RenderSettings (shaders/common.wgsl:10-18) */
} wdgs_tiled_forward_resources;

int wdgs_tiled_forward_create(wdgs_device* dev, const wdgs_tiled_forward_config* cfg, wdgs_tiled_forward** out);
int wdgs_tiled_forward_destroy(wdgs_tiled_forward* op);
/* The point cloud changed size (densify / prune).  applyPointCloudSwap (trainer.ts:201-237) destroys every pass and builds new ones; this
 * keeps the pass: its buffers are reused when large enough and re-allocated with 25 % headroom when not, and the pass is left in the
 * state of a freshly created one for num_points (zeroed per-Gaussian buffers, nothing encoded, max_tile_entries by the creation
 * formula).  Synchronises; not allowed while recording; command buffers recorded against the pass must be dropped by the caller. */
int wdgs_tiled_forward_resize(wdgs_tiled_forward* op, uint32_t num_points);
/* encode(encoder, {skipSort}): K1 project+count, scan, stats, K6 emit, sort. gaussians/sh/camera are device pointers. */
int wdgs_tiled_forward_encode(wdgs_tiled_forward* op, const void* gaussians_dev, const void* sh_dev, const void* camera_dev, int skip_sort);
int wdgs_tiled_forward_set_viewport(wdgs_tiled_forward* op, uint32_t width, uint32_t height);
int wdgs_tiled_forward_set_render_mode(wdgs_tiled_forward* op, uint32_t render_mode);
int wdgs_tiled_forward_set_point_size(wdgs_tiled_forward* op, float point_size_px);
int wdgs_tiled_forward_set_gaussian_scale(wdgs_tiled_forward* op, float scale);
int wdgs_tiled_forward_get_resources(wdgs_tiled_forward* op, wdgs_tiled_forward_resources* out);
/* Synchronises and returns WDGS_E_CAPACITY if the last encode overflowed max_tile_entries; stats_out (4 x u32) optional. */
int wdgs_tiled_forward_check(wdgs_tiled_forward* op, uint32_t* stats_out);
/* Long tile lists.  The reference stages at most 32 x 256 = 8 192 entries of a tile (tiled-rasterizer.wgsl:59-60, 125) and drops the rest; this
 * library composites every entry, and gives the blocks of a tile with more than `threshold` entries per-PIXEL lists, built and walked by tasks that the
 * waves of the rasterization kernels work off themselves (csrc/longlist.h) -- no launch, no host decision: the path is part of every recording and taken
 * on the device.  Results do not depend on it.  A pass is created with threshold 2048, room for 1024 (block, 64-entry chunk) slots and 8192 list rows (13 MB);
 * the path is taken by ALL long tiles of a frame or by none: a frame whose long tiles want more slots than there are is composited the ordinary way
 * (correct; slow if its long tiles are few, right if they are many: a frame full of long tiles keeps the chip busy without it).  set_long_lists re-sizes the work (threshold 0: off; items / rows 0: keep);
 * synchronises, not allowed while recording, command buffers recorded against the pass must be dropped.  long_list_stats (synchronises) returns
 * {block records wanted, item slots wanted, forward queue position, backward queue position, rows handed out, rows wanted, stall code (0 = none), 0,
 *  max_items, max_blocks, max_rows, threshold} of the last frame: a host compares wanted with the capacities and enlarges them -- up to a cap of its
 * choosing (the Trainers: 8 192 slots, 65 536 rows). */
int wdgs_tiled_forward_set_long_lists(wdgs_tiled_forward* op, uint32_t threshold, uint32_t max_items, uint32_t max_rows);
int wdgs_tiled_forward_long_list_stats(wdgs_tiled_forward* op, uint32_t stats_out[12]);

/* ---------------------------------------------------------------- TiledRasterizer
 * Replaces `new TiledRasterizer({device, forwardPass, format})` (renderers/tiled-rasterizer.ts:57), .encode(encoder,w,h)
 * (180-242: tile ranges K12-K13 + composite K14), texture getters (308-330), .destroy (359).  The tile grid follows
 * the forward pass's CURRENT viewport (fixes the stale-grid host bug). */
int wdgs_tiled_rasterizer_create(wdgs_device* dev, wdgs_tiled_forward* forward, uint32_t compat_caps, wdgs_tiled_rasterizer** out);
int wdgs_tiled_rasterizer_destroy(wdgs_tiled_rasterizer* op);
int wdgs_tiled_rasterizer_encode(wdgs_tiled_rasterizer* op, uint32_t width, uint32_t height);
/* Getters fail with WDGS_E_STATE before the first encode, as the reference throws. */
int wdgs_tiled_rasterizer_get_output(wdgs_tiled_rasterizer* op, void** rgba8_dev);        /* getOutputTextureView   */
int wdgs_tiled_rasterizer_get_alpha(wdgs_tiled_rasterizer* op, void** final_t_dev);       /* getAlphaTextureView  f32[W*H] */
int wdgs_tiled_rasterizer_get_n_contrib(wdgs_tiled_rasterizer* op, void** n_contrib_dev); /* getNContribTextureView u32[W*H] */
int wdgs_tiled_rasterizer_get_tile_offsets(wdgs_tiled_rasterizer* op, void** ranges_dev); /* getTileOffsetsBuffer: per-TILE table u32[T+1] */
/* blitToTexture (tiled-rasterizer.ts:333-357, shaders/blit.wgsl vs_main/fs_main): full-target draw that samples the output
 * image with a linear clamp-to-edge sampler into an rgba8 target of any size (the viewer's swap-chain image, viewer.ts:76-86).
 * WDGS_E_STATE before the first encode, as the reference throws. */
int wdgs_tiled_rasterizer_blit(wdgs_tiled_rasterizer* op, void* target_rgba8_dev, uint32_t target_width, uint32_t target_height);

/* ---------------------------------------------------------------- TiledBackwardPass
 * Replaces `new TiledBackwardPass(device, pointCloud, config)` (renderers/tiled-backward-pass.ts:136-140, config 27-34,
 * TrainingConfig 19-25), .encode (592-740), .computeLossOnly (383), .computeMetricMap (425), .computeMetricCounts (514),
 * .normalizeMetricCounts (565), getters (409-421, 832), .setViewport (742), .setTrainingConfig (812), .destroy (836). */
typedef struct wdgs_training_config {
    float lambda_l1, lambda_l2, lambda_dssim, c1, c2; /* defaults 0.8, 0.0, 0.2, 1e-4, 9e-4 (trainer.ts:100-104) */
} wdgs_training_config;
typedef struct wdgs_tiled_backward_config {
    uint32_t num_points;
    uint32_t sh_deg;
    uint32_t viewport_width, viewport_height;
    wdgs_training_config training;
    float gaussian_scale, point_size_px, max_splat_radius_px;
} wdgs_tiled_backward_config;
typedef struct wdgs_tiled_backward_resources { /* TiledBackwardResources (tiled-backward-pass.ts:40-50) */
    const void* splat_buffer;
    const void* tile_offsets_buffer; /* the RASTERIZER's per-tile table (trainer.ts:621) */
    const void* tile_indices_buffer; /* sorted indices */
    const void* camera_buffer;
    const void* alpha_texture;       /* f32[W*H] final T */
    const void* n_contrib_texture;   /* u32[W*H] */
} wdgs_tiled_backward_resources;

int wdgs_tiled_backward_create(wdgs_device* dev, const wdgs_tiled_backward_config* cfg, wdgs_tiled_backward** out);
/* The point cloud changed size: see wdgs_tiled_forward_resize. */
int wdgs_tiled_backward_resize(wdgs_tiled_backward* op, uint32_t num_points);
int wdgs_tiled_backward_destroy(wdgs_tiled_backward* op);
/* encode: K15 loss gradient, clear accumulators, K16 backward raster, K17 geometry backward -> GaussianGradient[N]. */
int wdgs_tiled_backward_encode(wdgs_tiled_backward* op, const void* predicted_rgba8_dev, const void* target_rgba8_dev,
                               const wdgs_tiled_backward_resources* res, const void* gaussians_dev);
/* The two halves of encode, for a batched step (views_per_rank > 1; no counterpart in the reference, whose step has one view):
 * encode_raster = K15 + clear + K16, encode_geometry = K17.  With `into`, K17 also adds the view's gradient -- the fp16-rounded values
 * it has just packed -- to the step's fp32 block (f32[N][14] + visibility counts, `first` stores instead of adding) and folds the
 * forward pass's overflow word into the step's guard word: what wdgs_store_gradients / wdgs_accumulate_gradients +
 * wdgs_guard_accumulate do in two more launches.  The sums into the block must follow the previous view's (wdgs_device_lane_order). */
typedef struct wdgs_view_accumulate {
    void* sums;                /* f32[N][14] */
    void* visible;             /* u32[N] */
    const void* tile_counts;   /* forward pass: u32[N] */
    void* guard;               /* u32: the step's guard word */
    const void* overflow_word; /* u32: the forward pass's overflow word (stats_buffer + 8) */
    int first;                 /* first view of the step: store, do not add */
} wdgs_view_accumulate;
int wdgs_tiled_backward_encode_raster(wdgs_tiled_backward* op, const void* predicted_rgba8_dev, const void* target_rgba8_dev,
                                      const wdgs_tiled_backward_resources* res);
int wdgs_tiled_backward_encode_geometry(wdgs_tiled_backward* op, const void* camera_dev, const void* gaussians_dev, const wdgs_view_accumulate* into);
int wdgs_tiled_backward_compute_loss_only(wdgs_tiled_backward* op, const void* predicted_rgba8_dev, const void* target_rgba8_dev);
int wdgs_tiled_backward_compute_metric_map(wdgs_tiled_backward* op, const void* predicted_rgba8_dev, const void* target_rgba8_dev, float threshold);
int wdgs_tiled_backward_compute_metric_counts(wdgs_tiled_backward* op, const wdgs_tiled_backward_resources* res, uint32_t num_instances, int clear);
int wdgs_tiled_backward_normalize_metric_counts(wdgs_tiled_backward* op, uint32_t divisor);
int wdgs_tiled_backward_set_viewport(wdgs_tiled_backward* op, uint32_t width, uint32_t height);
int wdgs_tiled_backward_set_training_config(wdgs_tiled_backward* op, const wdgs_training_config* cfg);
void* wdgs_tiled_backward_gradients(wdgs_tiled_backward* op);     /* getGradientsBuffer: GaussianGradient[N] */
void* wdgs_tiled_backward_metric_counts(wdgs_tiled_backward* op); /* getMetricCountsBuffer: u32[N] */
/* computeMetricCounts of this pass adds into `counts_dev` (u32[num_points], e.g. another pass's getMetricCountsBuffer) instead of its own
 * array; NULL restores its own.  No reference counterpart: runDensifyPruneMultiView (trainer.ts:373-497) walks its metric views one after
 * the other through ONE pass; with a shared target several passes can take the views of one densify event on different lanes -- the
 * counts are integer atomics, so every order gives the same bits.  The target must outlive the calls and hold at least num_points words. */
int wdgs_tiled_backward_set_metric_counts_target(wdgs_tiled_backward* op, void* counts_dev);
void* wdgs_tiled_backward_loss_image(wdgs_tiled_backward* op);    /* getLossTextureView: rgba32float W*H */
void* wdgs_tiled_backward_metric_map(wdgs_tiled_backward* op);    /* getMetricMapTextureView: u32[W*H] */
/* Whether the fused step (wdgs_optimizer_step_with_geometry) also writes K17's packed GaussianGradient[N] to the pass's gradient buffer
 * (default 1, as tiled-backward-pass.ts:629-640 does).  A host that never reads getGradientsBuffer() -- trainer.ts hands it to
 * optimizer.step() only, which the fused step replaces -- saves the 32 bytes per Gaussian.  The separate K17 always writes it. */
int wdgs_tiled_backward_set_gradient_output(wdgs_tiled_backward* op, int enabled);
void* wdgs_tiled_backward_accumulators(wdgs_tiled_backward* op);  /* INTERNAL i32[N*12] (for parity tests) */
void* wdgs_tiled_backward_metric_minmax(wdgs_tiled_backward* op); /* u32[2] global (min,max) of the last metric map */
/* Bilinear down-sample of an rgba8 image (the blit render pass of trainer.ts:303-328, shaders/blit.wgsl fs_main). */
int wdgs_downsample_rgba8(wdgs_device* dev, const void* src_dev, uint32_t src_w, uint32_t src_h, void* dst_dev, uint32_t dst_w, uint32_t dst_h);

/* Exact integer sum of squared differences over the R,G,B bytes of two rgba8 images -> *out_u64_dev (u64, device).
 * The reference has no scalar loss (it only visualises the gradient image, trainer.ts:695-768); PSNR = 10 log10(255^2 * 3P / SSE). */
int wdgs_image_sse_rgb8(wdgs_device* dev, const void* a_rgba8_dev, const void* b_rgba8_dev, uint32_t num_pixels, void* out_u64_dev);

/* Test hook, not part of the reference's surface: evaluates one pinned arithmetic primitive of the kernels elementwise over
 * `count` 32-bit patterns (0 exp, 1 log, 2 f32->f16 bits, 3 f16 bits->f32, 4 saturating f32->i32, 5 saturating f32->u32,
 * 6 sqrt, 7 1/x), so that a parity suite can compare them with its oracle directly. */
int wdgs_debug_eval_math(wdgs_device* dev, uint32_t which, uint32_t count, const void* in_u32_dev, void* out_u32_dev);

/* ---------------------------------------------------------------- Optimizer
 * Replaces allocateOptimizerStateBuffers (renderers/optimizer.ts:27-38), `new Optimizer(device, pointCloud, params?,
 * initialState?)` (71-88), .step (295-350), hyperparameter accessors (256-278), .destroy (352). */
typedef struct wdgs_adam_hyperparameters { /* AdamHyperparameters (renderers/adam-config.ts:1-21) */
    float lr_pos, lr_color, lr_opacity, lr_scale, lr_rot, beta1, beta2, epsilon;
} wdgs_adam_hyperparameters;
typedef struct wdgs_optimizer_state { /* OptimizerStateBuffers (optimizer.ts:13-20); device pointers */
    void* opt_pos;     /* OptVec4[N]   48 B */
    void* opt_rot;     /* OptVec4[N]   48 B */
    void* opt_scale;   /* OptVec4[N]   48 B */
    void* opt_opacity; /* OptFloat[N]  12 B */
    void* param_sh;    /* f32[N*48] */
    void* state_sh;    /* (m,v)[N*48] */
} wdgs_optimizer_state;
/* Byte sizes of the six state arrays for n points, in the struct's field order. */
int wdgs_optimizer_state_sizes(uint32_t num_points, size_t sizes_out[6]);
/* initial_state == NULL: the optimizer allocates zeroed state and initialises the fp32 masters from the fp16 point
 * cloud (K20, optimizer.ts:166-253).  Otherwise the given buffers are ADOPTED as-is (optimizer.ts:81-88) and `owns_state`
 * says whether destroy frees them. */
int wdgs_optimizer_create(wdgs_device* dev, uint32_t num_points, const wdgs_adam_hyperparameters* params,
                          const void* gaussians_dev, const void* sh_dev,
                          const wdgs_optimizer_state* initial_state, int owns_state, uint32_t initial_iteration, wdgs_optimizer** out);
int wdgs_optimizer_destroy(wdgs_optimizer* op);
/* K20 (optimizer.ts:145-253 initBuffers): fp16 point cloud -> fp32 master parameters of this optimizer's state; m, v untouched. */
int wdgs_optimizer_init_from_point_cloud(wdgs_optimizer* op, const void* gaussians_dev, const void* sh_dev);
/* step(encoder, pointCloud, gradientsBuffer, tileCountsBuffer): iteration++, K18 Adam, K19 re-pack into gaussians/sh. */
int wdgs_optimizer_step(wdgs_optimizer* op, void* gaussians_dev, void* sh_dev, const void* gradients_dev, const void* tile_counts_dev);
/* The same step fused with K17: after wdgs_tiled_backward_encode_raster(bwd, ...) for the view, one pass over the Gaussians computes the
 * geometry backward (the packed gradient is still written to bwd's gradient buffer), then Adam and the re-pack of the same Gaussian from
 * the fp16-rounded values in registers -- optimizer.step() of trainer.ts:635 without a second pass over N and a read-back of the gradient. */
int wdgs_optimizer_step_with_geometry(wdgs_optimizer* op, wdgs_tiled_backward* bwd, const void* camera_dev, void* gaussians_dev, void* sh_dev,
                                      const void* tile_counts_dev);
/* Data-parallel variant: gradients are fp32 sums over views, 14 f32 per Gaussian in GaussianGradient
 * component order {pos3, opacity, rot4, logsigma3, rgb3}, plus u32 visibility counts (Adam runs where count > 0). */
int wdgs_optimizer_step_f32(wdgs_optimizer* op, void* gaussians_dev, void* sh_dev, const void* grad_f32_dev, const void* visible_counts_dev);
/* acc_f32[N*14] += unpack(GaussianGradient[N]) where tile_counts > 0; visible[N] += (tile_counts > 0). */
int wdgs_accumulate_gradients(wdgs_device* dev, uint32_t num_points, const void* gradients_dev, const void* tile_counts_dev,
                              void* acc_f32_dev, void* visible_counts_dev);
/* Overwrite form for the first view of a batch: acc_f32[N*14] = unpack(GaussianGradient[N]) (zeros where tile_counts == 0),
 * visible[N] = (tile_counts > 0).  Spares the batch a clearing pass over the 60 B/Gaussian block. */
int wdgs_store_gradients(wdgs_device* dev, uint32_t num_points, const void* gradients_dev, const void* tile_counts_dev,
                         void* acc_f32_dev, void* visible_counts_dev);
/* Adam + re-pack on Gaussians [first, first + count) only: the slice a data-parallel rank owns after
 * wdgs_comm_exchange_gradients.  rows_out_dev (nullable, u32[num_points rounded up][8]) also receives each re-packed row in the
 * 32-byte form {6 Gaussian words, SH word 0, low half of SH word 1} that wdgs_comm_allgather_rows publishes. */
int wdgs_optimizer_step_f32_range(wdgs_optimizer* op, void* gaussians_dev, void* sh_dev, const void* grad_f32_dev, const void* visible_counts_dev,
                                  uint32_t first, uint32_t count, void* rows_out_dev);
/* Writes the rows published by the other ranks into this replica's point cloud: every Gaussian outside [skip_first, skip_first +
 * skip_count).  guard_dev (nullable): non-zero at execution time -> no-op. */
int wdgs_apply_repacked_rows(wdgs_device* dev, uint32_t num_points, const void* rows_dev, uint32_t skip_first, uint32_t skip_count,
                             const void* guard_dev, void* gaussians_dev, void* sh_dev);
/* Guard word: while *flag_u32_dev != 0 at EXECUTION time, step / step_f32 / step_f32_range leave every buffer untouched.  Point it at
 * TiledForwardResources.stats_buffer + 8 (the overflow word): a step whose tile-entry list was truncated -- it will be reported
 * as WDGS_E_CAPACITY by the next wdgs_device_synchronize -- then does not corrupt the optimizer state first.  NULL removes it. */
int wdgs_optimizer_set_guard(wdgs_optimizer* op, const void* flag_u32_dev);
/* Deferred SH writes (no reference counterpart).  update-gaussians.wgsl:59-75 (K19) rewrites the first six bytes of every 96-byte SH row
 * each step; on HBM a 6-byte store is a read-modify-write of a 64-byte memory word.  With deferral on, the step functions write those
 * three fp16 halves to a compact array owned by the optimizer (u32[N][2]: r,g | b,0) and leave the rows alone; a forward pass given that
 * array (wdgs_tiled_forward_set_dc_source) reads the halves from it, so rendering and training see the same values as before.  The rows
 * are brought up to date by wdgs_optimizer_flush_sh, which the host calls at every hand-over: before the cloud's SH buffer is read by the
 * host, exported, rendered by a forward pass that has no dc source (a viewer), or copied by DensifyPrunePass.encodeScatter.
 * `sh_dev` = the point cloud's SH buffer: set_deferred_sh(enabled) takes the current halves from it, (disabled) flushes into it. */
int wdgs_optimizer_set_deferred_sh(wdgs_optimizer* op, void* sh_dev, int enabled);
void* wdgs_optimizer_dc_words(wdgs_optimizer* op);            /* the compact array, or NULL while deferral is off */
int wdgs_optimizer_flush_sh(wdgs_optimizer* op, void* sh_dev); /* no-op unless deferral is on and a step ran since the last flush */
/* wdgs_apply_repacked_rows for a replica whose optimizer defers its SH writes (the gathered halves go to the compact array) */
int wdgs_optimizer_apply_repacked_rows(wdgs_optimizer* op, const void* rows_dev, uint32_t skip_first, uint32_t skip_count, const void* guard_u32_dev,
                                       void* gaussians_dev, void* sh_dev);
/* project_count (K1) takes the SH-DC halves from `dc_words_dev` (wdgs_optimizer_dc_words) instead of the rows; NULL restores the rows */
int wdgs_tiled_forward_set_dc_source(wdgs_tiled_forward* op, const void* dc_words_dev);

/* View-batched step (no reference counterpart: the reference is batch-1, trainer.ts:573).  A step over V views projects every Gaussian
 * for all V cameras in ONE launch -- the 24-byte Gaussian and its 96-byte SH row are read once, not V times -- into the V forward passes'
 * own buffers (`ops[v]`, all created for the same cloud and viewport); each pass then runs the rest of its encode (scan, emit, sort) with
 * wdgs_tiled_forward_encode_projected, on any lane, recorded or not.  Results are those of wdgs_tiled_forward_encode per view, bit for bit. */
int wdgs_tiled_forward_project_views(wdgs_tiled_forward* const* ops, const void* const* cameras_dev, uint32_t count, const void* gaussians_dev, const void* sh_dev);
int wdgs_tiled_forward_encode_projected(wdgs_tiled_forward* op);
/* 1 while the pass holds a projection that wdgs_tiled_forward_encode_projected has not consumed yet and no wdgs_tiled_forward_encode has
 * overwritten (the scan works in place on K1's workgroup sums: the rest of the pass can run once per projection). */
int wdgs_tiled_forward_is_projected(const wdgs_tiled_forward* op);
/* ... and K17 of all V views in one launch: per view the accumulators of `ops[v]` (wdgs_tiled_backward_encode_raster ran for it) are turned
 * into the view's fp16 gradient under `cameras_dev[v]` and summed, in view order, into the step's fp32 block `sums_f32_dev` [N][14] with the
 * visibility counts `visible_dev` [N] and the guard word (OR of the views' overflow words `overflow_words_dev[v]`) -- what V calls of
 * wdgs_tiled_backward_encode_geometry(into = {first: v == 0}) produce, bit for bit, with the Gaussians read once and the fp32 block written
 * once.  `write_gradients` != 0 also stores each view's packed gradient in its pass's gradient buffer.  A step may hand its views over
 * in several GROUPS (so that one group's K17 runs beside the next group's rasterization): `continues` = 0 for the group that holds the
 * step's first view, 1 for every later group, in view order -- the block then goes on from what the earlier groups left. */
int wdgs_tiled_backward_encode_geometry_views(wdgs_tiled_backward* const* ops, const void* const* cameras_dev, const void* const* tile_counts_dev,
                                              const void* const* overflow_words_dev, uint32_t count, const void* gaussians_dev, void* sums_f32_dev,
                                              void* visible_dev, void* guard_u32_dev, int write_gradients, int continues);
/* flag = (overwrite ? 0 : flag) | (*src != 0): folds the overflow words of the views of a batched step into one guard word. */
int wdgs_guard_accumulate(wdgs_device* dev, void* flag_u32_dev, const void* src_u32_dev, int overwrite);
/* The caller rewrote the state arrays (gathered slices from other ranks): refresh the optimizer's internal compact copies. */
int wdgs_optimizer_state_changed(wdgs_optimizer* op);
uint32_t wdgs_optimizer_get_iteration(const wdgs_optimizer* op);
/* Host-side counter only (optimizer.ts:301): call when a recorded command buffer containing step() is re-submitted. */
int wdgs_optimizer_advance_iteration(wdgs_optimizer* op, uint32_t count);
int wdgs_optimizer_get_hyperparameters(const wdgs_optimizer* op, wdgs_adam_hyperparameters* out);
int wdgs_optimizer_set_hyperparameters(wdgs_optimizer* op, const wdgs_adam_hyperparameters* params);
/* getStateBuffers.  The optimizer trains a compact copy of the SH-DC parameter and moments (36 B per Gaussian instead of
 * three cache lines at 192/384-byte strides); param_sh / state_sh rows 0..2 are brought up to date HERE, in
 * wdgs_optimizer_release_state and in wdgs_optimizer_destroy (adopted state).  Always fetch the state through one of these
 * before reading those two arrays or handing them to wdgs_densify_prune_encode_scatter. */
int wdgs_optimizer_get_state(wdgs_optimizer* op, wdgs_optimizer_state* out);
/* Detaches the state so destroy does not free it (hand-over across a densify swap, trainer.ts:492-495). */
int wdgs_optimizer_release_state(wdgs_optimizer* op, wdgs_optimizer_state* out);

/* ---------------------------------------------------------------- DensifyPrunePass
 * Replaces `new DensifyPrunePass(device, config)` (renderers/densify-prune.ts:108, config 17-30), .setConfig/.ensureSize
 * (279-312), .encodePrepare (458-468: decide K26, scan, cap K27, scan, total K28), .encodeScatter (470-678: K29 + K30). */
typedef struct wdgs_densify_config {
    uint32_t num_views;              /* numViews */
    uint32_t clone_threshold;        /* cloneThreshold (metric count) */
    float split_threshold;           /* splitThreshold (max scale) */
    float prune_threshold;           /* pruneThreshold (opacity) */
    uint32_t max_new_points_per_step;/* 0 = unlimited */
    uint64_t max_buffer_bytes;       /* 128 MiB in the reference; 0 = unlimited */
} wdgs_densify_config;
typedef struct wdgs_densify_prepared { /* DensifyPrunePrepared (densify-prune.ts:42-48) */
    void* action_buffer;     /* u32[N] 0 keep, 1 clone, 2 split, 3 prune */
    void* out_count_buffer;  /* u32[N] */
    void* out_offset_buffer; /* u32[N] exclusive scan */
    void* out_total_buffer;  /* u32[1] */
    uint32_t max_out_points;
} wdgs_densify_prepared;

int wdgs_densify_prune_create(wdgs_device* dev, const wdgs_densify_config* cfg, wdgs_densify_prune** out);
int wdgs_densify_prune_destroy(wdgs_densify_prune* op);
int wdgs_densify_prune_set_config(wdgs_densify_prune* op, const wdgs_densify_config* cfg);
int wdgs_densify_prune_ensure_size(wdgs_densify_prune* op, uint32_t num_points);
int wdgs_densify_prune_encode_prepare(wdgs_densify_prune* op, uint32_t num_points, const void* gaussians_dev,
                                      const void* metric_counts_dev, wdgs_densify_prepared* out);
/* The stages encodePrepare is made of, individually recordable as in the reference: .encodeDecision (densify-prune.ts:412-456,
 * K26), .encodePrefixSum (327-337: exclusive scan of out_count into out_offset), .encodeCapToMax (363-388, K27),
 * .encodeTotalOut (339-361, K28) and computeMaxOutPoints (390-410).  The work buffers are the ones encode_prepare reports. */
int wdgs_densify_prune_encode_decision(wdgs_densify_prune* op, uint32_t num_points, const void* gaussians_dev, const void* metric_counts_dev);
int wdgs_densify_prune_encode_prefix_sum(wdgs_densify_prune* op, uint32_t num_points);
int wdgs_densify_prune_encode_cap_to_max(wdgs_densify_prune* op, uint32_t num_points, uint32_t max_out_points);
int wdgs_densify_prune_encode_total_out(wdgs_densify_prune* op, uint32_t num_points);
int wdgs_densify_prune_compute_max_out_points(wdgs_densify_prune* op, uint32_t num_points, uint32_t* max_out_points);
/* getOutTotalBuffer & co. (densify-prune.ts:314-324): the pass's work buffers; WDGS_E_STATE before they exist. */
int wdgs_densify_prune_get_buffers(wdgs_densify_prune* op, wdgs_densify_prepared* out);
/* Reads the 4-byte total back (the one device->host crossing of the densify path, trainer.ts:440-458). */
int wdgs_densify_prune_read_total(wdgs_densify_prune* op, uint32_t* total_out);
/* encodeScatter: out_num_points must equal the size of the output buffers (the reference throws otherwise,
 * densify-prune.ts:478-480).  in_state/out_state may both be NULL (point cloud only). */
int wdgs_densify_prune_encode_scatter(wdgs_densify_prune* op, uint32_t in_points, const void* in_gaussians_dev, const void* in_sh_dev,
                                      const wdgs_optimizer_state* in_state, uint32_t out_num_points, int reset_new_optimizer_state,
                                      void* out_gaussians_dev, void* out_sh_dev, const wdgs_optimizer_state* out_state);

/* ---------------------------------------------------------------- data-parallel communicator (RCCL over xGMI)
 * No counterpart in the reference, which trains one view per step on one GPU (trainer.ts:573).  View-sharded data parallelism
 * (SURVEY 8(e)): every rank holds a full replica, runs K1-K17 on its own views, sums GaussianGradients into acc_f32[N*14] +
 * visible[N] (wdgs_accumulate_gradients), all-reduces both, and applies wdgs_optimizer_step_f32 -- identical on every rank,
 * so replicas stay bit-identical; world_size = 1 reduces to the reference step.  One process (or host thread) per GPU: rank 0
 * calls wdgs_comm_get_unique_id and ships the 128 bytes to the others over any host channel; all ranks then call
 * wdgs_comm_create.  Reductions are queued on the device's stream (no host sync).  RCCL is bound lazily, on first use. */
#define WDGS_COMM_ID_BYTES 128
typedef struct wdgs_comm wdgs_comm;
int wdgs_comm_get_unique_id(uint8_t id_out[WDGS_COMM_ID_BYTES]);
int wdgs_comm_create(wdgs_device* dev, const uint8_t id[WDGS_COMM_ID_BYTES], int world_size, int rank, wdgs_comm** out);
int wdgs_comm_destroy(wdgs_comm* comm);
int wdgs_comm_world_size(const wdgs_comm* comm);
int wdgs_comm_rank(const wdgs_comm* comm);
/* In place: grad_f32[N*14] (f32 sum) and visible_counts[N] (u32 sum) over all ranks, one RCCL group. */
int wdgs_comm_allreduce_gradients(wdgs_comm* comm, void* grad_f32_dev, void* visible_counts_dev, uint32_t num_points);
/* In place u32 sum (densify metric counts, SURVEY 8(e) "Determinism"). */
int wdgs_comm_allreduce_counts(wdgs_comm* comm, void* counts_u32_dev, uint32_t count);
/* The bandwidth-optimal form of the same exchange: reduce-scatter -> Adam on the owned slice -> all-gather of the
 * re-packed rows.  Gaussians are dealt to ranks in contiguous slices of slice_points = ceil(N / world_size) rounded up to 64; rank r
 * owns [r*slice, min((r+1)*slice, N)).  grad_f32_dev and visible_counts_dev hold world_size*slice_points rows (tail zero); after the
 * call rank r's OWN slice holds the sums over all ranks (the other slices are scratch).  flag_u32_dev (nullable, one u32) is summed
 * over all ranks in the same RCCL group: a per-rank "tile entries overflowed" word becomes a global one, so every rank skips the
 * step together (wdgs_optimizer_set_guard).  Then wdgs_optimizer_step_f32_range(first = r*slice, rows_out = rows) and
 * wdgs_comm_allgather_rows(rows) (in place, world_size*slice_points rows of 32 bytes) and wdgs_apply_repacked_rows.  Per step and
 * rank this moves (7/8)(60 + 32) B per Gaussian instead of the all-reduce's 2(7/8)60 B, and divides the Adam pass by world_size.
 * The optimizer state of a slice lives on its owner; gather it with wdgs_comm_broadcast (one call per owner and array, optionally
 * inside wdgs_comm_group_start/end) + wdgs_optimizer_state_changed before a densify rebuild or an export. */
int wdgs_comm_exchange_gradients(wdgs_comm* comm, void* grad_f32_dev, void* visible_counts_dev, void* flag_u32_dev, uint32_t slice_points);
int wdgs_comm_allgather_rows(wdgs_comm* comm, void* rows_dev, uint32_t slice_points);
int wdgs_comm_broadcast(wdgs_comm* comm, void* ptr_dev, size_t bytes, int root);
int wdgs_comm_group_start(void);
int wdgs_comm_group_end(void);

#ifdef __cplusplus
}
#endif
#endif /* WEBDGS_H */
