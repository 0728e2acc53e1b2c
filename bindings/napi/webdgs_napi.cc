// N-API addon over the C ABI of libwebdgs_hip.so (include/webdgs.h) -- the binding a WebDGS maintainer would add so that
// the TypeScript-side host (bindings/ts/webdgs_hip.js + .d.ts and trainer.js, shaped like the reference's src/renderers/*.ts) drives the HIP kernels.
// Handles and device pointers cross the boundary as BigInt (64-bit); configs cross as plain objects.
// Build: make -C bindings/napi   (g++, /usr/include/node/node_api.h; no node-gyp needed).
#include <node_api.h>

#include <cstdint>
#include <cstring>
#include <string>

#include "../../include/webdgs.h"

#define NAPI_OK(call)                                                         \
    do {                                                                      \
        if ((call) != napi_ok) {                                              \
            napi_throw_error(env, nullptr, "N-API call failed: " #call);      \
            return nullptr;                                                   \
        }                                                                     \
    } while (0)

static napi_value throw_wdgs(napi_env env, int code) {
    std::string msg = "[wdgs " + std::to_string(code) + "] " + wdgs_last_error();
    napi_throw_error(env, code == WDGS_E_CAPACITY ? "WDGS_E_CAPACITY" : (code == WDGS_E_STATE ? "WDGS_E_STATE" : "WDGS_E"), msg.c_str());
    return nullptr;
}
#define WDGS_OK_OR_THROW(expr)                      \
    do {                                            \
        int _rc = (expr);                           \
        if (_rc != WDGS_OK) return throw_wdgs(env, _rc); \
    } while (0)

static void* get_ptr(napi_env env, napi_value v) {
    napi_valuetype t;
    napi_typeof(env, v, &t);
    if (t == napi_null || t == napi_undefined) return nullptr;
    uint64_t u = 0;
    bool lossless = true;
    if (t == napi_bigint) napi_get_value_bigint_uint64(env, v, &u, &lossless);
    else { double d = 0; napi_get_value_double(env, v, &d); u = (uint64_t)d; }
    return (void*)(uintptr_t)u;
}
static napi_value make_ptr(napi_env env, const void* p) {
    napi_value v;
    napi_create_bigint_uint64(env, (uint64_t)(uintptr_t)p, &v);
    return v;
}
static uint32_t get_u32(napi_env env, napi_value v) { uint32_t u = 0; napi_get_value_uint32(env, v, &u); return u; }
static double get_f64(napi_env env, napi_value v) { double d = 0; napi_get_value_double(env, v, &d); return d; }
static napi_value make_u32(napi_env env, uint32_t u) { napi_value v; napi_create_uint32(env, u, &v); return v; }
static double prop_f64(napi_env env, napi_value obj, const char* name, double dflt) {
    bool has = false;
    napi_has_named_property(env, obj, name, &has);
    if (!has) return dflt;
    napi_value v; napi_get_named_property(env, obj, name, &v);
    napi_valuetype t; napi_typeof(env, v, &t);
    if (t != napi_number) return dflt;
    return get_f64(env, v);
}
static void* prop_ptr(napi_env env, napi_value obj, const char* name) {
    bool has = false;
    napi_has_named_property(env, obj, name, &has);
    if (!has) return nullptr;
    napi_value v; napi_get_named_property(env, obj, name, &v);
    return get_ptr(env, v);
}
static void set_prop(napi_env env, napi_value obj, const char* name, napi_value v) { napi_set_named_property(env, obj, name, v); }

#define ARGS(n)                                                          \
    size_t argc = n;                                                     \
    napi_value argv[n > 0 ? n : 1];                                      \
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr)); \
    if (argc < n) { napi_throw_type_error(env, nullptr, "too few arguments"); return nullptr; }

static napi_value js_undefined(napi_env env) { napi_value v; napi_get_undefined(env, &v); return v; }
static napi_value js_null(napi_env env) { napi_value v; napi_get_null(env, &v); return v; }

// ---- device / queue ---------------------------------------------------------------------------------------------
static napi_value abiVersion(napi_env env, napi_callback_info) { return make_u32(env, (uint32_t)wdgs_abi_version()); }
static napi_value deviceCreate(napi_env env, napi_callback_info info) {
    ARGS(1);
    wdgs_device* d = nullptr;
    WDGS_OK_OR_THROW(wdgs_device_create((int)get_u32(env, argv[0]), nullptr, &d));
    return make_ptr(env, d);
}
static napi_value deviceDestroy(napi_env env, napi_callback_info info) { ARGS(1); wdgs_device_destroy((wdgs_device*)get_ptr(env, argv[0])); return js_undefined(env); }
static napi_value deviceSynchronize(napi_env env, napi_callback_info info) {  // queue.onSubmittedWorkDone(); the TS shim wraps it in a Promise
    ARGS(1);
    WDGS_OK_OR_THROW(wdgs_device_synchronize((wdgs_device*)get_ptr(env, argv[0])));
    return js_undefined(env);
}
static napi_value deviceMemoryInfo(napi_env env, napi_callback_info info) {  // (device) -> {free, total, cached} in bytes (doubles: exact below 2^53)
    ARGS(1);
    size_t f = 0, t = 0, c = 0;
    WDGS_OK_OR_THROW(wdgs_device_memory_info((wdgs_device*)get_ptr(env, argv[0]), &f, &t, &c));
    napi_value o, v; napi_create_object(env, &o);
    napi_create_double(env, (double)f, &v); set_prop(env, o, "free", v);
    napi_create_double(env, (double)t, &v); set_prop(env, o, "total", v);
    napi_create_double(env, (double)c, &v); set_prop(env, o, "cached", v);
    return o;
}
// ---- buffers ----------------------------------------------------------------------------------------------------
static napi_value bufferCreate(napi_env env, napi_callback_info info) {
    ARGS(2);
    wdgs_buffer* b = nullptr;
    WDGS_OK_OR_THROW(wdgs_buffer_create((wdgs_device*)get_ptr(env, argv[0]), (size_t)get_f64(env, argv[1]), &b));
    napi_value o; napi_create_object(env, &o);
    set_prop(env, o, "handle", make_ptr(env, b));
    set_prop(env, o, "ptr", make_ptr(env, wdgs_buffer_ptr(b)));
    napi_value sz; napi_create_double(env, (double)wdgs_buffer_size(b), &sz);
    set_prop(env, o, "size", sz);
    return o;
}
static napi_value bufferDestroy(napi_env env, napi_callback_info info) { ARGS(1); wdgs_buffer_destroy((wdgs_buffer*)get_ptr(env, argv[0])); return js_undefined(env); }
static napi_value copyToDevice(napi_env env, napi_callback_info info) {  // (device, dstPtr, ArrayBufferView)
    ARGS(3);
    void* data = nullptr; size_t len = 0; napi_typedarray_type tt; napi_value ab; size_t off;
    bool is_ta = false; napi_is_typedarray(env, argv[2], &is_ta);
    if (is_ta) { NAPI_OK(napi_get_typedarray_info(env, argv[2], &tt, &len, &data, &ab, &off)); size_t es[] = {1,1,1,2,2,4,4,4,8,8,8}; len *= es[tt]; }
    else NAPI_OK(napi_get_arraybuffer_info(env, argv[2], &data, &len));
    WDGS_OK_OR_THROW(wdgs_copy_to_device((wdgs_device*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), data, len));
    return js_undefined(env);
}
static napi_value copyToHost(napi_env env, napi_callback_info info) {  // (device, srcPtr, byteLength) -> ArrayBuffer
    ARGS(3);
    size_t len = (size_t)get_f64(env, argv[2]);
    void* data = nullptr; napi_value ab;
    NAPI_OK(napi_create_arraybuffer(env, len, &data, &ab));
    WDGS_OK_OR_THROW(wdgs_copy_to_host((wdgs_device*)get_ptr(env, argv[0]), data, get_ptr(env, argv[1]), len));
    return ab;
}
// ---- TiledForwardPass -------------------------------------------------------------------------------------------
static napi_value tiledForwardCreate(napi_env env, napi_callback_info info) {  // (device, {numPoints, shDeg, viewportWidth, ...})
    ARGS(2);
    wdgs_tiled_forward_config c; std::memset(&c, 0, sizeof(c));
    c.num_points = (uint32_t)prop_f64(env, argv[1], "numPoints", 0);
    c.sh_deg = (uint32_t)prop_f64(env, argv[1], "shDeg", 0);
    c.viewport_width = (uint32_t)prop_f64(env, argv[1], "viewportWidth", 1);
    c.viewport_height = (uint32_t)prop_f64(env, argv[1], "viewportHeight", 1);
    c.gaussian_scale = (float)prop_f64(env, argv[1], "gaussianScale", 1.0);
    c.point_size_px = (float)prop_f64(env, argv[1], "pointSizePx", 3.0);
    c.max_splat_radius_px = (float)prop_f64(env, argv[1], "maxSplatRadiusPx", 128.0);
    c.render_mode = (uint32_t)prop_f64(env, argv[1], "renderMode", 1);
    c.max_tile_entries = (uint32_t)prop_f64(env, argv[1], "maxTileEntries", 0);
    c.compat_caps = (uint32_t)prop_f64(env, argv[1], "compatCaps", 0);
    wdgs_tiled_forward* op = nullptr;
    WDGS_OK_OR_THROW(wdgs_tiled_forward_create((wdgs_device*)get_ptr(env, argv[0]), &c, &op));
    return make_ptr(env, op);
}
static napi_value tiledForwardEncode(napi_env env, napi_callback_info info) {  // (op, gaussiansPtr, shPtr, cameraPtr, skipSort)
    ARGS(5);
    WDGS_OK_OR_THROW(wdgs_tiled_forward_encode((wdgs_tiled_forward*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_ptr(env, argv[2]), get_ptr(env, argv[3]),
                                               (int)get_u32(env, argv[4])));
    return js_undefined(env);
}
static napi_value tiledForwardSetViewport(napi_env env, napi_callback_info info) {
    ARGS(3);
    WDGS_OK_OR_THROW(wdgs_tiled_forward_set_viewport((wdgs_tiled_forward*)get_ptr(env, argv[0]), get_u32(env, argv[1]), get_u32(env, argv[2])));
    return js_undefined(env);
}
static napi_value tiledForwardGetResources(napi_env env, napi_callback_info info) {
    ARGS(1);
    wdgs_tiled_forward_resources r;
    WDGS_OK_OR_THROW(wdgs_tiled_forward_get_resources((wdgs_tiled_forward*)get_ptr(env, argv[0]), &r));
    napi_value o; napi_create_object(env, &o);
    set_prop(env, o, "splatBuffer", make_ptr(env, r.splat_buffer));
    set_prop(env, o, "tileKeysBuffer", make_ptr(env, r.tile_keys_buffer));
    set_prop(env, o, "tileIndicesBuffer", make_ptr(env, r.tile_indices_buffer));
    set_prop(env, o, "tileOffsetsBuffer", make_ptr(env, r.tile_offsets_buffer));
    set_prop(env, o, "tileCountsBuffer", make_ptr(env, r.tile_counts_buffer));
    set_prop(env, o, "statsBuffer", make_ptr(env, r.stats_buffer));
    set_prop(env, o, "numTilesX", make_u32(env, r.num_tiles_x));
    set_prop(env, o, "numTilesY", make_u32(env, r.num_tiles_y));
    set_prop(env, o, "totalTiles", make_u32(env, r.total_tiles));
    set_prop(env, o, "maxTileEntries", make_u32(env, r.max_tile_entries));
    return o;
}
static napi_value tiledForwardDestroy(napi_env env, napi_callback_info info) { ARGS(1); wdgs_tiled_forward_destroy((wdgs_tiled_forward*)get_ptr(env, argv[0])); return js_undefined(env); }
// ---- TiledRasterizer --------------------------------------------------------------------------------------------
static napi_value tiledRasterizerCreate(napi_env env, napi_callback_info info) {
    ARGS(2);
    wdgs_tiled_rasterizer* op = nullptr;
    WDGS_OK_OR_THROW(wdgs_tiled_rasterizer_create((wdgs_device*)get_ptr(env, argv[0]), (wdgs_tiled_forward*)get_ptr(env, argv[1]), 0, &op));
    return make_ptr(env, op);
}
static napi_value tiledRasterizerEncode(napi_env env, napi_callback_info info) {
    ARGS(3);
    WDGS_OK_OR_THROW(wdgs_tiled_rasterizer_encode((wdgs_tiled_rasterizer*)get_ptr(env, argv[0]), get_u32(env, argv[1]), get_u32(env, argv[2])));
    return js_undefined(env);
}
static napi_value tiledRasterizerGet(napi_env env, napi_callback_info info) {  // (op, which: 0 output, 1 alpha, 2 nContrib, 3 tileOffsets)
    ARGS(2);
    void* p = nullptr;
    wdgs_tiled_rasterizer* op = (wdgs_tiled_rasterizer*)get_ptr(env, argv[0]);
    switch (get_u32(env, argv[1])) {
        case 0: WDGS_OK_OR_THROW(wdgs_tiled_rasterizer_get_output(op, &p)); break;
        case 1: WDGS_OK_OR_THROW(wdgs_tiled_rasterizer_get_alpha(op, &p)); break;
        case 2: WDGS_OK_OR_THROW(wdgs_tiled_rasterizer_get_n_contrib(op, &p)); break;
        default: WDGS_OK_OR_THROW(wdgs_tiled_rasterizer_get_tile_offsets(op, &p)); break;
    }
    return make_ptr(env, p);
}
static napi_value tiledRasterizerDestroy(napi_env env, napi_callback_info info) { ARGS(1); wdgs_tiled_rasterizer_destroy((wdgs_tiled_rasterizer*)get_ptr(env, argv[0])); return js_undefined(env); }
// ---- TiledBackwardPass ------------------------------------------------------------------------------------------
static napi_value tiledBackwardCreate(napi_env env, napi_callback_info info) {
    ARGS(2);
    wdgs_tiled_backward_config c; std::memset(&c, 0, sizeof(c));
    c.num_points = (uint32_t)prop_f64(env, argv[1], "numPoints", 0);
    c.sh_deg = (uint32_t)prop_f64(env, argv[1], "shDeg", 0);
    c.viewport_width = (uint32_t)prop_f64(env, argv[1], "viewportWidth", 1);
    c.viewport_height = (uint32_t)prop_f64(env, argv[1], "viewportHeight", 1);
    c.training.lambda_l1 = (float)prop_f64(env, argv[1], "lambda_l1", 0.8);
    c.training.lambda_l2 = (float)prop_f64(env, argv[1], "lambda_l2", 0.0);
    c.training.lambda_dssim = (float)prop_f64(env, argv[1], "lambda_dssim", 0.2);
    c.training.c1 = (float)prop_f64(env, argv[1], "c1", 0.0001);
    c.training.c2 = (float)prop_f64(env, argv[1], "c2", 0.0009);
    c.max_splat_radius_px = (float)prop_f64(env, argv[1], "maxSplatRadiusPx", 128.0);
    wdgs_tiled_backward* op = nullptr;
    WDGS_OK_OR_THROW(wdgs_tiled_backward_create((wdgs_device*)get_ptr(env, argv[0]), &c, &op));
    return make_ptr(env, op);
}
static napi_value tiledBackwardEncode(napi_env env, napi_callback_info info) {  // (op, predPtr, targetPtr, resources{...Ptr}, gaussiansPtr)
    ARGS(5);
    wdgs_tiled_backward_resources r;
    r.splat_buffer = prop_ptr(env, argv[3], "splatBuffer");
    r.tile_offsets_buffer = prop_ptr(env, argv[3], "tileOffsetsBuffer");
    r.tile_indices_buffer = prop_ptr(env, argv[3], "tileIndicesBuffer");
    r.camera_buffer = prop_ptr(env, argv[3], "cameraBuffer");
    r.alpha_texture = prop_ptr(env, argv[3], "alphaTexture");
    r.n_contrib_texture = prop_ptr(env, argv[3], "nContribTexture");
    WDGS_OK_OR_THROW(wdgs_tiled_backward_encode((wdgs_tiled_backward*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_ptr(env, argv[2]), &r, get_ptr(env, argv[4])));
    return js_undefined(env);
}
static napi_value tiledBackwardGradients(napi_env env, napi_callback_info info) { ARGS(1); return make_ptr(env, wdgs_tiled_backward_gradients((wdgs_tiled_backward*)get_ptr(env, argv[0]))); }
static napi_value tiledBackwardDestroy(napi_env env, napi_callback_info info) { ARGS(1); wdgs_tiled_backward_destroy((wdgs_tiled_backward*)get_ptr(env, argv[0])); return js_undefined(env); }
// ---- Optimizer --------------------------------------------------------------------------------------------------
static napi_value optimizerCreate(napi_env env, napi_callback_info info) {  // (device, numPoints, gaussiansPtr, shPtr) -> library-owned state
    ARGS(4);
    wdgs_optimizer* op = nullptr;
    WDGS_OK_OR_THROW(wdgs_optimizer_create((wdgs_device*)get_ptr(env, argv[0]), get_u32(env, argv[1]), nullptr, get_ptr(env, argv[2]), get_ptr(env, argv[3]), nullptr, 1, 0, &op));
    return make_ptr(env, op);
}
static napi_value optimizerStep(napi_env env, napi_callback_info info) {  // (op, gaussiansPtr, shPtr, gradientsPtr, tileCountsPtr)
    ARGS(5);
    WDGS_OK_OR_THROW(wdgs_optimizer_step((wdgs_optimizer*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_ptr(env, argv[2]), get_ptr(env, argv[3]), get_ptr(env, argv[4])));
    return js_undefined(env);
}
static napi_value optimizerGetIteration(napi_env env, napi_callback_info info) { ARGS(1); return make_u32(env, wdgs_optimizer_get_iteration((wdgs_optimizer*)get_ptr(env, argv[0]))); }
static napi_value optimizerDestroy(napi_env env, napi_callback_info info) { ARGS(1); wdgs_optimizer_destroy((wdgs_optimizer*)get_ptr(env, argv[0])); return js_undefined(env); }

// ---- remaining surface: setters, blit, recorded command buffers, completion Promise, metric path, densify, DP ----------------
static napi_value tiledForwardSet(napi_env env, napi_callback_info info) {  // (op, what: 'renderMode'|'pointSize'|'gaussianScale' as 0|1|2, value)
    ARGS(3);
    wdgs_tiled_forward* op = (wdgs_tiled_forward*)get_ptr(env, argv[0]);
    switch (get_u32(env, argv[1])) {
        case 0: WDGS_OK_OR_THROW(wdgs_tiled_forward_set_render_mode(op, get_u32(env, argv[2]))); break;
        case 1: WDGS_OK_OR_THROW(wdgs_tiled_forward_set_point_size(op, (float)get_f64(env, argv[2]))); break;
        default: WDGS_OK_OR_THROW(wdgs_tiled_forward_set_gaussian_scale(op, (float)get_f64(env, argv[2]))); break;
    }
    return js_undefined(env);
}
static napi_value tiledForwardCheck(napi_env env, napi_callback_info info) {  // -> {totalTileEntries, visibleCount}; throws WDGS_E_CAPACITY on overflow
    ARGS(1);
    uint32_t st[4] = {0, 0, 0, 0};
    WDGS_OK_OR_THROW(wdgs_tiled_forward_check((wdgs_tiled_forward*)get_ptr(env, argv[0]), st));
    napi_value o; napi_create_object(env, &o);
    set_prop(env, o, "totalTileEntries", make_u32(env, st[0]));
    set_prop(env, o, "visibleCount", make_u32(env, st[1]));
    return o;
}
static napi_value tiledForwardSetLongLists(napi_env env, napi_callback_info info) {  // (op, threshold, maxItems, maxRows)
    ARGS(4);
    WDGS_OK_OR_THROW(wdgs_tiled_forward_set_long_lists((wdgs_tiled_forward*)get_ptr(env, argv[0]), get_u32(env, argv[1]), get_u32(env, argv[2]), get_u32(env, argv[3])));
    return js_undefined(env);
}
static napi_value tiledForwardLongListStats(napi_env env, napi_callback_info info) {  // (op) -> {blocksWanted, itemsWanted, ..., threshold}
    ARGS(1);
    uint32_t st[12] = {0};
    WDGS_OK_OR_THROW(wdgs_tiled_forward_long_list_stats((wdgs_tiled_forward*)get_ptr(env, argv[0]), st));
    static const char* const keys[12] = {"blocksWanted", "itemsWanted", "forwardQueue", "backwardQueue", "rowsUsed", "rowsWanted", "stalled", nullptr, "maxItems", "maxBlocks", "maxRows", "threshold"};
    napi_value o; napi_create_object(env, &o);
    for (int i = 0; i < 12; i++) if (keys[i]) set_prop(env, o, keys[i], make_u32(env, st[i]));
    return o;
}
static napi_value tiledRasterizerBlit(napi_env env, napi_callback_info info) {  // (op, targetPtr, width, height): blitToTexture
    ARGS(4);
    WDGS_OK_OR_THROW(wdgs_tiled_rasterizer_blit((wdgs_tiled_rasterizer*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_u32(env, argv[2]), get_u32(env, argv[3])));
    return js_undefined(env);
}
static napi_value bufferClear(napi_env env, napi_callback_info info) {  // encoder.clearBuffer: (device, ptr, byteLength)
    ARGS(3);
    WDGS_OK_OR_THROW(wdgs_memset((wdgs_device*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), 0, (size_t)get_f64(env, argv[2])));
    return js_undefined(env);
}
static napi_value encoderBegin(napi_env env, napi_callback_info info) { ARGS(1); WDGS_OK_OR_THROW(wdgs_encoder_begin((wdgs_device*)get_ptr(env, argv[0]))); return js_undefined(env); }
static napi_value encoderFinish(napi_env env, napi_callback_info info) {
    ARGS(1);
    wdgs_command_buffer* c = nullptr;
    WDGS_OK_OR_THROW(wdgs_encoder_finish((wdgs_device*)get_ptr(env, argv[0]), &c));
    return make_ptr(env, c);
}
static napi_value queueSubmit(napi_env env, napi_callback_info info) {
    ARGS(2);
    WDGS_OK_OR_THROW(wdgs_queue_submit((wdgs_device*)get_ptr(env, argv[0]), (wdgs_command_buffer*)get_ptr(env, argv[1])));
    return js_undefined(env);
}
static napi_value commandBufferDestroy(napi_env env, napi_callback_info info) { ARGS(1); wdgs_command_buffer_destroy((wdgs_command_buffer*)get_ptr(env, argv[0])); return js_undefined(env); }

// queue.onSubmittedWorkDone(): Promise<void> resolved from the HIP runtime thread through a thread-safe function.
struct DoneCtx { napi_deferred deferred; napi_threadsafe_function tsfn; };
static void done_call_js(napi_env env, napi_value, void* context, void*) {
    DoneCtx* c = static_cast<DoneCtx*>(context);
    if (env) { napi_value u; napi_get_undefined(env, &u); napi_resolve_deferred(env, c->deferred, u); }
    napi_release_threadsafe_function(c->tsfn, napi_tsfn_release);
    delete c;
}
static void done_from_runtime_thread(void* user) {
    DoneCtx* c = static_cast<DoneCtx*>(user);
    napi_call_threadsafe_function(c->tsfn, nullptr, napi_tsfn_blocking);
}
static napi_value queueOnSubmittedWorkDone(napi_env env, napi_callback_info info) {
    ARGS(1);
    DoneCtx* c = new DoneCtx();
    napi_value promise, name;
    NAPI_OK(napi_create_promise(env, &c->deferred, &promise));
    napi_create_string_utf8(env, "wdgs onSubmittedWorkDone", NAPI_AUTO_LENGTH, &name);
    NAPI_OK(napi_create_threadsafe_function(env, nullptr, nullptr, name, 0, 1, nullptr, nullptr, c, done_call_js, &c->tsfn));
    int rc = wdgs_queue_on_done((wdgs_device*)get_ptr(env, argv[0]), done_from_runtime_thread, c);
    if (rc != WDGS_OK) { napi_release_threadsafe_function(c->tsfn, napi_tsfn_abort); delete c; return throw_wdgs(env, rc); }
    return promise;
}

static void read_backward_resources(napi_env env, napi_value o, wdgs_tiled_backward_resources* r) {
    r->splat_buffer = prop_ptr(env, o, "splatBuffer");
    r->tile_offsets_buffer = prop_ptr(env, o, "tileOffsetsBuffer");
    r->tile_indices_buffer = prop_ptr(env, o, "tileIndicesBuffer");
    r->camera_buffer = prop_ptr(env, o, "cameraBuffer");
    r->alpha_texture = prop_ptr(env, o, "alphaTexture");
    r->n_contrib_texture = prop_ptr(env, o, "nContribTexture");
}
static napi_value tiledBackwardMetric(napi_env env, napi_callback_info info) {
    // (op, stage, a, b, c): 0 computeLossOnly(pred, target) | 1 computeMetricMap(pred, target, threshold) |
    //                       2 computeMetricCounts(resources, numInstances, clear) | 3 normalizeMetricCounts(divisor) | 4 setViewport(w, h) |
    //                       5 setTrainingConfig(config)
    ARGS(5);
    wdgs_tiled_backward* op = (wdgs_tiled_backward*)get_ptr(env, argv[0]);
    switch (get_u32(env, argv[1])) {
        case 0: WDGS_OK_OR_THROW(wdgs_tiled_backward_compute_loss_only(op, get_ptr(env, argv[2]), get_ptr(env, argv[3]))); break;
        case 1: WDGS_OK_OR_THROW(wdgs_tiled_backward_compute_metric_map(op, get_ptr(env, argv[2]), get_ptr(env, argv[3]), (float)get_f64(env, argv[4]))); break;
        case 2: {
            wdgs_tiled_backward_resources r;
            read_backward_resources(env, argv[2], &r);
            WDGS_OK_OR_THROW(wdgs_tiled_backward_compute_metric_counts(op, &r, get_u32(env, argv[3]), (int)get_u32(env, argv[4])));
            break;
        }
        case 3: WDGS_OK_OR_THROW(wdgs_tiled_backward_normalize_metric_counts(op, get_u32(env, argv[2]))); break;
        case 5: {  // setTrainingConfig({lambda_l1, lambda_l2, lambda_dssim, c1, c2}): the complete config (tiled-backward-pass.ts:812-830)
            wdgs_training_config t;
            t.lambda_l1 = (float)prop_f64(env, argv[2], "lambda_l1", 0.8);
            t.lambda_l2 = (float)prop_f64(env, argv[2], "lambda_l2", 0.0);
            t.lambda_dssim = (float)prop_f64(env, argv[2], "lambda_dssim", 0.2);
            t.c1 = (float)prop_f64(env, argv[2], "c1", 0.0001);
            t.c2 = (float)prop_f64(env, argv[2], "c2", 0.0009);
            WDGS_OK_OR_THROW(wdgs_tiled_backward_set_training_config(op, &t));
            break;
        }
        default: WDGS_OK_OR_THROW(wdgs_tiled_backward_set_viewport(op, get_u32(env, argv[2]), get_u32(env, argv[3]))); break;
    }
    return js_undefined(env);
}
static napi_value tiledBackwardGet(napi_env env, napi_callback_info info) {  // (op, which: 0 gradients, 1 metricCounts, 2 lossImage, 3 metricMap)
    ARGS(2);
    wdgs_tiled_backward* op = (wdgs_tiled_backward*)get_ptr(env, argv[0]);
    switch (get_u32(env, argv[1])) {
        case 0: return make_ptr(env, wdgs_tiled_backward_gradients(op));
        case 1: return make_ptr(env, wdgs_tiled_backward_metric_counts(op));
        case 2: return make_ptr(env, wdgs_tiled_backward_loss_image(op));
        default: return make_ptr(env, wdgs_tiled_backward_metric_map(op));
    }
}
static napi_value downsampleRGBA8(napi_env env, napi_callback_info info) {  // (device, srcPtr, srcW, srcH, dstPtr, dstW, dstH)
    ARGS(7);
    WDGS_OK_OR_THROW(wdgs_downsample_rgba8((wdgs_device*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_u32(env, argv[2]), get_u32(env, argv[3]),
                                           get_ptr(env, argv[4]), get_u32(env, argv[5]), get_u32(env, argv[6])));
    return js_undefined(env);
}
static napi_value imageSSE(napi_env env, napi_callback_info info) {  // (device, aPtr, bPtr, numPixels, outU64Ptr)
    ARGS(5);
    WDGS_OK_OR_THROW(wdgs_image_sse_rgb8((wdgs_device*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_ptr(env, argv[2]), get_u32(env, argv[3]), get_ptr(env, argv[4])));
    return js_undefined(env);
}

static bool read_state(napi_env env, napi_value o, wdgs_optimizer_state* s) {
    napi_valuetype t; napi_typeof(env, o, &t);
    if (t != napi_object) return false;
    s->opt_pos = prop_ptr(env, o, "optPosBuffer"); s->opt_rot = prop_ptr(env, o, "optRotBuffer"); s->opt_scale = prop_ptr(env, o, "optScaleBuffer");
    s->opt_opacity = prop_ptr(env, o, "optOpacityBuffer"); s->param_sh = prop_ptr(env, o, "paramSH"); s->state_sh = prop_ptr(env, o, "stateSH");
    return true;
}
static napi_value make_state(napi_env env, const wdgs_optimizer_state& s) {
    napi_value o; napi_create_object(env, &o);
    set_prop(env, o, "optPosBuffer", make_ptr(env, s.opt_pos)); set_prop(env, o, "optRotBuffer", make_ptr(env, s.opt_rot));
    set_prop(env, o, "optScaleBuffer", make_ptr(env, s.opt_scale)); set_prop(env, o, "optOpacityBuffer", make_ptr(env, s.opt_opacity));
    set_prop(env, o, "paramSH", make_ptr(env, s.param_sh)); set_prop(env, o, "stateSH", make_ptr(env, s.state_sh));
    return o;
}
static napi_value optimizerStateSizes(napi_env env, napi_callback_info info) {  // allocateOptimizerStateBuffers: byte sizes of the six arrays
    ARGS(1);
    size_t sz[6];
    WDGS_OK_OR_THROW(wdgs_optimizer_state_sizes(get_u32(env, argv[0]), sz));
    napi_value arr; napi_create_array_with_length(env, 6, &arr);
    for (uint32_t i = 0; i < 6; i++) { napi_value v; napi_create_double(env, (double)sz[i], &v); napi_set_element(env, arr, i, v); }
    return arr;
}
static napi_value optimizerCreateWithState(napi_env env, napi_callback_info info) {
    // (device, numPoints, gaussiansPtr, shPtr, initialState{...Ptr} (adopted), ownsState, initialIteration)
    ARGS(7);
    wdgs_optimizer_state st; std::memset(&st, 0, sizeof(st));
    const bool has = read_state(env, argv[4], &st);
    wdgs_optimizer* op = nullptr;
    WDGS_OK_OR_THROW(wdgs_optimizer_create((wdgs_device*)get_ptr(env, argv[0]), get_u32(env, argv[1]), nullptr, get_ptr(env, argv[2]), get_ptr(env, argv[3]),
                                           has ? &st : nullptr, (int)get_u32(env, argv[5]), get_u32(env, argv[6]), &op));
    return make_ptr(env, op);
}
static napi_value optimizerState(napi_env env, napi_callback_info info) {  // (op, release: 0 getStateBuffers | 1 detach for a densify swap)
    ARGS(2);
    wdgs_optimizer_state st;
    if (get_u32(env, argv[1])) WDGS_OK_OR_THROW(wdgs_optimizer_release_state((wdgs_optimizer*)get_ptr(env, argv[0]), &st));
    else WDGS_OK_OR_THROW(wdgs_optimizer_get_state((wdgs_optimizer*)get_ptr(env, argv[0]), &st));
    return make_state(env, st);
}
static napi_value optimizerHyperparameters(napi_env env, napi_callback_info info) {  // (op, next?) -> current (after applying `next`)
    ARGS(2);
    wdgs_optimizer* op = (wdgs_optimizer*)get_ptr(env, argv[0]);
    wdgs_adam_hyperparameters h;
    WDGS_OK_OR_THROW(wdgs_optimizer_get_hyperparameters(op, &h));
    napi_valuetype t; napi_typeof(env, argv[1], &t);
    if (t == napi_object) {
        h.lr_pos = (float)prop_f64(env, argv[1], "lr_pos", h.lr_pos); h.lr_color = (float)prop_f64(env, argv[1], "lr_color", h.lr_color);
        h.lr_opacity = (float)prop_f64(env, argv[1], "lr_opacity", h.lr_opacity); h.lr_scale = (float)prop_f64(env, argv[1], "lr_scale", h.lr_scale);
        h.lr_rot = (float)prop_f64(env, argv[1], "lr_rot", h.lr_rot); h.beta1 = (float)prop_f64(env, argv[1], "beta1", h.beta1);
        h.beta2 = (float)prop_f64(env, argv[1], "beta2", h.beta2); h.epsilon = (float)prop_f64(env, argv[1], "epsilon", h.epsilon);
        WDGS_OK_OR_THROW(wdgs_optimizer_set_hyperparameters(op, &h));
    }
    napi_value o; napi_create_object(env, &o);
    const char* names[8] = {"lr_pos", "lr_color", "lr_opacity", "lr_scale", "lr_rot", "beta1", "beta2", "epsilon"};
    const float vals[8] = {h.lr_pos, h.lr_color, h.lr_opacity, h.lr_scale, h.lr_rot, h.beta1, h.beta2, h.epsilon};
    for (int i = 0; i < 8; i++) { napi_value v; napi_create_double(env, vals[i], &v); set_prop(env, o, names[i], v); }
    return o;
}
static napi_value optimizerAdvanceIteration(napi_env env, napi_callback_info info) {
    ARGS(2);
    WDGS_OK_OR_THROW(wdgs_optimizer_advance_iteration((wdgs_optimizer*)get_ptr(env, argv[0]), get_u32(env, argv[1])));
    return js_undefined(env);
}
static napi_value optimizerStepF32(napi_env env, napi_callback_info info) {  // data-parallel step: (op, gaussiansPtr, shPtr, gradF32Ptr, visibleCountsPtr)
    ARGS(5);
    WDGS_OK_OR_THROW(wdgs_optimizer_step_f32((wdgs_optimizer*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_ptr(env, argv[2]), get_ptr(env, argv[3]), get_ptr(env, argv[4])));
    return js_undefined(env);
}
static napi_value accumulateGradients(napi_env env, napi_callback_info info) {  // (device, numPoints, gradientsPtr, tileCountsPtr, accF32Ptr, visiblePtr)
    ARGS(6);
    WDGS_OK_OR_THROW(wdgs_accumulate_gradients((wdgs_device*)get_ptr(env, argv[0]), get_u32(env, argv[1]), get_ptr(env, argv[2]), get_ptr(env, argv[3]),
                                               get_ptr(env, argv[4]), get_ptr(env, argv[5])));
    return js_undefined(env);
}

static void read_densify_config(napi_env env, napi_value o, wdgs_densify_config* c) {
    c->num_views = (uint32_t)prop_f64(env, o, "numViews", 1);
    c->clone_threshold = (uint32_t)prop_f64(env, o, "cloneThreshold", 0);
    c->split_threshold = (float)prop_f64(env, o, "splitThreshold", 1e9);
    c->prune_threshold = (float)prop_f64(env, o, "pruneThreshold", 0.0);
    c->max_new_points_per_step = (uint32_t)prop_f64(env, o, "maxNewPointsPerStep", 0);
    c->max_buffer_bytes = (uint64_t)prop_f64(env, o, "maxBufferBytes", 128.0 * 1024 * 1024);
}
static napi_value densifyCreate(napi_env env, napi_callback_info info) {  // (device, config)
    ARGS(2);
    wdgs_densify_config c; read_densify_config(env, argv[1], &c);
    wdgs_densify_prune* op = nullptr;
    WDGS_OK_OR_THROW(wdgs_densify_prune_create((wdgs_device*)get_ptr(env, argv[0]), &c, &op));
    return make_ptr(env, op);
}
static napi_value densifySetConfig(napi_env env, napi_callback_info info) {
    ARGS(2);
    wdgs_densify_config c; read_densify_config(env, argv[1], &c);
    WDGS_OK_OR_THROW(wdgs_densify_prune_set_config((wdgs_densify_prune*)get_ptr(env, argv[0]), &c));
    return js_undefined(env);
}
static napi_value densifyEncodePrepare(napi_env env, napi_callback_info info) {  // (op, numPoints, gaussiansPtr, metricCountsPtr) -> DensifyPrunePrepared
    ARGS(4);
    wdgs_densify_prepared p;
    WDGS_OK_OR_THROW(wdgs_densify_prune_encode_prepare((wdgs_densify_prune*)get_ptr(env, argv[0]), get_u32(env, argv[1]), get_ptr(env, argv[2]), get_ptr(env, argv[3]), &p));
    napi_value o; napi_create_object(env, &o);
    set_prop(env, o, "actionBuffer", make_ptr(env, p.action_buffer)); set_prop(env, o, "outCountBuffer", make_ptr(env, p.out_count_buffer));
    set_prop(env, o, "outOffsetBuffer", make_ptr(env, p.out_offset_buffer)); set_prop(env, o, "outTotalBuffer", make_ptr(env, p.out_total_buffer));
    set_prop(env, o, "maxOutPoints", make_u32(env, p.max_out_points));
    return o;
}
static napi_value densifyStage(napi_env env, napi_callback_info info) {
    // (op, stage, numPoints, a, b): 0 encodeDecision(gaussiansPtr, metricCountsPtr) | 1 encodePrefixSum | 2 encodeCapToMax(maxOutPoints) |
    //                               3 encodeTotalOut | 4 ensureSize;  returns the pass's work buffers (DensifyPrunePrepared shape)
    ARGS(5);
    wdgs_densify_prune* op = (wdgs_densify_prune*)get_ptr(env, argv[0]);
    const uint32_t n = get_u32(env, argv[2]);
    switch (get_u32(env, argv[1])) {
        case 0: WDGS_OK_OR_THROW(wdgs_densify_prune_encode_decision(op, n, get_ptr(env, argv[3]), get_ptr(env, argv[4]))); break;
        case 1: WDGS_OK_OR_THROW(wdgs_densify_prune_encode_prefix_sum(op, n)); break;
        case 2: WDGS_OK_OR_THROW(wdgs_densify_prune_encode_cap_to_max(op, n, get_u32(env, argv[3]))); break;
        case 3: WDGS_OK_OR_THROW(wdgs_densify_prune_encode_total_out(op, n)); break;
        default: WDGS_OK_OR_THROW(wdgs_densify_prune_ensure_size(op, n)); break;
    }
    wdgs_densify_prepared p;
    WDGS_OK_OR_THROW(wdgs_densify_prune_get_buffers(op, &p));
    uint32_t max_out = 0;
    WDGS_OK_OR_THROW(wdgs_densify_prune_compute_max_out_points(op, n, &max_out));
    napi_value o; napi_create_object(env, &o);
    set_prop(env, o, "actionBuffer", make_ptr(env, p.action_buffer)); set_prop(env, o, "outCountBuffer", make_ptr(env, p.out_count_buffer));
    set_prop(env, o, "outOffsetBuffer", make_ptr(env, p.out_offset_buffer)); set_prop(env, o, "outTotalBuffer", make_ptr(env, p.out_total_buffer));
    set_prop(env, o, "maxOutPoints", make_u32(env, max_out));
    return o;
}
static napi_value densifyReadTotal(napi_env env, napi_callback_info info) {
    ARGS(1);
    uint32_t t = 0;
    WDGS_OK_OR_THROW(wdgs_densify_prune_read_total((wdgs_densify_prune*)get_ptr(env, argv[0]), &t));
    return make_u32(env, t);
}
static napi_value densifyEncodeScatter(napi_env env, napi_callback_info info) {
    // (op, inPoints, inGaussiansPtr, inShPtr, inState|null, outNumPoints, resetNewOptimizerState, outGaussiansPtr, outShPtr, outState|null)
    ARGS(10);
    wdgs_optimizer_state in_st, out_st;
    const bool has_in = read_state(env, argv[4], &in_st), has_out = read_state(env, argv[9], &out_st);
    WDGS_OK_OR_THROW(wdgs_densify_prune_encode_scatter((wdgs_densify_prune*)get_ptr(env, argv[0]), get_u32(env, argv[1]), get_ptr(env, argv[2]), get_ptr(env, argv[3]),
                                                       has_in ? &in_st : nullptr, get_u32(env, argv[5]), (int)get_u32(env, argv[6]), get_ptr(env, argv[7]),
                                                       get_ptr(env, argv[8]), has_out ? &out_st : nullptr));
    return js_undefined(env);
}
static napi_value densifyDestroy(napi_env env, napi_callback_info info) { ARGS(1); wdgs_densify_prune_destroy((wdgs_densify_prune*)get_ptr(env, argv[0])); return js_undefined(env); }

static napi_value commUniqueId(napi_env env, napi_callback_info) {  // -> ArrayBuffer(128); rank 0 ships it to the other ranks
    void* data = nullptr; napi_value ab;
    NAPI_OK(napi_create_arraybuffer(env, WDGS_COMM_ID_BYTES, &data, &ab));
    WDGS_OK_OR_THROW(wdgs_comm_get_unique_id((uint8_t*)data));
    return ab;
}
static napi_value commCreate(napi_env env, napi_callback_info info) {  // (device, idArrayBuffer, worldSize, rank)
    ARGS(4);
    void* data = nullptr; size_t len = 0;
    NAPI_OK(napi_get_arraybuffer_info(env, argv[1], &data, &len));
    if (len != WDGS_COMM_ID_BYTES) { napi_throw_type_error(env, nullptr, "unique id must be 128 bytes"); return nullptr; }
    wdgs_comm* c = nullptr;
    WDGS_OK_OR_THROW(wdgs_comm_create((wdgs_device*)get_ptr(env, argv[0]), (const uint8_t*)data, (int)get_u32(env, argv[2]), (int)get_u32(env, argv[3]), &c));
    return make_ptr(env, c);
}
static napi_value commAllreduceGradients(napi_env env, napi_callback_info info) {  // (comm, gradF32Ptr, visiblePtr, numPoints)
    ARGS(4);
    WDGS_OK_OR_THROW(wdgs_comm_allreduce_gradients((wdgs_comm*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_ptr(env, argv[2]), get_u32(env, argv[3])));
    return js_undefined(env);
}
static napi_value commAllreduceCounts(napi_env env, napi_callback_info info) {
    ARGS(3);
    WDGS_OK_OR_THROW(wdgs_comm_allreduce_counts((wdgs_comm*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_u32(env, argv[2])));
    return js_undefined(env);
}
static napi_value commDestroy(napi_env env, napi_callback_info info) { ARGS(1); wdgs_comm_destroy((wdgs_comm*)get_ptr(env, argv[0])); return js_undefined(env); }

// ---- additions of round 2: recording abort, pinned read-back, scanner / sorter, optimizer guard, sliced data-parallel exchange ------
static napi_value encoderAbort(napi_env env, napi_callback_info info) { ARGS(1); WDGS_OK_OR_THROW(wdgs_encoder_abort((wdgs_device*)get_ptr(env, argv[0]))); return js_undefined(env); }
// resize instead of rebuild (include/webdgs.h: wdgs_tiled_forward_resize / wdgs_tiled_backward_resize): (passHandle, numPoints)
static napi_value tiledForwardResize(napi_env env, napi_callback_info info) { ARGS(2); WDGS_OK_OR_THROW(wdgs_tiled_forward_resize((wdgs_tiled_forward*)get_ptr(env, argv[0]), get_u32(env, argv[1]))); return js_undefined(env); }
static napi_value tiledBackwardResize(napi_env env, napi_callback_info info) { ARGS(2); WDGS_OK_OR_THROW(wdgs_tiled_backward_resize((wdgs_tiled_backward*)get_ptr(env, argv[0]), get_u32(env, argv[1]))); return js_undefined(env); }
// tickets (include/webdgs.h: wdgs_queue_mark / wdgs_queue_wait): the ticket travels as a double (exact below 2^53)
static napi_value queueMark(napi_env env, napi_callback_info info) {
    ARGS(1);
    uint64_t t = 0;
    WDGS_OK_OR_THROW(wdgs_queue_mark((wdgs_device*)get_ptr(env, argv[0]), &t));
    napi_value v;
    napi_create_double(env, (double)t, &v);
    return v;
}
static napi_value queueWait(napi_env env, napi_callback_info info) { ARGS(2); WDGS_OK_OR_THROW(wdgs_queue_wait((wdgs_device*)get_ptr(env, argv[0]), (uint64_t)get_f64(env, argv[1]))); return js_undefined(env); }
// lanes (include/webdgs.h): (device, lane) / (device, waiterLane, signalLane)
static napi_value deviceSelectLane(napi_env env, napi_callback_info info) { ARGS(2); WDGS_OK_OR_THROW(wdgs_device_select_lane((wdgs_device*)get_ptr(env, argv[0]), (int)get_u32(env, argv[1]))); return js_undefined(env); }
static napi_value deviceLaneOrder(napi_env env, napi_callback_info info) {
    ARGS(3);
    WDGS_OK_OR_THROW(wdgs_device_lane_order((wdgs_device*)get_ptr(env, argv[0]), (int)get_u32(env, argv[1]), (int)get_u32(env, argv[2])));
    return js_undefined(env);
}
static void free_pinned(napi_env, void* data, void*) { wdgs_host_free(data); }
static napi_value hostAlloc(napi_env env, napi_callback_info info) {  // (byteLength) -> ArrayBuffer over PINNED host memory (wdgs_host_alloc)
    ARGS(1);
    const size_t len = (size_t)get_f64(env, argv[0]);
    void* p = nullptr;
    WDGS_OK_OR_THROW(wdgs_host_alloc(len, &p));
    napi_value ab;
    if (napi_create_external_arraybuffer(env, p, len, free_pinned, nullptr, &ab) != napi_ok) { wdgs_host_free(p); napi_throw_error(env, nullptr, "external ArrayBuffer refused"); return nullptr; }
    return ab;
}
static napi_value bufferReadAsync(napi_env env, napi_callback_info info) {
    // (device, bufferHandle, offset, pinnedArrayBuffer, byteLength): mapAsync(READ) -- queues the copy; the bytes are valid once a later
    // queueOnSubmittedWorkDone() Promise resolves (trainer.ts:455-458)
    ARGS(5);
    void* data = nullptr; size_t len = 0;
    NAPI_OK(napi_get_arraybuffer_info(env, argv[3], &data, &len));
    const size_t want = (size_t)get_f64(env, argv[4]);
    if (want > len) { napi_throw_range_error(env, nullptr, "bufferReadAsync: destination too small"); return nullptr; }
    WDGS_OK_OR_THROW(wdgs_buffer_read_async((wdgs_device*)get_ptr(env, argv[0]), (const wdgs_buffer*)get_ptr(env, argv[1]), (size_t)get_f64(env, argv[2]), data, want));
    return js_undefined(env);
}
static napi_value prefixScanner(napi_env env, napi_callback_info info) {
    // (op: 0 create(device, maxElements) -> {handle, inputBuffer, outputBuffer} | 1 setCount(handle, n) | 2 scan(handle) | 3 destroy(handle))
    ARGS(3);
    switch (get_u32(env, argv[0])) {
        case 0: {
            wdgs_prefix_scanner* s = nullptr;
            WDGS_OK_OR_THROW(wdgs_prefix_scanner_create((wdgs_device*)get_ptr(env, argv[1]), get_u32(env, argv[2]), &s));
            napi_value o; napi_create_object(env, &o);
            set_prop(env, o, "handle", make_ptr(env, s));
            set_prop(env, o, "inputBuffer", make_ptr(env, wdgs_prefix_scanner_input(s)));
            set_prop(env, o, "outputBuffer", make_ptr(env, wdgs_prefix_scanner_output(s)));
            return o;
        }
        case 1: WDGS_OK_OR_THROW(wdgs_prefix_scanner_set_count((wdgs_prefix_scanner*)get_ptr(env, argv[1]), get_u32(env, argv[2]))); break;
        case 2: WDGS_OK_OR_THROW(wdgs_prefix_scanner_scan((wdgs_prefix_scanner*)get_ptr(env, argv[1]))); break;
        default: wdgs_prefix_scanner_destroy((wdgs_prefix_scanner*)get_ptr(env, argv[1])); break;
    }
    return js_undefined(env);
}
static napi_value dynamicSorter(napi_env env, napi_callback_info info) {
    // (op: 0 create(device, maxCapacity, statsPtr) -> {handle, capacity, keys0, values0, keys1, values1} | 1 sort(handle, keyBits) -> final_out_index |
    //      2 destroy(handle))
    ARGS(4);
    switch (get_u32(env, argv[0])) {
        case 0: {
            wdgs_sorter* s = nullptr;
            WDGS_OK_OR_THROW(wdgs_sorter_create((wdgs_device*)get_ptr(env, argv[1]), get_u32(env, argv[2]), get_ptr(env, argv[3]), &s));
            napi_value o; napi_create_object(env, &o);
            set_prop(env, o, "handle", make_ptr(env, s));
            set_prop(env, o, "capacity", make_u32(env, wdgs_sorter_capacity(s)));
            set_prop(env, o, "keys0", make_ptr(env, wdgs_sorter_keys(s, 0))); set_prop(env, o, "values0", make_ptr(env, wdgs_sorter_values(s, 0)));
            set_prop(env, o, "keys1", make_ptr(env, wdgs_sorter_keys(s, 1))); set_prop(env, o, "values1", make_ptr(env, wdgs_sorter_values(s, 1)));
            return o;
        }
        case 1: {
            wdgs_sorter* s = (wdgs_sorter*)get_ptr(env, argv[1]);
            WDGS_OK_OR_THROW(wdgs_sorter_sort(s, get_u32(env, argv[2])));
            return make_u32(env, (uint32_t)wdgs_sorter_final_out_index(s));
        }
        default: wdgs_sorter_destroy((wdgs_sorter*)get_ptr(env, argv[1])); break;
    }
    return js_undefined(env);
}
static napi_value optimizerSetGuard(napi_env env, napi_callback_info info) {  // (optimizer, flagPtr | null)
    ARGS(2);
    WDGS_OK_OR_THROW(wdgs_optimizer_set_guard((wdgs_optimizer*)get_ptr(env, argv[0]), get_ptr(env, argv[1])));
    return js_undefined(env);
}
// ---- deferred SH writes (include/webdgs.h: wdgs_optimizer_set_deferred_sh)
static napi_value optimizerDeferredSH(napi_env env, napi_callback_info info) {  // (optimizer, shPtr, enabled) -> dcWordsPtr | null
    ARGS(3);
    wdgs_optimizer* op = (wdgs_optimizer*)get_ptr(env, argv[0]);
    WDGS_OK_OR_THROW(wdgs_optimizer_set_deferred_sh(op, get_ptr(env, argv[1]), (int)get_u32(env, argv[2])));
    void* w = wdgs_optimizer_dc_words(op);
    return w ? make_ptr(env, w) : js_null(env);
}
static napi_value optimizerFlushSH(napi_env env, napi_callback_info info) {  // (optimizer, shPtr)
    ARGS(2);
    WDGS_OK_OR_THROW(wdgs_optimizer_flush_sh((wdgs_optimizer*)get_ptr(env, argv[0]), get_ptr(env, argv[1])));
    return js_undefined(env);
}
static napi_value tiledForwardSetDcSource(napi_env env, napi_callback_info info) {  // (forward, dcWordsPtr | null)
    ARGS(2);
    WDGS_OK_OR_THROW(wdgs_tiled_forward_set_dc_source((wdgs_tiled_forward*)get_ptr(env, argv[0]), get_ptr(env, argv[1])));
    return js_undefined(env);
}
static napi_value optimizerStepF32Range(napi_env env, napi_callback_info info) {  // (optimizer, gaussiansPtr, shPtr, gradF32Ptr, visiblePtr, first, count, rowsPtr | null)
    ARGS(8);
    WDGS_OK_OR_THROW(wdgs_optimizer_step_f32_range((wdgs_optimizer*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_ptr(env, argv[2]), get_ptr(env, argv[3]),
                                                   get_ptr(env, argv[4]), get_u32(env, argv[5]), get_u32(env, argv[6]), get_ptr(env, argv[7])));
    return js_undefined(env);
}
static napi_value optimizerStateChanged(napi_env env, napi_callback_info info) { ARGS(1); WDGS_OK_OR_THROW(wdgs_optimizer_state_changed((wdgs_optimizer*)get_ptr(env, argv[0]))); return js_undefined(env); }
static napi_value storeGradients(napi_env env, napi_callback_info info) {  // (device, numPoints, gradientsPtr, tileCountsPtr, accF32Ptr, visiblePtr)
    ARGS(6);
    WDGS_OK_OR_THROW(wdgs_store_gradients((wdgs_device*)get_ptr(env, argv[0]), get_u32(env, argv[1]), get_ptr(env, argv[2]), get_ptr(env, argv[3]), get_ptr(env, argv[4]),
                                          get_ptr(env, argv[5])));
    return js_undefined(env);
}
static napi_value guardAccumulate(napi_env env, napi_callback_info info) {  // (device, flagPtr, srcPtr, overwrite)
    ARGS(4);
    WDGS_OK_OR_THROW(wdgs_guard_accumulate((wdgs_device*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_ptr(env, argv[2]), (int)get_u32(env, argv[3])));
    return js_undefined(env);
}
static napi_value applyRepackedRows(napi_env env, napi_callback_info info) {  // (device, numPoints, rowsPtr, skipFirst, skipCount, guardPtr | null, gaussiansPtr, shPtr)
    ARGS(8);
    WDGS_OK_OR_THROW(wdgs_apply_repacked_rows((wdgs_device*)get_ptr(env, argv[0]), get_u32(env, argv[1]), get_ptr(env, argv[2]), get_u32(env, argv[3]),
                                              get_u32(env, argv[4]), get_ptr(env, argv[5]), get_ptr(env, argv[6]), get_ptr(env, argv[7])));
    return js_undefined(env);
}
static napi_value commExchangeGradients(napi_env env, napi_callback_info info) {  // (comm, gradF32Ptr, visiblePtr, flagPtr | null, slicePoints)
    ARGS(5);
    WDGS_OK_OR_THROW(wdgs_comm_exchange_gradients((wdgs_comm*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_ptr(env, argv[2]), get_ptr(env, argv[3]), get_u32(env, argv[4])));
    return js_undefined(env);
}
static napi_value commAllgatherRows(napi_env env, napi_callback_info info) {  // (comm, rowsPtr, slicePoints)
    ARGS(3);
    WDGS_OK_OR_THROW(wdgs_comm_allgather_rows((wdgs_comm*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_u32(env, argv[2])));
    return js_undefined(env);
}
static napi_value commBroadcast(napi_env env, napi_callback_info info) {  // (comm, ptr, bytes, root)
    ARGS(4);
    WDGS_OK_OR_THROW(wdgs_comm_broadcast((wdgs_comm*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), (size_t)get_f64(env, argv[2]), (int)get_u32(env, argv[3])));
    return js_undefined(env);
}
static napi_value commInfo(napi_env env, napi_callback_info info) {  // (comm) -> {worldSize, rank}
    ARGS(1);
    napi_value o; napi_create_object(env, &o);
    set_prop(env, o, "worldSize", make_u32(env, (uint32_t)wdgs_comm_world_size((wdgs_comm*)get_ptr(env, argv[0]))));
    set_prop(env, o, "rank", make_u32(env, (uint32_t)wdgs_comm_rank((wdgs_comm*)get_ptr(env, argv[0]))));
    return o;
}


// ---- additions of round 4: the fused single-view step, the view-batched K1 / K17, lane marks, per-kernel timing, RCCL groups --------
static napi_value optimizerStepWithGeometry(napi_env env, napi_callback_info info) {  // (optimizer, backward, cameraPtr, gaussiansPtr, shPtr, tileCountsPtr)
    ARGS(6);
    WDGS_OK_OR_THROW(wdgs_optimizer_step_with_geometry((wdgs_optimizer*)get_ptr(env, argv[0]), (wdgs_tiled_backward*)get_ptr(env, argv[1]), get_ptr(env, argv[2]),
                                                       get_ptr(env, argv[3]), get_ptr(env, argv[4]), get_ptr(env, argv[5])));
    return js_undefined(env);
}
static napi_value optimizerApplyRepackedRows(napi_env env, napi_callback_info info) {  // (optimizer, rowsPtr, skipFirst, skipCount, guardPtr | null, gaussiansPtr, shPtr)
    ARGS(7);
    WDGS_OK_OR_THROW(wdgs_optimizer_apply_repacked_rows((wdgs_optimizer*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_u32(env, argv[2]), get_u32(env, argv[3]),
                                                        get_ptr(env, argv[4]), get_ptr(env, argv[5]), get_ptr(env, argv[6])));
    return js_undefined(env);
}
static napi_value tiledBackwardEncodeRaster(napi_env env, napi_callback_info info) {  // (op, predPtr, targetPtr, resources{...Ptr})
    ARGS(4);
    wdgs_tiled_backward_resources r;
    read_backward_resources(env, argv[3], &r);
    WDGS_OK_OR_THROW(wdgs_tiled_backward_encode_raster((wdgs_tiled_backward*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_ptr(env, argv[2]), &r));
    return js_undefined(env);
}
static napi_value tiledBackwardEncodeGeometry(napi_env env, napi_callback_info info) {
    // (op, cameraPtr, gaussiansPtr, into | null): into = {sums, visible, tileCounts, guard, overflowWord (pointers), first}
    ARGS(4);
    wdgs_view_accumulate a; std::memset(&a, 0, sizeof(a));
    napi_valuetype t; napi_typeof(env, argv[3], &t);
    const bool has = t == napi_object;
    if (has) {
        a.sums = prop_ptr(env, argv[3], "sums"); a.visible = prop_ptr(env, argv[3], "visible"); a.tile_counts = prop_ptr(env, argv[3], "tileCounts");
        a.guard = prop_ptr(env, argv[3], "guard"); a.overflow_word = prop_ptr(env, argv[3], "overflowWord"); a.first = (int)prop_f64(env, argv[3], "first", 0);
    }
    WDGS_OK_OR_THROW(wdgs_tiled_backward_encode_geometry((wdgs_tiled_backward*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_ptr(env, argv[2]), has ? &a : nullptr));
    return js_undefined(env);
}
static napi_value tiledBackwardSetMetricCountsTarget(napi_env env, napi_callback_info info) {  // (op, countsPtr | null)
    ARGS(2);
    WDGS_OK_OR_THROW(wdgs_tiled_backward_set_metric_counts_target((wdgs_tiled_backward*)get_ptr(env, argv[0]), get_ptr(env, argv[1])));
    return js_undefined(env);
}
static napi_value tiledBackwardSetGradientOutput(napi_env env, napi_callback_info info) {  // (op, enabled)
    ARGS(2);
    WDGS_OK_OR_THROW(wdgs_tiled_backward_set_gradient_output((wdgs_tiled_backward*)get_ptr(env, argv[0]), (int)get_u32(env, argv[1])));
    return js_undefined(env);
}
// a JS array of BigInt handles / device pointers -> a C array (at most WDGS_MAX_BATCH_VIEWS entries)
static uint32_t read_ptr_array(napi_env env, napi_value arr, void** out) {
    uint32_t n = 0;
    napi_get_array_length(env, arr, &n);
    if (n > WDGS_MAX_BATCH_VIEWS) n = WDGS_MAX_BATCH_VIEWS + 1;
    for (uint32_t i = 0; i < n && i < WDGS_MAX_BATCH_VIEWS; i++) { napi_value v; napi_get_element(env, arr, i, &v); out[i] = get_ptr(env, v); }
    return n;
}
static napi_value tiledForwardProjectViews(napi_env env, napi_callback_info info) {  // (forwardHandles[], cameraPtrs[], gaussiansPtr, shPtr)
    ARGS(4);
    void* ops[WDGS_MAX_BATCH_VIEWS]; void* cams[WDGS_MAX_BATCH_VIEWS];
    const uint32_t n = read_ptr_array(env, argv[0], ops), m = read_ptr_array(env, argv[1], cams);
    if (n != m || n == 0 || n > WDGS_MAX_BATCH_VIEWS) { napi_throw_range_error(env, nullptr, "projectViews: 1..16 passes and as many cameras"); return nullptr; }
    WDGS_OK_OR_THROW(wdgs_tiled_forward_project_views((wdgs_tiled_forward* const*)ops, (const void* const*)cams, n, get_ptr(env, argv[2]), get_ptr(env, argv[3])));
    return js_undefined(env);
}
static napi_value tiledForwardEncodeProjected(napi_env env, napi_callback_info info) { ARGS(1); WDGS_OK_OR_THROW(wdgs_tiled_forward_encode_projected((wdgs_tiled_forward*)get_ptr(env, argv[0]))); return js_undefined(env); }
static napi_value tiledForwardIsProjected(napi_env env, napi_callback_info info) { ARGS(1); return make_u32(env, (uint32_t)wdgs_tiled_forward_is_projected((const wdgs_tiled_forward*)get_ptr(env, argv[0]))); }
static napi_value tiledBackwardGeometryViews(napi_env env, napi_callback_info info) {
    // (backwardHandles[], cameraPtrs[], tileCountsPtrs[], overflowWordPtrs[], gaussiansPtr, sumsPtr, visiblePtr, guardPtr, writeGradients, continues)
    ARGS(10);
    void* ops[WDGS_MAX_BATCH_VIEWS]; void* cams[WDGS_MAX_BATCH_VIEWS]; void* counts[WDGS_MAX_BATCH_VIEWS]; void* words[WDGS_MAX_BATCH_VIEWS];
    const uint32_t n = read_ptr_array(env, argv[0], ops);
    if (n == 0 || n > WDGS_MAX_BATCH_VIEWS || read_ptr_array(env, argv[1], cams) != n || read_ptr_array(env, argv[2], counts) != n || read_ptr_array(env, argv[3], words) != n) {
        napi_throw_range_error(env, nullptr, "geometryViews: 1..16 passes and as many cameras, tile counts and overflow words");
        return nullptr;
    }
    WDGS_OK_OR_THROW(wdgs_tiled_backward_encode_geometry_views((wdgs_tiled_backward* const*)ops, (const void* const*)cams, (const void* const*)counts, (const void* const*)words, n,
                                                               get_ptr(env, argv[4]), get_ptr(env, argv[5]), get_ptr(env, argv[6]), get_ptr(env, argv[7]),
                                                               (int)get_u32(env, argv[8]), (int)get_u32(env, argv[9])));
    return js_undefined(env);
}
static napi_value deviceLaneMark(napi_env env, napi_callback_info info) {  // (device, lane, mark)
    ARGS(3);
    WDGS_OK_OR_THROW(wdgs_device_lane_mark((wdgs_device*)get_ptr(env, argv[0]), (int)get_u32(env, argv[1]), (int)get_u32(env, argv[2])));
    return js_undefined(env);
}
static napi_value deviceLaneWaitMark(napi_env env, napi_callback_info info) {  // (device, lane, mark)
    ARGS(3);
    WDGS_OK_OR_THROW(wdgs_device_lane_wait_mark((wdgs_device*)get_ptr(env, argv[0]), (int)get_u32(env, argv[1]), (int)get_u32(env, argv[2])));
    return js_undefined(env);
}
static napi_value deviceKernelTimes(napi_env env, napi_callback_info info) {
    // (device, op): 0 profiling off | 1 profiling on | 2 reset | 3 read -> {name: {launches, totalMs}} (after a synchronize)
    ARGS(2);
    wdgs_device* d = (wdgs_device*)get_ptr(env, argv[0]);
    const uint32_t op = get_u32(env, argv[1]);
    if (op <= 1) { WDGS_OK_OR_THROW(wdgs_device_set_profiling(d, (int)op)); return js_undefined(env); }
    if (op == 2) { WDGS_OK_OR_THROW(wdgs_device_reset_kernel_times(d)); return js_undefined(env); }
    static wdgs_kernel_time recs[128];
    uint32_t n = 0;
    WDGS_OK_OR_THROW(wdgs_device_get_kernel_times(d, recs, 128, &n));
    napi_value o; napi_create_object(env, &o);
    for (uint32_t i = 0; i < n && i < 128; i++) {
        napi_value r, ms; napi_create_object(env, &r);
        set_prop(env, r, "launches", make_u32(env, recs[i].launches));
        napi_create_double(env, recs[i].total_ms, &ms); set_prop(env, r, "totalMs", ms);
        char name[49]; std::memcpy(name, recs[i].name, 48); name[48] = 0;
        set_prop(env, o, name, r);
    }
    return o;
}
static napi_value copyBufferToBuffer(napi_env env, napi_callback_info info) {  // (device, dstPtr, srcPtr, byteLength): encoder.copyBufferToBuffer
    ARGS(4);
    WDGS_OK_OR_THROW(wdgs_copy_buffer_to_buffer((wdgs_device*)get_ptr(env, argv[0]), get_ptr(env, argv[1]), get_ptr(env, argv[2]), (size_t)get_f64(env, argv[3])));
    return js_undefined(env);
}
static napi_value commGroup(napi_env env, napi_callback_info info) {  // (end: 0 = wdgs_comm_group_start, 1 = wdgs_comm_group_end)
    ARGS(1);
    if (get_u32(env, argv[0])) WDGS_OK_OR_THROW(wdgs_comm_group_end()); else WDGS_OK_OR_THROW(wdgs_comm_group_start());
    return js_undefined(env);
}

#define EXPORT_FN(name)                                                              \
    do {                                                                             \
        napi_value fn;                                                               \
        napi_create_function(env, #name, NAPI_AUTO_LENGTH, name, nullptr, &fn);      \
        napi_set_named_property(env, exports, #name, fn);                            \
    } while (0)

static napi_value Init(napi_env env, napi_value exports) {
    EXPORT_FN(abiVersion); EXPORT_FN(deviceCreate); EXPORT_FN(deviceDestroy); EXPORT_FN(deviceSynchronize); EXPORT_FN(deviceMemoryInfo);
    EXPORT_FN(bufferCreate); EXPORT_FN(bufferDestroy); EXPORT_FN(copyToDevice); EXPORT_FN(copyToHost);
    EXPORT_FN(tiledForwardCreate); EXPORT_FN(tiledForwardEncode); EXPORT_FN(tiledForwardSetViewport); EXPORT_FN(tiledForwardGetResources); EXPORT_FN(tiledForwardDestroy);
    EXPORT_FN(tiledRasterizerCreate); EXPORT_FN(tiledRasterizerEncode); EXPORT_FN(tiledRasterizerGet); EXPORT_FN(tiledRasterizerDestroy);
    EXPORT_FN(tiledBackwardCreate); EXPORT_FN(tiledBackwardEncode); EXPORT_FN(tiledBackwardGradients); EXPORT_FN(tiledBackwardDestroy);
    EXPORT_FN(optimizerCreate); EXPORT_FN(optimizerStep); EXPORT_FN(optimizerGetIteration); EXPORT_FN(optimizerDestroy);
    EXPORT_FN(tiledForwardSet); EXPORT_FN(tiledForwardCheck); EXPORT_FN(tiledForwardSetLongLists); EXPORT_FN(tiledForwardLongListStats); EXPORT_FN(tiledRasterizerBlit); EXPORT_FN(bufferClear);
    EXPORT_FN(encoderBegin); EXPORT_FN(encoderFinish); EXPORT_FN(queueSubmit); EXPORT_FN(commandBufferDestroy); EXPORT_FN(queueOnSubmittedWorkDone);
    EXPORT_FN(tiledBackwardMetric); EXPORT_FN(tiledBackwardGet); EXPORT_FN(downsampleRGBA8); EXPORT_FN(imageSSE);
    EXPORT_FN(optimizerStateSizes); EXPORT_FN(optimizerCreateWithState); EXPORT_FN(optimizerState); EXPORT_FN(optimizerHyperparameters);
    EXPORT_FN(optimizerAdvanceIteration); EXPORT_FN(optimizerStepF32); EXPORT_FN(accumulateGradients);
    EXPORT_FN(densifyCreate); EXPORT_FN(densifySetConfig); EXPORT_FN(densifyEncodePrepare); EXPORT_FN(densifyReadTotal); EXPORT_FN(densifyEncodeScatter);
    EXPORT_FN(densifyStage);
    EXPORT_FN(densifyDestroy);
    EXPORT_FN(commUniqueId); EXPORT_FN(commCreate); EXPORT_FN(commAllreduceGradients); EXPORT_FN(commAllreduceCounts); EXPORT_FN(commDestroy);
    EXPORT_FN(encoderAbort); EXPORT_FN(hostAlloc); EXPORT_FN(bufferReadAsync); EXPORT_FN(prefixScanner); EXPORT_FN(dynamicSorter);
    EXPORT_FN(optimizerDeferredSH); EXPORT_FN(optimizerFlushSH); EXPORT_FN(tiledForwardSetDcSource);
    EXPORT_FN(optimizerSetGuard); EXPORT_FN(optimizerStepF32Range); EXPORT_FN(optimizerStateChanged); EXPORT_FN(storeGradients); EXPORT_FN(guardAccumulate);
    EXPORT_FN(deviceSelectLane); EXPORT_FN(deviceLaneOrder); EXPORT_FN(queueMark); EXPORT_FN(queueWait); EXPORT_FN(tiledForwardResize); EXPORT_FN(tiledBackwardResize);
    EXPORT_FN(applyRepackedRows); EXPORT_FN(commExchangeGradients); EXPORT_FN(commAllgatherRows); EXPORT_FN(commBroadcast); EXPORT_FN(commInfo);
    EXPORT_FN(optimizerStepWithGeometry); EXPORT_FN(optimizerApplyRepackedRows); EXPORT_FN(tiledBackwardEncodeRaster); EXPORT_FN(tiledBackwardEncodeGeometry);
    EXPORT_FN(tiledBackwardSetGradientOutput); EXPORT_FN(tiledBackwardSetMetricCountsTarget); EXPORT_FN(tiledForwardProjectViews); EXPORT_FN(tiledForwardEncodeProjected); EXPORT_FN(tiledForwardIsProjected);
    EXPORT_FN(tiledBackwardGeometryViews); EXPORT_FN(deviceLaneMark); EXPORT_FN(deviceLaneWaitMark); EXPORT_FN(deviceKernelTimes); EXPORT_FN(commGroup); EXPORT_FN(copyBufferToBuffer);
    return exports;
}
NAPI_MODULE(webdgs_napi, Init)
