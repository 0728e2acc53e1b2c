// N-API addon over the C ABI of libwebdgs_hip.so (include/webdgs.h) -- the binding a WebDGS maintainer would add so that
// the TypeScript host (bindings/ts/webdgs_hip.ts, shaped like the reference's src/renderers/*.ts) drives the HIP kernels.
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

#define EXPORT_FN(name)                                                              \
    do {                                                                             \
        napi_value fn;                                                               \
        napi_create_function(env, #name, NAPI_AUTO_LENGTH, name, nullptr, &fn);      \
        napi_set_named_property(env, exports, #name, fn);                            \
    } while (0)

static napi_value Init(napi_env env, napi_value exports) {
    EXPORT_FN(abiVersion); EXPORT_FN(deviceCreate); EXPORT_FN(deviceDestroy); EXPORT_FN(deviceSynchronize);
    EXPORT_FN(bufferCreate); EXPORT_FN(bufferDestroy); EXPORT_FN(copyToDevice); EXPORT_FN(copyToHost);
    EXPORT_FN(tiledForwardCreate); EXPORT_FN(tiledForwardEncode); EXPORT_FN(tiledForwardSetViewport); EXPORT_FN(tiledForwardGetResources); EXPORT_FN(tiledForwardDestroy);
    EXPORT_FN(tiledRasterizerCreate); EXPORT_FN(tiledRasterizerEncode); EXPORT_FN(tiledRasterizerGet); EXPORT_FN(tiledRasterizerDestroy);
    EXPORT_FN(tiledBackwardCreate); EXPORT_FN(tiledBackwardEncode); EXPORT_FN(tiledBackwardGradients); EXPORT_FN(tiledBackwardDestroy);
    EXPORT_FN(optimizerCreate); EXPORT_FN(optimizerStep); EXPORT_FN(optimizerGetIteration); EXPORT_FN(optimizerDestroy);
    return exports;
}
NAPI_MODULE(webdgs_napi, Init)
