// Data-parallel gradient exchange over RCCL: one wdgs_comm per (process, GPU), one all-reduce per global step.
//
// The reference trains one view per step on one GPU (trainer.ts:573); view-sharded data parallelism is this library's own
// extension, defined so that world_size = 1 reduces to the reference: g = sum over ranks of the per-rank fp32 gradient sums,
// visible = sum of per-rank visibility counts, then every rank applies the same Adam step (wdgs_optimizer_step_f32), so the
// replicas stay bit-identical.  Payload: 14 f32 + 1 u32 per Gaussian = 60 B x N, both reductions issued in one RCCL group on
// the device's stream (no host synchronisation; the optimizer kernel that follows is stream-ordered behind them).
//
// RCCL is bound at first use with dlopen("librccl.so.1"): a host that never creates a comm (single GPU, the N-API addon, the
// parity tests) does not load it, and a host that already carries RCCL (PyTorch-ROCm bundles one under the same soname) shares
// that copy instead of pulling a second one into the process.
#include <dlfcn.h>

#include <cstring>

#include "common.h"

namespace {

// The slice of the RCCL ABI used here (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE to ncclCommInitRank).
struct RcclUniqueId { char internal[WDGS_COMM_ID_BYTES]; };
typedef void* RcclComm;
enum { RCCL_SUCCESS = 0 };
enum { RCCL_UINT8 = 1, RCCL_UINT32 = 3, RCCL_FLOAT32 = 7 };  // ncclDataType_t
enum { RCCL_SUM = 0 };                       // ncclRedOp_t

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
    if (g_rccl.handle) return WDGS_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    WDGS_REQUIRE(h, WDGS_E_STATE, "RCCL not available: %s", dlerror());
    RcclApi a;
    a.handle = h;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.ReduceScatter = reinterpret_cast<decltype(a.ReduceScatter)>(dlsym(h, "ncclReduceScatter"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(dlsym(h, "ncclBroadcast"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    WDGS_REQUIRE(a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.ReduceScatter && a.AllGather && a.Broadcast && a.GroupStart && a.GroupEnd && a.GetErrorString, WDGS_E_STATE,
                 "librccl.so.1 lacks an expected symbol");
    g_rccl = a;
    return WDGS_OK;
}

#define WDGS_CHECK_RCCL(expr)                                                                              \
    do {                                                                                                   \
        int _e = (expr);                                                                                   \
        if (_e != RCCL_SUCCESS) {                                                                          \
            wdgs_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_e), __FILE__, __LINE__); \
            return WDGS_E_HIP;                                                                             \
        }                                                                                                  \
    } while (0)

}  // namespace

struct wdgs_comm {
    wdgs_device* dev;
    RcclComm comm;
    int world_size, rank;
};

extern "C" {

int wdgs_comm_get_unique_id(uint8_t id_out[WDGS_COMM_ID_BYTES]) {
    WDGS_REQUIRE(id_out, WDGS_E_INVALID, "wdgs_comm_get_unique_id: null argument");
    WDGS_TRY(rccl_load());
    RcclUniqueId id;
    WDGS_CHECK_RCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, id.internal, WDGS_COMM_ID_BYTES);
    return WDGS_OK;
}

int wdgs_comm_create(wdgs_device* dev, const uint8_t id[WDGS_COMM_ID_BYTES], int world_size, int rank, wdgs_comm** out) {
    WDGS_REQUIRE(dev && id && out, WDGS_E_INVALID, "wdgs_comm_create: null argument");
    WDGS_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, WDGS_E_INVALID, "wdgs_comm_create: rank %d outside world of %d", rank, world_size);
    WDGS_REQUIRE(!dev->capturing, WDGS_E_STATE, "wdgs_comm_create while recording a command buffer");
    WDGS_TRY(rccl_load());
    WDGS_CHECK_HIP(hipSetDevice(dev->ordinal));
    RcclUniqueId uid;
    std::memcpy(uid.internal, id, WDGS_COMM_ID_BYTES);
    RcclComm c = nullptr;
    WDGS_CHECK_RCCL(g_rccl.CommInitRank(&c, world_size, uid, rank));
    *out = new wdgs_comm{dev, c, world_size, rank};
    return WDGS_OK;
}

int wdgs_comm_destroy(wdgs_comm* c) {
    if (!c) return WDGS_OK;
    if (wdgs_device_alive(c->dev) && !c->dev->capturing) (void)wdgs_sync_lanes(c->dev);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return WDGS_OK;
}

int wdgs_comm_world_size(const wdgs_comm* c) { return c ? c->world_size : 0; }
int wdgs_comm_rank(const wdgs_comm* c) { return c ? c->rank : -1; }

int wdgs_comm_allreduce_gradients(wdgs_comm* c, void* grad_f32, void* visible_counts, uint32_t num_points) {
    WDGS_REQUIRE(c && (num_points == 0 || (grad_f32 && visible_counts)), WDGS_E_INVALID, "wdgs_comm_allreduce_gradients: null argument");
    if (num_points == 0) return WDGS_OK;
    WDGS_CHECK_RCCL(g_rccl.GroupStart());
    int e1 = g_rccl.AllReduce(grad_f32, grad_f32, (size_t)num_points * 14, RCCL_FLOAT32, RCCL_SUM, c->comm, c->dev->stream);
    int e2 = g_rccl.AllReduce(visible_counts, visible_counts, (size_t)num_points, RCCL_UINT32, RCCL_SUM, c->comm, c->dev->stream);
    WDGS_CHECK_RCCL(g_rccl.GroupEnd());
    WDGS_CHECK_RCCL(e1);
    WDGS_CHECK_RCCL(e2);
    return WDGS_OK;
}

// The bandwidth-optimal exchange: rank r ends up with the sums of rows [r*slice, (r+1)*slice) only, applies Adam to that
// slice, and publishes the re-packed 32-byte rows with wdgs_comm_allgather_rows.  Both buffers hold world_size*slice_points rows
// (the tail past num_points is zero); RCCL's in-place form recvbuff = sendbuff + rank*recvcount is used.  `flag` (one u32, e.g. a
// tile-overflow word) is summed over all ranks in the same group, so that every rank takes the same "skip this step" decision.
int wdgs_comm_exchange_gradients(wdgs_comm* c, void* grad_f32, void* visible_counts, void* flag, uint32_t slice_points) {
    WDGS_REQUIRE(c && grad_f32 && visible_counts, WDGS_E_INVALID, "wdgs_comm_exchange_gradients: null argument");
    if (slice_points == 0) return WDGS_OK;
    float* g = (float*)grad_f32;
    u32* v = (u32*)visible_counts;
    const size_t r = (size_t)c->rank;
    WDGS_CHECK_RCCL(g_rccl.GroupStart());
    int e1 = g_rccl.ReduceScatter(g, g + r * slice_points * 14, (size_t)slice_points * 14, RCCL_FLOAT32, RCCL_SUM, c->comm, c->dev->stream);
    int e2 = g_rccl.ReduceScatter(v, v + r * slice_points, (size_t)slice_points, RCCL_UINT32, RCCL_SUM, c->comm, c->dev->stream);
    int e3 = flag ? g_rccl.AllReduce(flag, flag, 1, RCCL_UINT32, RCCL_SUM, c->comm, c->dev->stream) : RCCL_SUCCESS;
    WDGS_CHECK_RCCL(g_rccl.GroupEnd());
    WDGS_CHECK_RCCL(e1);
    WDGS_CHECK_RCCL(e2);
    WDGS_CHECK_RCCL(e3);
    return WDGS_OK;
}

// In place: rows[world_size*slice_points][8 u32]; every rank contributes its own slice and receives the others'.
int wdgs_comm_allgather_rows(wdgs_comm* c, void* rows, uint32_t slice_points) {
    WDGS_REQUIRE(c && rows, WDGS_E_INVALID, "wdgs_comm_allgather_rows: null argument");
    if (slice_points == 0) return WDGS_OK;
    u32* p = (u32*)rows;
    WDGS_CHECK_RCCL(g_rccl.AllGather(p + (size_t)c->rank * slice_points * 8, p, (size_t)slice_points * 8, RCCL_UINT32, c->comm, c->dev->stream));
    return WDGS_OK;
}

// In place broadcast of `bytes` bytes at `ptr` from `root` (the slice-owned optimizer state is gathered with one call per owner
// and array before a densify rebuild or an export).
int wdgs_comm_broadcast(wdgs_comm* c, void* ptr, size_t bytes, int root) {
    WDGS_REQUIRE(c && (bytes == 0 || ptr), WDGS_E_INVALID, "wdgs_comm_broadcast: null argument");
    WDGS_REQUIRE(root >= 0 && root < c->world_size, WDGS_E_INVALID, "wdgs_comm_broadcast: root %d outside world of %d", root, c->world_size);
    if (bytes == 0) return WDGS_OK;
    WDGS_CHECK_RCCL(g_rccl.Broadcast(ptr, ptr, bytes, RCCL_UINT8, root, c->comm, c->dev->stream));
    return WDGS_OK;
}
int wdgs_comm_group_start(void) {
    WDGS_TRY(rccl_load());
    WDGS_CHECK_RCCL(g_rccl.GroupStart());
    return WDGS_OK;
}
int wdgs_comm_group_end(void) {
    WDGS_TRY(rccl_load());
    WDGS_CHECK_RCCL(g_rccl.GroupEnd());
    return WDGS_OK;
}

int wdgs_comm_allreduce_counts(wdgs_comm* c, void* counts_u32, uint32_t count) {
    WDGS_REQUIRE(c && (count == 0 || counts_u32), WDGS_E_INVALID, "wdgs_comm_allreduce_counts: null argument");
    if (count == 0) return WDGS_OK;
    WDGS_CHECK_RCCL(g_rccl.AllReduce(counts_u32, counts_u32, (size_t)count, RCCL_UINT32, RCCL_SUM, c->comm, c->dev->stream));
    return WDGS_OK;
}

}  // extern "C"
