// ry_comm.cpp -- RCCL bound at run time (dlopen: libry355.so carries no link-time dependency on it) for the one collective of the path:
// the broadcast of each predictor's flat weight blob from rank 0 at start-up (BASELINE.json north_star: "RCCL broadcast of weights over
// xGMI and no cross-chunk collectives"), plus a barrier and a max for the bench, and plain device buffers (ry_dev_*) for callers that
// have no tensor library to hold the blob.
#include "ry_host.h"

extern "C" {

// ---- device buffers for callers without a tensor library (weights for ry_comm_bcast_weights / ry_net_create) ----
int ry_dev_alloc(ry_ctx* ctx, size_t n_floats, float** out) {
    if (!ctx || !out) return fail(RY_EINVAL, "null argument");
    RT_TRY(rt::set_device(ctx->device));
    RY_TRY(ctx->alloc(out, n_floats));
    return RY_OK;
}
int ry_dev_free(ry_ctx* ctx, float* p) {
    if (!ctx) return fail(RY_EINVAL, "null argument");
    RT_TRY(rt::set_device(ctx->device));
    RT_TRY(rt::stream_sync(ctx->stream));
    if (p) RT_TRY(rt::dfree(p));
    return RY_OK;
}
int ry_dev_upload(ry_ctx* ctx, float* dst_dev, const float* src_host, size_t n_floats) {
    if (!ctx || !dst_dev || !src_host) return fail(RY_EINVAL, "null argument");
    RT_TRY(rt::set_device(ctx->device));
    RT_TRY(rt::h2d(dst_dev, src_host, n_floats * sizeof(float), ctx->stream));
    RT_TRY(rt::stream_sync(ctx->stream));
    return RY_OK;
}
int ry_dev_download(ry_ctx* ctx, float* dst_host, const float* src_dev, size_t n_floats) {
    if (!ctx || !dst_host || !src_dev) return fail(RY_EINVAL, "null argument");
    RT_TRY(rt::set_device(ctx->device));
    RT_TRY(rt::d2h(dst_host, src_dev, n_floats * sizeof(float), ctx->stream));
    RT_TRY(rt::stream_sync(ctx->stream));
    return RY_OK;
}

}  // extern "C"

// ---- RCCL, bound at run time (dlopen: libry355.so carries no link-time dependency on it; a process that already loaded an RCCL
// -- torch's -- gets that one back by soname).  Only what chunk parallelism needs: one broadcast of each weight blob at start-up
// (SURVEY.md 8(e)), and a max / barrier for timing.  No collective ever runs in the steady state.
#ifndef RY_HOST_EMU
#include <dlfcn.h>
struct RyNcclId { char internal[128]; };
typedef struct ncclComm* RyNcclComm;
struct RyNccl {
    void* h = nullptr;
    int (*GetUniqueId)(RyNcclId*) = nullptr;
    int (*CommInitRank)(RyNcclComm*, int, RyNcclId, int) = nullptr;
    int (*CommDestroy)(RyNcclComm) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int /*dtype*/, int /*root*/, RyNcclComm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int /*dtype*/, int /*op*/, RyNcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
static RyNccl g_nccl;
static int load_rccl() {
    if (g_nccl.h) return RY_OK;
    const char* names[] = {getenv("RY_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break; }
    if (!h) return fail(RY_ESTATE, "librccl.so not found (set RY_RCCL_LIB): %s", dlerror());
#define RY_SYM(field, name) do { *(void**)(&g_nccl.field) = dlsym(h, name); if (!g_nccl.field) return fail(RY_ESTATE, "librccl lacks %s", name); } while (0)
    RY_SYM(GetUniqueId, "ncclGetUniqueId"); RY_SYM(CommInitRank, "ncclCommInitRank"); RY_SYM(CommDestroy, "ncclCommDestroy");
    RY_SYM(Broadcast, "ncclBroadcast"); RY_SYM(AllReduce, "ncclAllReduce"); RY_SYM(GetErrorString, "ncclGetErrorString");
#undef RY_SYM
    g_nccl.h = h;
    return RY_OK;
}
#define NCCL_TRY(expr) do { int e__ = (expr); if (e__ != 0) return fail(RY_EHIP, "%s failed: %s", #expr, g_nccl.GetErrorString(e__)); } while (0)
enum { RY_NCCL_FLOAT32 = 7, RY_NCCL_FLOAT64 = 8, RY_NCCL_MAX = 2 };     // ncclDataType_t / ncclRedOp_t values (nccl.h)
#endif

struct ry_comm {
    ry_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    double* d_scalar = nullptr;
#ifndef RY_HOST_EMU
    RyNcclComm comm = nullptr;
#endif
};

extern "C" {

int ry_comm_unique_id(void* id128) {
    if (!id128) return fail(RY_EINVAL, "null argument");
#ifdef RY_HOST_EMU
    memset(id128, 0, 128);
    return RY_OK;
#else
    RY_TRY(load_rccl());
    RyNcclId id;
    NCCL_TRY(g_nccl.GetUniqueId(&id));
    memcpy(id128, &id, 128);
    return RY_OK;
#endif
}

int ry_comm_init(ry_ctx* ctx, const void* id128, int rank, int world, ry_comm** out) {
    if (!ctx || !id128 || !out) return fail(RY_EINVAL, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(RY_EINVAL, "bad rank %d of %d", rank, world);
    RT_TRY(rt::set_device(ctx->device));
    std::unique_ptr<ry_comm> c(new ry_comm());
    c->ctx = ctx; c->rank = rank; c->world = world;
    float* q = nullptr;
    RY_TRY(ctx->alloc(&q, 4));
    c->d_scalar = reinterpret_cast<double*>(q);
#ifdef RY_HOST_EMU
    if (world != 1) return fail(RY_ESTATE, "the emulator build has no RCCL: world must be 1");
#else
    RY_TRY(load_rccl());
    RyNcclId id;
    memcpy(&id, id128, 128);
    NCCL_TRY(g_nccl.CommInitRank(&c->comm, world, id, rank));
#endif
    *out = c.release();
    return RY_OK;
}

void ry_comm_destroy(ry_comm* c) {
    if (!c) return;
    rt::set_device(c->ctx->device);
    rt::stream_sync(c->ctx->stream);
#ifndef RY_HOST_EMU
    if (c->comm) g_nccl.CommDestroy(c->comm);
#endif
    if (c->d_scalar) rt::dfree(c->d_scalar);
    delete c;
}

// in-place broadcast of a flat weight blob (device memory on this context's GPU) from `root`; returns when it has arrived
int ry_comm_bcast_weights(ry_comm* c, float* blob_dev, size_t n_floats, int root) {
    if (!c || !blob_dev) return fail(RY_EINVAL, "null argument");
    if (root < 0 || root >= c->world) return fail(RY_EINVAL, "bad root %d of %d", root, c->world);
    RT_TRY(rt::set_device(c->ctx->device));
#ifndef RY_HOST_EMU
    NCCL_TRY(g_nccl.Broadcast(blob_dev, blob_dev, n_floats, RY_NCCL_FLOAT32, root, c->comm, c->ctx->stream));
#endif
    RT_TRY(rt::stream_sync(c->ctx->stream));
    return RY_OK;
}

// *value = max over the ranks (an all-reduce of one double: doubles as the barrier of the timed region)
int ry_comm_allreduce_max(ry_comm* c, double* value) {
    if (!c || !value) return fail(RY_EINVAL, "null argument");
    RT_TRY(rt::set_device(c->ctx->device));
    RT_TRY(rt::h2d(c->d_scalar, value, sizeof(double), c->ctx->stream));
#ifndef RY_HOST_EMU
    NCCL_TRY(g_nccl.AllReduce(c->d_scalar, c->d_scalar, 1, RY_NCCL_FLOAT64, RY_NCCL_MAX, c->comm, c->ctx->stream));
#endif
    RT_TRY(rt::d2h(value, c->d_scalar, sizeof(double), c->ctx->stream));
    RT_TRY(rt::stream_sync(c->ctx->stream));
    return RY_OK;
}

int ry_comm_barrier(ry_comm* c) {
    double one = 1.0;
    return ry_comm_allreduce_max(c, &one);
}

}  // extern "C"
