// ry_host.h -- what the translation units of libry355.so share on the host side: the runtime shim (HIP in the product, malloc / memcpy
// under the test emulator), status codes + message, and the executor's data structures (context, arena, layer, launch plan, predictor).
//   ry_plan.cpp  topology, filter re-layout, the launch planner and its switches (no device code; declarations shared with the next two: ry_plan.h)
//   ry_exec.cpp  the kernels (ry_kernels.h), their launchers, the captured forward, autotune, profiling, the single operators
//   ry_net.cpp   the predictor C ABI: context / predictor lifetime, dtype modes, forward / convert entry points, debug hooks
//   ry_vc.cpp    the window call (ring slots, lanes, silence gate, batch) = VoiceChanger.convert_from_acoustic_feature on the device
//   ry_comm.cpp  RCCL bound at run time (weight broadcast, barrier, max) and plain device buffers for callers without a tensor library
#pragma once
#include "ry_dev.h"

#include "../../include/ry355.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

// ------------------------------------------------------------------------------------------------
// runtime shim: HIP in the product, malloc/memcpy under the test emulator
// ------------------------------------------------------------------------------------------------
namespace rt {
#ifdef RY_HOST_EMU
typedef int err_t;
static const char* err_str(err_t) { return "emu"; }
static err_t set_device(int) { return 0; }
static err_t device_count(int* n) { *n = 1; return 0; }
static err_t stream_create(ry_stream_t* s) { *s = nullptr; return 0; }
static err_t stream_destroy(ry_stream_t) { return 0; }
static err_t stream_sync(ry_stream_t) { return 0; }
static err_t dmalloc(void** p, size_t bytes) { *p = aligned_alloc(256, (bytes + 255) / 256 * 256); return *p ? 0 : 1; }
static err_t dfree(void* p) { free(p); return 0; }
static err_t h2d(void* d, const void* h, size_t n, ry_stream_t) { memcpy(d, h, n); return 0; }
static err_t d2h(void* h, const void* d, size_t n, ry_stream_t) { memcpy(h, d, n); return 0; }
static err_t d2d(void* d, const void* s, size_t n, ry_stream_t) { memcpy(d, s, n); return 0; }
static err_t dmemset(void* d, int v, size_t n, ry_stream_t) { memset(d, v, n); return 0; }
static err_t last_error() { return 0; }
struct Event { double t; };
static err_t event_create(Event*) { return 0; }
static err_t event_destroy(Event&) { return 0; }
static err_t event_record(Event&, ry_stream_t) { return 0; }
static err_t event_sync(Event&) { return 0; }
static err_t event_elapsed(float* ms, Event&, Event&) { *ms = 0.f; return 0; }
static err_t stream_wait_event(ry_stream_t, Event&) { return 0; }
static err_t hmalloc(void** p, size_t bytes) { *p = aligned_alloc(256, (bytes + 255) / 256 * 256); return *p ? 0 : 1; }
static err_t hfree(void* p) { free(p); return 0; }
static err_t event_create_fast(Event*) { return 0; }
#else
typedef hipError_t err_t;
static const char* err_str(err_t e) { return hipGetErrorString(e); }
static err_t set_device(int d) { return hipSetDevice(d); }
static err_t device_count(int* n) { return hipGetDeviceCount(n); }
static err_t stream_create(ry_stream_t* s) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
static err_t stream_destroy(ry_stream_t s) { return hipStreamDestroy(s); }
static err_t stream_sync(ry_stream_t s) { return hipStreamSynchronize(s); }
static err_t dmalloc(void** p, size_t bytes) { return hipMalloc(p, bytes ? bytes : 256); }
static err_t dfree(void* p) { return hipFree(p); }
static err_t h2d(void* d, const void* h, size_t n, ry_stream_t s) { return hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s); }
static err_t d2h(void* h, const void* d, size_t n, ry_stream_t s) { return hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s); }
static err_t d2d(void* d, const void* s_, size_t n, ry_stream_t s) { return hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, s); }
static err_t dmemset(void* d, int v, size_t n, ry_stream_t s) { return hipMemsetAsync(d, v, n, s); }
static err_t last_error() { return hipGetLastError(); }
typedef hipEvent_t Event;
static err_t event_create(Event* e) { return hipEventCreate(e); }
static err_t event_destroy(Event& e) { return hipEventDestroy(e); }
static err_t event_record(Event& e, ry_stream_t s) { return hipEventRecord(e, s); }
static err_t event_sync(Event& e) { return hipEventSynchronize(e); }
static err_t event_elapsed(float* ms, Event& a, Event& b) { return hipEventElapsedTime(ms, a, b); }
static err_t stream_wait_event(ry_stream_t s, Event& e) { return hipStreamWaitEvent(s, e, 0); }
// pinned: async copies really are asynchronous
static err_t hmalloc(void** p, size_t bytes) { return hipHostMalloc(p, bytes ? bytes : 256, hipHostMallocDefault); }
static err_t hfree(void* p) { return hipHostFree(p); }
static err_t event_create_fast(Event* e) { return hipEventCreateWithFlags(e, hipEventDisableTiming); }
#endif
}  // namespace rt

// the message behind the last negative status code of this thread (ry_last_error); ONE per library, defined in ry_net.cpp
extern thread_local std::string g_ry_err;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_ry_err = buf;
    return code;
}

#define RT_TRY(expr)                                                                               \
    do {                                                                                           \
        rt::err_t e__ = (expr);                                                                    \
        if (e__ != 0) return fail(RY_EHIP, "%s failed: %s (%s:%d)", #expr, rt::err_str(e__), __FILE__, __LINE__); \
    } while (0)
#define RY_TRY(expr)                 \
    do {                             \
        int r__ = (expr);            \
        if (r__ != RY_OK) return r__; \
    } while (0)

// ------------------------------------------------------------------------------------------------
struct ry_net;
struct ry_ctx {
    std::vector<ry_net*> nets;
    int device = 0;
    ry_stream_t stream = nullptr;
    rt::Event t0, t1;
    bool timers = false;
    std::vector<void*> owned;            // context-lifetime device allocations

    int alloc(float** p, size_t nfloats) {
        void* q = nullptr;
        rt::err_t e = rt::dmalloc(&q, nfloats * sizeof(float));
        if (e != 0) return fail(RY_ENOMEM, "device allocation of %zu bytes failed: %s", nfloats * sizeof(float), rt::err_str(e));
        *p = (float*)q;
        return RY_OK;
    }
};

// arena of device buffers freed together
struct Arena {
    std::vector<void*> bufs;
    int alloc(float** p, size_t nfloats) {
        void* q = nullptr;
        rt::err_t e = rt::dmalloc(&q, nfloats * sizeof(float));
        if (e != 0) return fail(RY_ENOMEM, "device allocation of %zu bytes failed: %s", nfloats * sizeof(float), rt::err_str(e));
        bufs.push_back(q);
        *p = (float*)q;
        return RY_OK;
    }
    void release() {
        for (void* q : bufs) rt::dfree(q);
        bufs.clear();
    }
    // filter layouts built on first use (key = layer index): shared by the clones of a predictor like everything in this arena
    std::map<int, float*> lazy;
    void free_one(void* q) {               // a buffer that is being replaced by a larger one
        for (size_t i = 0; i < bufs.size(); ++i)
            if (bufs[i] == q) { rt::dfree(q); bufs.erase(bufs.begin() + (long)i); return; }
    }
    ~Arena() { release(); }
};

// ------------------------------------------------------------------------------------------------
// one layer of a predictor (topology + device parameters), its launch plan, a plan of the whole forward, the predictor
// ------------------------------------------------------------------------------------------------
struct Layer {
    char name[24];
    bool deconv = false;
    bool bn = false;
    int k = 1, stride = 1, pad = 0, dil = 1;
    int cin_a = 0, cin_b = 0, cout = 0;
    int src_a = -1, src_b = -2;          // producer layer index; -1 = network input; -2 = none
    int act = RY_ACT_NONE;               // activation applied to this layer's output
    // device parameters
    float* scale = nullptr;
    float* shift = nullptr;
    float* w1d = nullptr;                // stage-1 [Ctot][N][4] (weight-streaming kernel ry_conv1d_ws: lanes = output channels)
    float* w1os = nullptr;               // stage-1 [N][Ctot][4] (output-stationary kernel ry_c1d_os: lanes = input channels)
    float* wig = nullptr;                // stage-2 implicit-GEMM blocks [phase][N/64][tap][Ctot/32][fragment order], see wig_inblock()
    float* wdir = nullptr;               // stage-2 direct [phase][tap][Ctot][N]
    // stage-2 implicit-GEMM bf16 blocks [phase][N/64][tap][Ctot/64][fragment order], see wig16_inblock() (ry_net_set_dtype)
    float* wig16 = nullptr;
    // stage-2 output-stationary kernel ry_c2d_os: [phase][N/4][tap][Ctot/64][lane][4], see relayout_c2d_os() (the weight-streaming layers only)
    float* w2os = nullptr;
    // stage-2 Winograd F(2x2, 2x2) filters [phase][N/64][slice of 8 channels][position 9][n/32][lane][4], see relayout_wino() (op-level calls;
    // predictors: Arena::lazy)
    float* wwin = nullptr;
    // split-bf16 blocks [phase][N/64][tap][3 Ctot/64][fragment order]: K runs over [hi | hi | lo] per source, see build_wigx3()
    float* wigx3 = nullptr;
    int cin() const { return cin_a + cin_b; }
};

enum { PATH_IGEMM = 1, PATH_DIRECT = 2, PATH_FIRST = 3, PATH_LAST = 4, PATH_IGEMM_BF16 = 5,
       PATH_IGEMM_X3 = 6,     // op-level selector only (ry_conv2d): runs as PATH_IGEMM_BF16 with LayerPlan::x3
       PATH_OS2D = 7,         // output-stationary weight-streaming layer (ry_c2d_os): one node, no slabs
       PATH_WINO = 8 };       // k4 s2 p1 layer in Winograd F(2x2, 2x2) form on the fp32 matrix pipe (ry_wino_ldsdma): 9 / 16 of the direct form's MFMA work
// (2 and 7 were the 256-row tiles of the register-staged kernel, removed in round 3)
enum { TILE_128x128 = 1, TILE_64x128 = 3, TILE_32x128 = 4, TILE_128x64 = 5, TILE_96x128 = 6 };

struct LayerPlan {
    // geometry
    int Hi = 1, Wi = 1, Ho = 1, Wo = 1;       // stage-1: H = 1, W = length
    // stage-1
    int splits = 1;
    long long slab_stride = 0;
    float* raw = nullptr;                     // [splits][B*Lo][N] raw sums
    // output-stationary stage-1 kernel (ry_c1d_os): output channels / rows per workgroup slice, ci waves per position group
    int os_cb = 0, os_tp = 0, os_kt = 0;
    // stage-2
    int path = 0, tile = 0;
    bool any_m_patch = false;                 // op-level calls (tests): take the input-patch variants whatever the row count
    // PATH_OS2D: tile of 4 mt4 pixels x 4 nt4 channels per workgroup, waves that share the K axis, units in flight per wave
    int os2_mt4 = 0, os2_nt4 = 0, os2_waves = 0, os2_depth = 0;
    // PATH_WINO: workgroup shape (1 = 2 x 2 waves, one 8-channel slice per iteration; 2 = 4 x 2 waves, two slices) and M-blocks per tile row
    int wino_cfg = 0, wino_mbw = 0;
    int kg = 1;                               // K groups inside a workgroup (LDS-DMA implicit GEMM): 2 = split-K summed through the LDS
    float* out = nullptr;                     // NHWC activation, fp32
    unsigned short* out16 = nullptr;          // NHWC activation, bf16 copy for consumers on the bf16 path (bf16 / split-bf16 mode only)
    bool w32 = true, w16 = false;             // which copies the producer writes
    bool x3 = false;                          // PATH_IGEMM_BF16 in split-bf16 form: sources are [pixel][hi | lo], K = [hi | lo | hi] x filters [hi | hi | lo]
    bool o16x3 = false;                       // format of the out16 copy this layer writes: plain bf16 or split [hi | lo]
    float* slabs = nullptr;
    // > 0 (set per enqueue): output rows hole_lo .. hole_lo + hole_n - 1 equal row hole_lo - 1 (padding behind the real frames): not computed, copied
    int hole_lo = 0, hole_n = 0;
    // > 0 (set per enqueue, one window): the layer runs on the first crop_hi input rows only -- the rows behind them feed nothing but
    // output rows the convert wrapper throws away (dead padding rows, see enqueue_forward)
    int crop_hi = 0;
    int crop_lo = 0;                          // ... starting at this input row (> 0 when the caller discards the leading frames of the window too)
    int last_rows = 0, last_cols = 0, last_exp = 0;   // PATH_LAST: fused exp / edge-pad / crop
    int last_row0 = 0, last_out_rows = 0;     // PATH_LAST: first output row computed, rows per image of the caller's block (0: last_rows)
    double flops = 0, bytes = 0;
    double flops_exec = 0;                    // MFMA flops the launch executes when they differ from the algorithmic ones (Winograd: 9 / 16); 0 = the same
};

struct Plan {
    int B = 0, T = 0;
    int mode = 0;                             // 0 = forward, 1 = convert wrapper
    int n_frames = 0;
    // convert mode, stage 2: the caller throws away this many leading / trailing frames of every window (ry_sr_convert_rows)
    int disc_front = 0, disc_back = 0;
    Arena arena;
    std::vector<LayerPlan> lp;
    float* user_in = nullptr;                 // staging of the caller's input
    float* user_out = nullptr;
    float* x_in = nullptr;                    // padded predictor input
    size_t user_in_floats = 0, user_out_floats = 0;
    bool s1_os = false;                       // stage-1 plan runs the output-stationary kernels (dense activated buffers in lp.out, no slabs)
    bool s1_padfuse = false;                  // ... and its first layer takes the caller's block and pads it itself (no ry_pad_min_rows node)
    const float* cur_in = nullptr;            // where the forward reads the caller's block (staging, or the caller's device buffer)
    float* cur_out = nullptr;                 // where it writes the result
#ifndef RY_HOST_EMU
    // one captured graph per (input, output) address pair the plan has been run with: host callers (plan staging), device callers
    // and the ring slots of ry_vc each keep their own, so switching between them neither re-captures nor destroys an exec that
    // may still be in flight
    struct GraphSlot { const float* in; float* out; hipGraphExec_t gexec; bool tried; long long graph_n, last_n; unsigned long long used; };
    std::vector<GraphSlot> gslots;
    unsigned long long gclock = 0;
    ~Plan() { for (GraphSlot& g : gslots) if (g.gexec) hipGraphExecDestroy(g.gexec); }
#endif
};

struct KernelRec {           // filled by the launch helpers when profiling
    std::string name, layer;
    double flops, bytes, flops_exec;
    int grid[3];
};

struct ry_net {
    int dtype = 0;                           // 0 = fp32 MFMA, 1 = bf16 operands (fp32 accumulate) for the stage-2 implicit-GEMM layers,
                                             // 2 = split-bf16 (hi*hi + lo*hi + hi*lo on the bf16 pipe, fp32 accumulate: fp32-class results)
    ry_ctx* ctx = nullptr;
    ry_stream_t stream = nullptr;            // each predictor enqueues on its own stream: stage-1 of one window overlaps stage-2 of another
    rt::Event done;                          // (spare: ry_sync / ry_timer_stop join the predictor streams on the host)
    bool has_done = false;
    ry_net_desc desc;
    std::vector<Layer> layers;
    std::shared_ptr<Arena> weights = std::make_shared<Arena>();   // filters, scale / shift: shared by the clones of a predictor (ry_net_clone)
    std::map<std::tuple<int, int, int, int>, std::unique_ptr<Plan>> plans;
    bool use_graph = true;
    // profiling hook
    std::vector<KernelRec>* rec = nullptr;
    std::vector<std::pair<rt::Event, rt::Event>>* rec_events = nullptr;
};

static int upload(Arena& a, ry_ctx* ctx, const std::vector<float>& h, float** d) {
    RY_TRY(a.alloc(d, h.size()));
    RT_TRY(rt::h2d(*d, h.data(), h.size() * sizeof(float), ctx->stream));
    RT_TRY(rt::stream_sync(ctx->stream));      // h is a temporary
    return RY_OK;
}

// ---- launch helpers -----------------------------------------------------------------------------
struct Launcher {
    ry_net* net;
    ry_ctx* ctx;
    ry_stream_t stream;
    std::vector<KernelRec>* rec;
    std::vector<std::pair<rt::Event, rt::Event>>* ev;

    int begin(const char* name, const char* layer, double flops, double bytes, dim3 grid, double flops_exec = 0) {
        if (rec) {
            KernelRec r; r.name = name; r.layer = layer; r.flops = flops; r.bytes = bytes; r.flops_exec = flops_exec > 0 ? flops_exec : flops;
            r.grid[0] = (int)grid.x; r.grid[1] = (int)grid.y; r.grid[2] = (int)grid.z;
            rec->push_back(r);
        }
        if (ev) {
            std::pair<rt::Event, rt::Event> pr;
            RT_TRY(rt::event_create(&pr.first)); RT_TRY(rt::event_create(&pr.second));
            ev->push_back(pr);
            RT_TRY(rt::event_record(ev->back().first, stream));
        }
        return RY_OK;
    }
    int end() {
        if (ev) RT_TRY(rt::event_record(ev->back().second, stream));
        RT_TRY(rt::last_error());
        return RY_OK;
    }
};

