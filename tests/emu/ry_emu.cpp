// ry_emu.cpp -- fiber scheduler of the host-side SIMT emulator.  TEST INFRASTRUCTURE ONLY.
#include "ry_emu.h"

#if !defined(__x86_64__)
#include <ucontext.h>
#endif

#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

namespace ry_emu {

thread_local dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

namespace {
constexpr size_t kStack = 256 * 1024;
constexpr int kMaxWaves = 16;

#if defined(__x86_64__)
// A context switch without system calls: glibc's swapcontext saves and restores the signal mask with two rt_sigprocmask calls per switch,
// and a 64-lane wave exchanging operands for ONE emulated MFMA switches ~250 times -- the system calls were most of the CPU suite's time.
// ry_ctx_switch(&save_sp, new_sp): push the callee-saved registers + mxcsr / x87 control word, store the stack pointer, load the other one,
// pop, return.  A fresh fiber's stack is prepared so that the first switch to it "returns" into fiber_entry.
extern "C" void ry_ctx_switch(void** save_sp, void* new_sp);
asm(R"(
    .text
    .globl ry_ctx_switch
    .type ry_ctx_switch,@function
ry_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size ry_ctx_switch, .-ry_ctx_switch
)");
struct Context { void* sp = nullptr; };
#else
struct Context { ucontext_t uc; };
#endif

struct Fiber {
    Context ctx;
    char* stack = nullptr;
    bool done = false;
};

struct State {
    std::vector<Fiber> fibers;
    Context sched;
    int cur = 0;
    int nthreads = 0;
    int live = 0;                      // fibers that have not returned yet: a barrier waits for these only (like the hardware)
    int bar_arrived = 0, bar_gen = 0;
    int wave_arrived[kMaxWaves] = {0}, wave_gen[kMaxWaves] = {0};
    float wa[kMaxWaves][64], wb[kMaxWaves][64];
    unsigned short wa16[kMaxWaves][64][8], wb16[kMaxWaves][64][8];
    const std::function<void()>* body = nullptr;
};
thread_local State* S = nullptr;

#if defined(__x86_64__)
void switch_ctx(Context& from, Context& to) { ry_ctx_switch(&from.sp, to.sp); }
#else
void switch_ctx(Context& from, Context& to) { swapcontext(&from.uc, &to.uc); }
#endif

void yield_to_sched() { switch_ctx(S->fibers[S->cur].ctx, S->sched); }

void trampoline() {
    (*S->body)();
    S->fibers[S->cur].done = true;
    // a thread that returns no longer takes part in barriers: release one that was only waiting for it
    if (--S->live > 0 && S->bar_arrived == S->live) { S->bar_arrived = 0; ++S->bar_gen; }
    // back to the scheduler (x86-64: explicitly, for good; else through uc_link)
#if defined(__x86_64__)
    for (;;) yield_to_sched();
#endif
}

#if defined(__x86_64__)
void fiber_entry() { trampoline(); }

void prepare(Fiber& f, Context&) {
    // [A] = return address (fiber_entry), below it six zeroed callee-saved registers and the default mxcsr / x87 control word;
    // A is 16-byte aligned, so that fiber_entry starts with the stack alignment of a called function
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    uint64_t* A = (uint64_t*)(top - 16);
    A[0] = (uint64_t)(uintptr_t)&fiber_entry;
    for (int i = 1; i <= 6; ++i) A[-i] = 0;
    uint32_t* cw = (uint32_t*)(A - 7);
    cw[0] = 0x1F80u;                   // mxcsr: all exceptions masked, round to nearest
    cw[1] = 0x037Fu;                   // x87 control word
    f.ctx.sp = (void*)(A - 7);
}
#else
void prepare(Fiber& f, Context& sched) {
    getcontext(&f.ctx.uc);
    f.ctx.uc.uc_stack.ss_sp = f.stack;
    f.ctx.uc.uc_stack.ss_size = kStack;
    f.ctx.uc.uc_link = &sched.uc;
    makecontext(&f.ctx.uc, (void (*)())trampoline, 0);
}
#endif

int wave_size(int w) {
    int lo = w * 64, hi = lo + 64;
    if (hi > S->nthreads) hi = S->nthreads;
    return hi - lo;
}

void sync_wave() {
    const int w = S->cur >> 6;
    const int g = S->wave_gen[w];
    if (++S->wave_arrived[w] == wave_size(w)) {
        S->wave_arrived[w] = 0;
        ++S->wave_gen[w];
    } else {
        while (S->wave_gen[w] == g) yield_to_sched();
    }
}

void run_block(State& st, dim3 bidx, dim3 grid, dim3 block) {
    S = &st;
    const int n = (int)block.x;
    st.nthreads = n;
    st.live = n;
    st.bar_arrived = 0;
    for (int w = 0; w < kMaxWaves; ++w) st.wave_arrived[w] = 0;
    if ((int)st.fibers.size() < n) {
        st.fibers.resize(n);
    }
    for (int i = 0; i < n; ++i) {
        Fiber& f = st.fibers[i];
        if (!f.stack) f.stack = (char*)malloc(kStack);
        f.done = false;
        prepare(f, st.sched);
    }
    g_blockIdx = bidx;
    g_gridDim = grid;
    g_blockDim = block;
    int remaining = n;
    while (remaining > 0) {
        for (int i = 0; i < n; ++i) {
            Fiber& f = st.fibers[i];
            if (f.done) continue;
            st.cur = i;
            g_threadIdx = dim3((unsigned)i, 0, 0);
            switch_ctx(st.sched, f.ctx);
            g_threadIdx = dim3((unsigned)i, 0, 0);
            if (f.done) --remaining;
        }
    }
}
}  // namespace

void wave_sync() { sync_wave(); }

void sync_block() {
    const int g = S->bar_gen;
    if (++S->bar_arrived == S->live) {
        S->bar_arrived = 0;
        ++S->bar_gen;
    } else {
        while (S->bar_gen == g) {
            yield_to_sched();
            g_threadIdx = dim3((unsigned)S->cur, 0, 0);
        }
    }
    g_threadIdx = dim3((unsigned)S->cur, 0, 0);
}

// Emulated v_mfma_f32_32x32x2_f32 (cdna_hip_programming.md section 3):
//   A operand of lane l = A[i = l & 31][k = l >> 5],  B operand of lane l = B[k = l >> 5][j = l & 31]
//   D register r of lane l = D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]
//   numerics: k-ordered f32 fma chain  D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)).
f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    const int me = S->cur, w = me >> 6, l = me & 63;
    S->wa[w][l] = a;
    S->wb[w][l] = b;
    sync_wave();
    f32x16 d = c;
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = d[r];
        acc = fmaf(S->wa[w][row], S->wb[w][col], acc);
        acc = fmaf(S->wa[w][row + 32], S->wb[w][col + 32], acc);
        d[r] = acc;
    }
    sync_wave();
    return d;
}

// Emulated v_mfma_f32_4x4x1_16B_f32: sixteen independent 4 x 4 x 1 blocks; lane l is in block l >> 2 and supplies A_b[i = l & 3] and
// B_b[j = l & 3]; register r of lane l holds D_b[i = r][j = l & 3] (the general MFMA rule: lanes run over the columns, registers over the rows).
f32x4 mfma_4x4x1(float a, float b, f32x4 c) {
    const int me = S->cur, w = me >> 6, l = me & 63;
    S->wa[w][l] = a;
    S->wb[w][l] = b;
    sync_wave();
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) d[r] = fmaf(S->wa[w][(l & ~3) + r], S->wb[w][l], d[r]);
    sync_wave();
    return d;
}

unsigned short f2bf(float f) {                      // round to nearest even, like v_cvt_pk_bf16_f32
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

static float bf2f(unsigned short h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// Emulated v_mfma_f32_32x32x16_bf16: lane l supplies A[i = l & 31][k = 8 * (l >> 5) + j], B[k = 8 * (l >> 5) + j][n = l & 31];
// same C/D map as the fp32 form; fp32 accumulation in k order (the hardware's internal order is not specified; tests of
// this path use a bf16-sized tolerance).
f32x16 mfma_32x32x16_bf16(u16x8 a, u16x8 b, f32x16 c) {
    const int me = S->cur, w = me >> 6, l = me & 63;
    for (int j = 0; j < 8; ++j) { S->wa16[w][l][j] = a[j]; S->wb16[w][l][j] = b[j]; }
    sync_wave();
    f32x16 d = c;
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = d[r];
        for (int k = 0; k < 16; ++k) {
            const int src_a = row + 32 * (k >> 3), src_b = col + 32 * (k >> 3);
            acc = fmaf(bf2f(S->wa16[w][src_a][k & 7]), bf2f(S->wb16[w][src_b][k & 7]), acc);
        }
        d[r] = acc;
    }
    sync_wave();
    return d;
}

float shfl_xor(float v, int mask) {
    const int me = S->cur, w = me >> 6, l = me & 63;
    S->wa[w][l] = v;
    sync_wave();
    const float r = S->wa[w][(l ^ mask) & 63];
    sync_wave();
    return r;
}

float shfl(float v, int src) {
    const int me = S->cur, w = me >> 6, l = me & 63;
    S->wa[w][l] = v;
    sync_wave();
    const float r = S->wa[w][src & 63];
    sync_wave();
    return r;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    if (block.y != 1 || block.z != 1 || block.x > 64 * kMaxWaves) {
        fprintf(stderr, "ry_emu: only 1-D blocks up to %d threads\n", 64 * kMaxWaves);
        abort();
    }
    const long nblocks = (long)grid.x * grid.y * grid.z;
    if (nblocks <= 0) return;
    int nos = 8;
    if (const char* e = getenv("RY_EMU_THREADS")) nos = atoi(e);
    if (nos < 1) nos = 1;
    if (nos > nblocks) nos = (int)nblocks;
    std::atomic<long> next{0};
    auto worker = [&]() {
        State st;
        st.body = &body;
        for (;;) {
            const long b = next.fetch_add(1);
            if (b >= nblocks) break;
            dim3 bidx((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
            run_block(st, bidx, grid, block);
        }
        for (auto& f : st.fibers) free(f.stack);
        S = nullptr;
    };
    if (nos == 1) {
        worker();
    } else {
        std::vector<std::thread> ts;
        for (int i = 0; i < nos; ++i) ts.emplace_back(worker);
        for (auto& t : ts) t.join();
    }
}

}  // namespace ry_emu
