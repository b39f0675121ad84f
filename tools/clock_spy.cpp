// clock_spy -- what clock does the chip sustain while a workload runs?  One wave on its own stream samples the shader-cycle counter (s_memtime, clock64())
// against the constant-rate counter (s_memrealtime, wall_clock64()) every `period_us` for `total_us`; the host turns consecutive samples into MHz.
// The DVFS note of MI355X_MICROARCH.md says that denser kernels clock lower: this measures it for back-to-back graph replays (round 6: the Winograd plans that
// won layer by layer lost together).  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/clock_spy.cpp -o tools/libclock_spy.so   (scripts/gpu_r6_clocks.py)
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void spy_kernel(unsigned long long* out, int n, unsigned long long period_ticks) {
    if (threadIdx.x != 0) return;
    unsigned long long t_next = wall_clock64();
    for (int i = 0; i < n; ++i) {
        unsigned long long w;
        do { w = wall_clock64(); __builtin_amdgcn_s_sleep(32); } while (w < t_next);
        out[2 * i] = w;
        out[2 * i + 1] = clock64();
        t_next = w + period_ticks;
    }
}

static hipStream_t g_stream = nullptr;
static unsigned long long* g_buf = nullptr;
static int g_n = 0;

extern "C" int spy_start(int period_us, int total_us) {
    int khz = 100000;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0) != hipSuccess || khz <= 0) khz = 100000;
    if (!g_stream && hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return -1;
    g_n = total_us / period_us;
    if (g_buf) hipFree(g_buf);
    if (hipMalloc(&g_buf, (size_t)g_n * 16) != hipSuccess) return -2;
    hipLaunchKernelGGL(spy_kernel, dim3(1), dim3(64), 0, g_stream, g_buf, g_n, (unsigned long long)period_us * (unsigned long long)khz / 1000ull);
    return khz;
}

// -> number of samples copied into out (pairs: wall ticks, shader cycles)
extern "C" int spy_collect(unsigned long long* out, int max_pairs) {
    if (hipStreamSynchronize(g_stream) != hipSuccess) return -1;
    const int n = g_n < max_pairs ? g_n : max_pairs;
    if (hipMemcpy(out, g_buf, (size_t)n * 16, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    return n;
}
