// stream_probe.cpp -- how fast can the workgroups of ONE launch pull operands into registers on this box (measurement aid for ry_c2d_os,
// round 5)?  G workgroups of 4 waves; every wave streams `kb` KiB as 1-KiB wave-loads (16 bytes per lane), `depth` loads in flight, and
// adds them up.  What is varied is WHERE the waves read at any one time:
//   private   every wave its own region (weights: HBM / MALL), regions a power-of-two apart, all waves walking in lockstep from offset 0
//   rotated   the same regions, wave w of workgroup g starts (7 g + 11 w) KiB into its region and wraps (lockstep broken)
//   shared    every workgroup reads the SAME `kb` KiB x 4 waves (activations: L2 hits), in lockstep / rotated
//   nt        the private stream with non-temporal loads
// Reported: microseconds per launch (HIP events over `reps` back-to-back launches, so ~2 us of launch are inside) and GB/s per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/stream_probe.cpp -o tools/stream_probe     Run: tools/stream_probe [G] [KiB per wave] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, bool NT>
__global__ __launch_bounds__(256, 2) void stream(const float* base, float* out, int kb, long long wave_stride_floats, int rot_g, int rot_w, int shared) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = blockIdx.x;
    const float* p = base + (shared ? (long long)wave : ((long long)g * 4 + wave)) * wave_stride_floats + lane * 4;
    int u = (int)(((unsigned)(g * rot_g + wave * rot_w)) % (unsigned)kb);
    f32x4 v[DEPTH];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const f32x4* q = reinterpret_cast<const f32x4*>(p + (long long)u * 256);
        v[d] = NT ? __builtin_nontemporal_load(q) : *q;
        if (++u == kb) u = 0;
    }
    for (int i = DEPTH; i < kb; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            acc += v[d];
            const f32x4* q = reinterpret_cast<const f32x4*>(p + (long long)u * 256);
            v[d] = NT ? __builtin_nontemporal_load(q) : *q;
            if (++u == kb) u = 0;
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[g] = acc[0];
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 256, kb = argc > 2 ? atoi(argv[2]) : 160, reps = argc > 3 ? atoi(argv[3]) : 20;
    const long long wave_stride = (long long)kb * 256;                         // floats: regions back to back (kb KiB each)
    const size_t total = (size_t)G * 4 * kb * 1024;
    float *buf, *out, *flush;
    const size_t flush_bytes = (size_t)512 << 20;
    CK(hipMalloc(&buf, total + 4096)); CK(hipMalloc(&out, G * 4)); CK(hipMalloc(&flush, flush_bytes));
    CK(hipMemset(buf, 0, total)); CK(hipMemset(flush, 0, flush_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# stream_probe: %d workgroups x 4 waves x %d KiB = %.1f MB per launch, %d reps (a 512 MB memset between timed launches evicts MALL when cold=1)\n", G, kb, total / 1e6, reps);
    struct Case { const char* name; int rot_g, rot_w, shared, nt, depth, cold; };
    const Case cases[] = {
        {"private lockstep  depth 8", 0, 0, 0, 0, 8, 1}, {"private rotated   depth 8", 7, 11, 0, 0, 8, 1},
        {"private lockstep  depth 4", 0, 0, 0, 0, 4, 1}, {"private rotated   depth 4", 7, 11, 0, 0, 4, 1},
        {"private lockstep  depth 16", 0, 0, 0, 0, 16, 1}, {"private rotated   depth 16", 7, 11, 0, 0, 16, 1},
        {"private rotated nt depth 8", 7, 11, 0, 1, 8, 1}, {"private lockstep nt depth 8", 0, 0, 0, 1, 8, 1},
        {"private rotated   depth 8 warm", 7, 11, 0, 0, 8, 0}, {"private lockstep  depth 8 warm", 0, 0, 0, 0, 8, 0},
        {"shared  lockstep  depth 8", 0, 0, 1, 0, 8, 0}, {"shared  rotated   depth 8", 7, 11, 1, 0, 8, 0},
        {"shared  rotated   depth 16", 7, 11, 1, 0, 16, 0},
    };
    for (const Case& c : cases) {
        float tot = 0.f;
        for (int r = 0; r < reps + 2; ++r) {
            if (c.cold) CK(hipMemsetAsync(flush, r & 1, flush_bytes, 0));
            CK(hipEventRecord(e0, 0));
#define L(D, N) hipLaunchKernelGGL((stream<D, N>), dim3(G), dim3(256), 0, 0, buf, out, kb, wave_stride, c.rot_g, c.rot_w, c.shared)
            if (c.nt) L(8, true); else if (c.depth == 4) L(4, false); else if (c.depth == 16) L(16, false); else L(8, false);
#undef L
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) tot += ms;
        }
        const double us = tot / reps * 1e3;
        const double cus = G < 256 ? G : 256;
        printf("%-32s %8.2f us   %7.1f GB/s per CU   %6.2f TB/s chip\n", c.name, us, (double)total / cus / (us * 1e-6) / 1e9, (double)total / (us * 1e-6) / 1e12);
    }
    return 0;
}
