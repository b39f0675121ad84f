// mfma_issue_cost -- what does ONE more instruction of a given kind cost a wave that runs v_mfma_f32_32x32x2_f32 back to back (independent accumulators)?
// One wave per SIMD (256-thread workgroups, one per CU) and two waves per SIMD (512 threads); the fillers are independent of each other and of the MFMAs
// (inline asm, nothing for the compiler to fold).  Output: cycles per MFMA (64 = the pipe's rate) and the cost per filler instruction.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_issue_cost.cpp -o tools/mfma_issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND: 0 none, 1 v_add_f32 (independent), 2 v_pk_add_f32, 3 s_add_u32, 4 ds_read_b128, 5 global_load_lds_dwordx4, 6 s_nop 1, 7 v_mov_b32, 8 v_lshl_add_u64,
//       9 s_waitcnt lgkmcnt(0) behind a ds_read_b128, 10 ds_read_b128 x1 + s_barrier per 8 MFMAs
template <int KIND, int PER, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(const float* in, float* out, unsigned long long* cyc, int n_mfma) {
    __shared__ __attribute__((aligned(16))) float L[16 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16 * 1024; i += THREADS) L[i] = in[i & 511];
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = in[tid & 255], b = in[256 + (tid & 255)];
    float v0 = a, v1 = b, v2 = a + 1, v3 = b + 1, v4 = a + 2, v5 = b + 2, v6 = a + 3, v7 = b + 3;
    f32x2 p0 = {a, b}, p1 = {b, a}, p2 = {a, a}, p3 = {b, b};
    f32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    const unsigned laddr = (unsigned)(lane * 16 + wave * 1024);
    const float* gp = in + lane * 4;
    unsigned long long ga = (unsigned long long)gp;
    float* lds_dst = L + 8192 + wave * 256;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < n_mfma / 8; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int w = (q * PER + u) & 3;
                if (KIND == 1) { if (w == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v0) : "v"(v1)); else if (w == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v2) : "v"(v3));
                                 else if (w == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v4) : "v"(v5)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(v6) : "v"(v7)); }
                if (KIND == 2) { if (w == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(p1)); else if (w == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p1) : "v"(p2));
                                 else if (w == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2) : "v"(p3)); else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p3) : "v"(p0)); }
                if (KIND == 3) { if (w == 0) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) :: "scc"); else if (w == 1) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s1) :: "scc");
                                 else if (w == 2) asm volatile("s_add_u32 %0, %0, 5" : "+s"(s2) :: "scc"); else asm volatile("s_add_u32 %0, %0, 7" : "+s"(s3) :: "scc"); }
                if (KIND == 4) { if (w == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(d0) : "v"(laddr)); else if (w == 1) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(d1) : "v"(laddr));
                                 else if (w == 2) asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(d2) : "v"(laddr)); else asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(d3) : "v"(laddr)); }
                if (KIND == 5) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
                if (KIND == 6) asm volatile("s_nop 1");
                if (KIND == 7) { if (w & 1) asm volatile("v_mov_b32 %0, %1" : "=v"(v0) : "v"(v1)); else asm volatile("v_mov_b32 %0, %1" : "=v"(v2) : "v"(v3)); }
                if (KIND == 8) asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(ga) : "s"((unsigned long long)(16 + u)));
                if (KIND == 9) { asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d0) : "v"(laddr)); }
            }
        }
        if (KIND == 10) { asm volatile("ds_read_b128 %0, %1" : "=v"(d0) : "v"(laddr)); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
        if (KIND == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (KIND >= 11 && KIND <= 14) {                                        // KIND - 10 DMA pieces per 8 MFMAs (the Winograd kernel: 1.4 - 1.9), waited for one group later
#pragma unroll
            for (int u = 0; u < KIND - 10; ++u)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + u * 256), (__attribute__((address_space(3))) void*)(lds_dst + u * 1024), 16, 0, 0);
        }
        if (KIND == 15) {                                                      // two pieces per 8 MFMAs from a wave-uniform base (scalar-base addressing), no 64-bit VALU add
            asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(lane * 16), "s"(in), "s"(__builtin_amdgcn_readfirstlane(8192 * 4 + wave * 2048)) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
    const unsigned long long t1 = clock64();
    float s = v0 + v2 + v4 + v6 + p0[0] + p1[1] + p2[0] + p3[1] + d0[0] + d1[1] + d2[2] + d3[3] + (float)(s0 + s1 + s2 + s3) + (float)(ga & 255);
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * THREADS + tid] = s + L[8192 + tid];
    if (lane == 0) { cyc[blockIdx.x * 16 + wave * 2] = t0; cyc[blockIdx.x * 16 + wave * 2 + 1] = t1; }
}

template <int KIND, int PER, int THREADS>
double run(const float* din, float* dout, unsigned long long* dcyc) {
    const int blocks = 256, n = 8192, nw = THREADS / 64;
    std::vector<unsigned long long> h(blocks * 16);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        probe<KIND, PER, THREADS><<<blocks, THREADS>>>(din, dout, dcyc, n);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
        double sp = 0;
        for (int b = 0; b < blocks; ++b) {
            unsigned long long lo = ~0ull, hi = 0;
            for (int w = 0; w < nw; ++w) { lo = h[b * 16 + w * 2] < lo ? h[b * 16 + w * 2] : lo; hi = h[b * 16 + w * 2 + 1] > hi ? h[b * 16 + w * 2 + 1] : hi; }
            sp += (double)(hi - lo);
        }
        sp /= blocks;
        if (sp < best) best = sp;
    }
    return best / n / (nw / 4);          // cycles per MFMA of the SIMD
}

template <int KIND>
void kind(const char* name, const float* din, float* dout, unsigned long long* dcyc) {
    const double a1 = run<KIND, 1, 256>(din, dout, dcyc), a2 = run<KIND, 2, 256>(din, dout, dcyc), a4 = run<KIND, 4, 256>(din, dout, dcyc);
    const double b1 = run<KIND, 1, 512>(din, dout, dcyc), b2 = run<KIND, 2, 512>(din, dout, dcyc), b4 = run<KIND, 4, 512>(din, dout, dcyc);
    printf("%-34s one wave / SIMD: %6.1f %6.1f %6.1f cycles per MFMA with 1 / 2 / 4 per MFMA (slope %5.1f per instruction)   two waves / SIMD: %6.1f %6.1f %6.1f (slope %5.1f)\n",
           name, a1, a2, a4, (a4 - a1) / 3, b1, b2, b4, (b4 - b1) / 3);
}

int main() {
    float *din, *dout; unsigned long long* dcyc;
    (void)hipMalloc(&din, 8192); (void)hipMalloc(&dout, 256 * 512 * 4); (void)hipMalloc(&dcyc, 256 * 16 * 8);
    std::vector<float> h(2048); for (int i = 0; i < 2048; ++i) h[i] = (float)((i * 37) % 101) / 101.f - 0.5f;
    (void)hipMemcpy(din, h.data(), 8192, hipMemcpyHostToDevice);
    printf("bare MFMAs: %.1f (one wave per SIMD) %.1f (two) cycles per MFMA of the SIMD\n", run<0, 1, 256>(din, dout, dcyc), run<0, 1, 512>(din, dout, dcyc));
    kind<1>("v_add_f32 (independent)", din, dout, dcyc);
    kind<2>("v_pk_add_f32", din, dout, dcyc);
    kind<7>("v_mov_b32", din, dout, dcyc);
    kind<8>("v_lshl_add_u64", din, dout, dcyc);
    kind<3>("s_add_u32", din, dout, dcyc);
    kind<6>("s_nop 1", din, dout, dcyc);
    kind<4>("ds_read_b128", din, dout, dcyc);
    kind<9>("ds_read_b128 + s_waitcnt lgkmcnt(0)", din, dout, dcyc);
    kind<5>("global_load_lds_dwordx4", din, dout, dcyc);
    printf("global_load_lds_dwordx4, 1 / 2 / 3 / 4 per 8 MFMAs: one wave per SIMD %.1f %.1f %.1f %.1f   two waves %.1f %.1f %.1f %.1f   (2 per 8, scalar base, one m0: %.1f / %.1f)\n",
           run<11, 1, 256>(din, dout, dcyc), run<12, 1, 256>(din, dout, dcyc), run<13, 1, 256>(din, dout, dcyc), run<14, 1, 256>(din, dout, dcyc),
           run<11, 1, 512>(din, dout, dcyc), run<12, 1, 512>(din, dout, dcyc), run<13, 1, 512>(din, dout, dcyc), run<14, 1, 512>(din, dout, dcyc), run<15, 1, 256>(din, dout, dcyc), run<15, 1, 512>(din, dout, dcyc));
    printf("one ds_read_b128 + s_waitcnt + s_barrier per 8 MFMAs: %.1f (one wave per SIMD) %.1f (two)\n", run<10, 1, 256>(din, dout, dcyc), run<10, 1, 512>(din, dout, dcyc));
    return 0;
}
