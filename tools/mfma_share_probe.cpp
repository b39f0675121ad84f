// mfma_share_probe -- what does a wave running v_mfma_f32_32x32x2_f32 lose to the OTHER wave of its SIMD, by what that wave does?
// 512-thread workgroups (two waves per SIMD, one workgroup per CU): waves 0-3 run N independent-accumulator fp32 MFMAs and time themselves (s_memtime);
// waves 4-7 run a partner loop until the MFMA waves are done: nothing / fp32 VALU (v_sub_f32) / integer VALU / ds_read_b128 / SALU / fp32 MFMAs too.
// Round 6: the Winograd kernel's time came out as MFMA time PLUS the time of everything else, two waves per SIMD or not (profiles/r06/wino_ablation.txt).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_share_probe.cpp -o tools/mfma_share_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int SELF>
__global__ __launch_bounds__(512, 2) void probe(const float* in, float* out, unsigned long long* cyc, int n_mfma) {
    __shared__ __attribute__((aligned(16))) float L[32 * 1024];           // 128 KiB: one workgroup per CU
    __shared__ volatile int done;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 32 * 1024; i += 512) L[i] = in[i & 511];
    if (tid == 0) done = 0;
    __syncthreads();
    if (wave < 4 || MODE == 9) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float a = in[tid], b = in[256 + tid];
        f32x4 x = *(const f32x4*)&L[lane * 4], y = x;
        int ii = lane;
        const unsigned long long t0 = clock64();
        for (int it = 0; it < n_mfma / 8; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
                // SELF: the MFMA wave's own filler between MFMAs (per MFMA): 1 = two fp32 VALU, 2 = two integer VALU, 3 = one ds_read_b128
                if (SELF == 1) { x[q & 3] -= y[(q + 1) & 3]; y[q & 3] -= x[(q + 2) & 3]; }
                if (SELF == 2) { ii = ii * 3 + q; ii ^= (ii >> 3); }
                if (SELF == 3) { y += *(const f32x4*)&L[((ii + q * 64) & 8191) * 4]; }
                if (SELF == 4) { x[q & 3] -= y[(q + 1) & 3]; y[q & 3] -= x[(q + 2) & 3]; x[(q + 1) & 3] -= y[(q + 2) & 3]; y[(q + 3) & 3] -= x[q & 3]; }      // four fp32 VALU
                if (SELF == 5) { x[q & 3] -= y[(q + 1) & 3]; y[q & 3] -= x[(q + 2) & 3]; x[(q + 1) & 3] -= y[(q + 2) & 3]; y[(q + 3) & 3] -= x[q & 3];
                                 x[(q + 2) & 3] -= y[q & 3]; y[(q + 1) & 3] -= x[(q + 3) & 3]; x[(q + 3) & 3] -= y[(q + 1) & 3]; y[(q + 2) & 3] -= x[(q + 1) & 3]; }   // eight
            }
            if (SELF == 6) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { x[q & 3] -= y[(q + 1) & 3]; y[q & 3] -= x[(q + 2) & 3]; }
            }
            if (SELF == 7) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    f32x2 a2 = {x[0], x[1]}, b2 = {y[2], y[3]}, c2 = {x[2], x[3]}, d2 = {y[0], y[1]};
                    a2 -= b2; c2 -= d2; x[0] = a2[0]; x[1] = a2[1]; x[2] = c2[0]; x[3] = c2[1];
                    f32x2 e2 = {y[0], y[1]}, f2 = {y[2], y[3]};
                    e2 -= c2; f2 -= a2; y[0] = e2[0]; y[1] = e2[1]; y[2] = f2[0]; y[3] = f2[1];
                }
            }
            if (SELF == 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q) y += *(const f32x4*)&L[((ii + q * 64) & 8191) * 4];
            }
        }
        const unsigned long long t1 = clock64();
        float s = x[0] + y[1] + (float)ii;
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * 512 + tid] = s;
        if (lane == 0) { cyc[blockIdx.x * 16 + wave * 2] = t0; cyc[blockIdx.x * 16 + wave * 2 + 1] = t1; }
        if (lane == 0) atomicAdd((int*)&done, 1);
    } else {
        f32x4 x = *(const f32x4*)&L[lane * 4], y = *(const f32x4*)&L[lane * 4 + 256];
        int ii = lane, k = 0, npart = 0;
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        if (MODE != 0) {
            while (done < 4) {
                ++npart;
                for (int u = 0; u < 16; ++u) {
                    if (MODE == 1) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) { x[q & 3] -= y[(q + 1) & 3]; y[q & 3] -= x[(q + 2) & 3]; }
                    } else if (MODE == 2) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) { ii = ii * 3 + q; ii ^= (ii >> 3); }
                    } else if (MODE == 3) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) y += *(const f32x4*)&L[((ii + q * 64 + k) & 8191) * 4];
                        k += 512;
                    } else if (MODE == 4) {
                        asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 5\n s_add_u32 %0, %0, 7\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 5\n s_add_u32 %0, %0, 7" : "+s"(k));
                    } else if (MODE == 5) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[q], y[q], acc[q], 0, 0, 0);
                    }
                }
            }
        }
        if (lane == 0 && wave == 4) cyc[blockIdx.x * 16 + 15] = (unsigned long long)npart;        // partner loop bodies executed while the MFMA waves ran
        float s = x[0] + x[1] + x[2] + x[3] + y[0] + y[1] + y[2] + y[3] + (float)ii + (float)k;
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * 512 + tid] = s;
    }
}

static double g_part = 0;
template <int MODE, int SELF>
void run(const char* name, const float* din, float* dout, unsigned long long* dcyc) {
    const int blocks = 256, n = 16384;
    std::vector<unsigned long long> h(blocks * 16);
    double best = 1e30, best_pair = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(dcyc, 0, h.size() * 8);
        probe<MODE, SELF><<<blocks, 512>>>(din, dout, dcyc, n);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0, sp = 0;
        for (int b = 0; b < blocks; ++b) {
            unsigned long long lo = ~0ull, hi = 0;
            for (int w = 0; w < 8; ++w) {
                const unsigned long long t0 = h[b * 16 + w * 2], t1 = h[b * 16 + w * 2 + 1];
                if (t1 == 0) continue;
                if (w < 4) s += (double)(t1 - t0) / 4;
                lo = t0 < lo ? t0 : lo; hi = t1 > hi ? t1 : hi;
            }
            sp += (double)(hi - lo);
        }
        s /= blocks; sp /= blocks;
        if (MODE != 9 && MODE != 0) { double np = 0; for (int b = 0; b < blocks; ++b) np += (double)h[b * 16 + 15]; g_part = np / blocks; }
        if (s < best) best = s;
        if (sp < best_pair) best_pair = sp;
    }
    if (MODE == 9) printf("%-58s older wave %6.1f cycles per MFMA; both waves done after %6.1f cycles per MFMA PAIR (128 = pipe full)\n", name, best / n, best_pair / n);
    else if (MODE != 0) printf("%-58s %6.1f cycles per MFMA of the timed wave (64 = the pipe's rate); partner: %.0f loop bodies = one per %.1f cycles\n", name, best / n, g_part, best / g_part);
    else printf("%-58s %6.1f cycles per MFMA of the timed wave (64 = the pipe's rate)\n", name, best / n);
}

int main() {
    float *din, *dout; unsigned long long* dcyc;
    (void)hipMalloc(&din, 4096); (void)hipMalloc(&dout, 256 * 512 * 4); (void)hipMalloc(&dcyc, 256 * 16 * 8);
    std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 37) % 101) / 101.f - 0.5f;
    (void)hipMemcpy(din, h.data(), 4096, hipMemcpyHostToDevice);
    run<0, 0>("MFMA wave alone on its SIMD", din, dout, dcyc);
    run<1, 0>("partner wave: fp32 VALU (v_sub_f32) back to back", din, dout, dcyc);
    run<2, 0>("partner wave: integer VALU back to back", din, dout, dcyc);
    run<3, 0>("partner wave: ds_read_b128 back to back", din, dout, dcyc);
    run<4, 0>("partner wave: SALU back to back", din, dout, dcyc);
    run<5, 0>("partner wave: fp32 MFMAs too", din, dout, dcyc);
    run<0, 1>("alone, two fp32 VALU behind every MFMA of the timed wave", din, dout, dcyc);
    run<0, 2>("alone, two integer VALU behind every MFMA", din, dout, dcyc);
    run<0, 3>("alone, one ds_read_b128 + add behind every MFMA", din, dout, dcyc);
    run<1, 1>("two fp32 VALU behind every MFMA + fp32 VALU partner", din, dout, dcyc);
    run<0, 4>("alone, four fp32 VALU behind every MFMA", din, dout, dcyc);
    run<0, 5>("alone, eight fp32 VALU behind every MFMA", din, dout, dcyc);
    run<0, 6>("alone, 8 MFMAs then 16 fp32 VALU in one block", din, dout, dcyc);
    run<0, 7>("alone, 8 MFMAs then 32 packed-fp32 VALU in one block", din, dout, dcyc);
    run<0, 8>("alone, 8 MFMAs then 8 ds_read_b128 + add in one block", din, dout, dcyc);
    run<9, 6>("both: 8 MFMAs then 16 fp32 VALU in one block", din, dout, dcyc);
    run<9, 8>("both: 8 MFMAs then 8 ds_read_b128 + add in one block", din, dout, dcyc);
    printf("-- both waves of a SIMD run the SAME loop (cycles per MFMA of one wave: 128 = the pipe shared evenly and full)\n");
    run<9, 0>("both: bare MFMAs", din, dout, dcyc);
    run<9, 1>("both: two fp32 VALU behind every MFMA", din, dout, dcyc);
    run<9, 4>("both: four fp32 VALU behind every MFMA", din, dout, dcyc);
    run<9, 5>("both: eight fp32 VALU behind every MFMA", din, dout, dcyc);
    run<9, 2>("both: two integer VALU behind every MFMA", din, dout, dcyc);
    run<9, 3>("both: one ds_read_b128 + add behind every MFMA", din, dout, dcyc);
    return 0;
}
