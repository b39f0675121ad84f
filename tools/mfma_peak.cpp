// mfma_peak.cpp -- what v_mfma_f32_32x32x2_f32 sustains on THIS box at real clocks (random operands),
// bare and with the LDS fragment-read pattern of ry_igemm_f32.  Measurement aid only.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.cpp -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: one operand pair, 4 accumulators.  MODE 1: 2x2 tiles, operands from registers (8 k-steps).
// MODE 2: as MODE 1 but operands re-read from LDS (ds_read_b128) every 32 MFMAs, like the igemm inner loop.
// MODE 3: MODE 2 + one __syncthreads() per 64 MFMAs; MODE 4: MODE 2 + two per 64 MFMAs (the igemm's barrier cadence).
template <int MODE>
__global__ __launch_bounds__(256) void mfma_loop(const float* in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float As[128 * 36];
    __shared__ __attribute__((aligned(16))) float Bs[128 * 36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 128 * 36; i += 256) { As[i] = in[i & 511]; Bs[i] = in[(i * 7) & 511]; }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm = wave >> 1, wn = wave & 1;
    if (MODE == 0) {
        float a = in[tid], b = in[256 + tid];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q >> 1][q & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q >> 1][q & 1], 0, 0, 0);
            a += 1e-6f;
        }
    } else {
        f32x4 af[2][2], bf[2][2];
        for (int s = 0; s < 2; ++s) for (int i = 0; i < 2; ++i) {
            af[s][i] = *(const f32x4*)&As[((wm * 2 + i) * 32 + lr) * 36 + s * 8 + lh * 4];
            bf[s][i] = *(const f32x4*)&Bs[((wn * 2 + i) * 32 + lr) * 36 + s * 8 + lh * 4];
        }
        for (int it = 0; it < iters / 8; ++it) {          // 32 MFMAs per iteration
            if (MODE >= 3 && (it & 1) == 0) { __syncthreads(); if (MODE == 4) __syncthreads(); }
            if (MODE >= 2) {
                int o = (it & 1) * 16;
                asm volatile("" : "+v"(o));                 // opaque: the fragment reads cannot be hoisted out of the loop
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        af[s][i] = *(const f32x4*)&As[((wm * 2 + i) * 32 + lr) * 36 + o + s * 8 + lh * 4];
                        bf[s][i] = *(const f32x4*)&Bs[((wn * 2 + i) * 32 + lr) * 36 + o + s * 8 + lh * 4];
                    }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s][i][t], bf[s][j][t], acc[i][j], 0, 0, 0);
            if (MODE == 1) asm volatile("" : "+v"(af[0][0]), "+v"(bf[0][0]));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks, const float* din, float* dout) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        mfma_loop<MODE><<<blocks, 256>>>(din, dout, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 * iters * 4 * (2.0 * 32 * 32 * 2);
    printf("%-44s blocks=%4d  %.3f ms  %.1f TFLOP/s\n", name, blocks, best, flops / best / 1e9);
}

int main() {
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    float *din, *dout;
    (void)hipMalloc(&din, 512 * 4); (void)hipMalloc(&dout, 1024 * 256 * 4);
    (void)hipMemcpy(din, h.data(), 512 * 4, hipMemcpyHostToDevice);
    for (int blocks : {256, 512, 768}) {
        run<0>("mfma only, 1 operand pair", blocks, din, dout);
        run<1>("mfma 2x2 tiles, register operands", blocks, din, dout);
        run<2>("mfma 2x2 tiles, ds_read_b128 operands", blocks, din, dout);
        run<3>("  + 1 barrier per 64 MFMAs", blocks, din, dout);
        run<4>("  + 2 barriers per 64 MFMAs", blocks, din, dout);
    }
    return 0;
}
