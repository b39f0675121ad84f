// ry_vc_kernels.h -- the small kernels of the window call (ry_vc.cpp): combine_silent (row scatter), the silence gate
// (frame power in numpy's summation order, gate + ordered compaction) and decode_spectrogram (mc2sp = exp(mc @ M)).
// Reference: /root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:27-39.
#pragma once
#include "ry_dev.h"

// ---------------------------------------------------------------------------------------------
// Between the two CNNs: `AcousticConverter.combine_silent` (scatter converted rows into an all-silent block) and
// `decode_spectrogram` = pysptk.mc2sp.  mc2sp is freqt(-alpha) -> c0 *= 2 -> symmetric extension -> Re(rfft) -> exp;
// everything before the exp is linear in the mel-cepstrum, so sp = exp(mc @ M) with an (order+1) x (fftlen/2+1)
// matrix M computed once on the host in float64 (realtime_yukarin_amd/sptk.py).  `floor` is the 1e-16 the reference
// adds before stage-2 (voice_changer.py:39).
// ---------------------------------------------------------------------------------------------
struct RyScatterParams { const float* src; const int* row_of; float* dst; int n_src, cols; };

RY_KERNEL(256) void ry_scatter_rows(RyScatterParams p) {       // dst[row_of[i]][:] = src[i][:]
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)p.n_src * p.cols) return;
    const int c = (int)(idx % p.cols), i = (int)(idx / p.cols);
    p.dst[(size_t)p.row_of[i] * p.cols + c] = p.src[idx];
}

// ---------------------------------------------------------------------------------------------
// `AcousticConverter.separate_effective` on the device (voice_changer.py:27-31).
// ry_frame_power: mse[t] = librosa.feature.rms(wave, frame_length, hop, center=True, pad_mode='reflect')[t] ** 2 in float32 with
//   NUMPY'S summation order, so that the mask is the host formula's bit for bit: `mean(abs(x) ** 2, axis=0)` of librosa's frame view
//   is numpy's pairwise sum -- leaves of 128 elements, each 8 interleaved accumulators r[j] += a[8 i + j] combined as
//   ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), leaves combined in a binary tree.  One wave per frame: lane (leaf, j) walks its
//   16 elements in order, then a butterfly over the lanes IS that tree (fp addition is commutative, the tree shape is what matters).
//   fft_length = 128 .. 1024, a power of two.  Products and sums are rounded separately (no fma contraction), sqrt is correctly rounded.
// ry_gate_compact: one workgroup: the gate in the POWER domain -- effective = mse >= p_eff, or every frame when max(mse) >= p_all
//   (librosa.power_to_db's top_db clamp) -- with both thresholds derived on the host, once per threshold_db, by bisection over the
//   float32 values through the host's own log10 (realtime_yukarin_amd/gate.py), then an ordered compaction: row_of, the feature rows
//   of the effective frames, the mask and the count.
// ---------------------------------------------------------------------------------------------
struct RyFramePowerParams { const float* wave; int n, hop, fft, n_wave_frames; float* power; };

RY_DEV int ry_reflect_index(int i, int n) {                     // numpy.pad(mode='reflect') source index
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    int j = i % period;
    if (j < 0) j += period;
    return j < n ? j : period - j;
}

RY_KERNEL(256) void ry_frame_power(RyFramePowerParams p) {
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int frame = (int)blockIdx.x * 4 + (tid >> 6);
    if (frame >= p.n_wave_frames) return;                       // wave-uniform
    const int L = p.fft >> 4;                                   // active lanes: 8 per 128-element leaf
    float r = 0.f;
    if (lane < L) {
        const int base = frame * p.hop - (p.fft >> 1) + 128 * (lane >> 3) + (lane & 7);
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = p.wave[ry_reflect_index(base + 8 * i, p.n)];
        r = ry_mul_rn(v[0], v[0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) r = ry_add_rn(r, ry_mul_rn(v[i], v[i]));
    }
    for (int mask = 1; mask < L; mask <<= 1) r = ry_add_rn(r, ry_shfl_xor(r, mask));
    if (lane == 0) {
        const float rms = ry_sqrt_rn(r / (float)p.fft);        // a power of two: the division is exact scaling, as numpy's
        p.power[frame] = ry_mul_rn(rms, rms);
    }
}

struct RyGateParams {
    const float* power; int n_wave_frames, n_frames; float p_eff, p_all;
    const float* feat; int cin;                                // [n_frames][cin]
    float* x_eff; int* row_of; int* count; unsigned char* mask;
};

RY_KERNEL(1024) void ry_gate_compact(RyGateParams p) {
    __shared__ float fmx[1024];
    __shared__ int scan[1024];
    __shared__ int s_base;
    const int tid = (int)threadIdx.x;
    float m = 0.f;
    for (int t = tid; t < p.n_wave_frames; t += 1024) m = fmaxf(m, p.power[t]);
    fmx[tid] = m;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) { if (tid < s) fmx[tid] = fmaxf(fmx[tid], fmx[tid + s]); __syncthreads(); }
    const bool all = fmx[0] >= p.p_all;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int t0 = 0; t0 < p.n_frames; t0 += 1024) {
        const int t = t0 + tid;
        const bool eff = t < p.n_frames && t < p.n_wave_frames && (all || p.power[t] >= p.p_eff);
        scan[tid] = eff ? 1 : 0;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {                    // inclusive Hillis-Steele scan
            const int add = tid >= d ? scan[tid - d] : 0;
            __syncthreads();
            scan[tid] += add;
            __syncthreads();
        }
        const int base = s_base;
        if (t < p.n_frames) p.mask[t] = eff ? 1 : 0;
        if (eff) {
            const int pos = base + scan[tid] - 1;
            p.row_of[pos] = t;
            for (int c = 0; c < p.cin; ++c) p.x_eff[(size_t)pos * p.cin + c] = p.feat[(size_t)t * p.cin + c];
        }
        __syncthreads();
        if (tid == 1023) s_base = base + scan[1023];
        __syncthreads();
    }
    if (tid == 0) *p.count = s_base;
}

struct RyMc2spParams { const float* mc; const float* mtx; float* sp; int n, m, f; float floor; };

RY_KERNEL(256) void ry_mc2sp(RyMc2spParams p) {                 // sp[n][f] = exp(sum_m mc[n][m] * mtx[m][f]) + floor
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)p.n * p.f) return;
    const int f = (int)(idx % p.f), n = (int)(idx / p.f);
    const float* row = p.mc + (size_t)n * p.m;
    float z = 0.f;
    for (int m = 0; m < p.m; ++m) z = fmaf(row[m], p.mtx[(size_t)m * p.f + f], z);
    p.sp[idx] = expf(z) + p.floor;
}
