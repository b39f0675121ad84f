// ry_dev.h -- the few device primitives the gfx950 kernels use, behind one include.
//
// Product build (hipcc --offload-arch=gfx950): the real HIP runtime and CDNA4 builtins.
// Test build (-DRY_HOST_EMU, tests/emu/): the SAME kernel sources run on a host-side SIMT
// emulator (cooperative fibers, 64-lane waves, emulated v_mfma_f32_32x32x2_f32 fragment maps)
// so index arithmetic can be checked against the oracle without a GPU.  The emulator is test
// infrastructure; libry355.so never contains it.
#pragma once

#ifdef RY_HOST_EMU
#include "ry_emu.h"
#else
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RY_DEV __device__ __forceinline__
#define RY_DEV_STATIC static __device__ __forceinline__      // static member functions (the emulator build spells RY_DEV "static inline")
#define RY_KERNEL(...) __global__ __launch_bounds__(__VA_ARGS__)

// v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31];
// D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31] += sum_k A[row][k]*B[k][col]  (exact f32 fma chain).
RY_DEV f32x16 ry_mfma_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_4x4x1_16B_f32: SIXTEEN independent 4 x 4 x 1 outer products per instruction (8 cycles per SIMD, the same 64 FLOP / clk as the
// 32 x 32 form).  Lane l belongs to block l >> 2 and supplies A_b[i = l & 3] and B_b[j = l & 3]; register r of lane l accumulates
// D_b[i = r][j = l & 3] += A_b[r] * B_b[l & 3].  ry_c2d_os maps the blocks to sixteen K indices: a K-batched small-tile GEMM.
RY_DEV f32x4 ry_mfma_4x4x1(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ry_bf16x8 __attribute__((ext_vector_type(8)));
// v_mfma_f32_32x32x16_bf16: lane l supplies A[i=l&31][k=8*(l>>5)+j] and B[k=8*(l>>5)+j][n=l&31], j = 0..7 (bf16 bit
// patterns); C/D map as the fp32 form; products accumulate in fp32.
RY_DEV f32x16 ry_mfma_32x32x16_bf16(u16x8 a, u16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ry_bf16x8, a), __builtin_bit_cast(ry_bf16x8, b), c, 0, 0, 0);
}
RY_DEV unsigned short ry_f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }   // v_cvt_pk_bf16_f32 (RNE)
// Direct global -> LDS copy of 16 bytes per lane (global_load_lds_dwordx4): lane i's data lands at lds_wave_base + 16*i bytes;
// the LDS base must be wave-uniform, the global address is per lane.  Completion is tracked by vmcnt.
RY_DEV void ry_glds16(const float* gsrc_lane, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The same with a wave-uniform base and a 32-bit byte offset per lane: lets the compiler use the scalar-base addressing mode
// (one address VGPR per lane instead of a 64-bit pair, no 64-bit add per piece).
RY_DEV void ry_glds16_off(const float* base_uniform, unsigned byte_off_lane, float* lds_wave_base) {
    const char* q = reinterpret_cast<const char*>(base_uniform) + byte_off_lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)q,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
RY_DEV int ry_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Lanes of one wave exchange data through the LDS: LDS instructions of a wave execute in order, so no hardware barrier is
// needed -- only the compiler must not move the reads above the writes.
RY_DEV void ry_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Workgroup barrier for data exchanged through the LDS only: waits for this wave's LDS operations, not for its vector-memory loads --
// __syncthreads() carries a fence that drains vmcnt, i.e. every global load in flight (ry_c2d_os requests its first filters before the barrier).
RY_DEV void ry_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// A wave reads LDS bytes that its OWN earlier global_load_lds wrote, with AFTER younger vector-memory instructions of this wave allowed to be still
// in flight: an explicit s_waitcnt, and nothing moves across it.  The compiler's own timing of such reads is not to be relied on -- hipcc 7.2 waited for
// the DMA of the first KiB of a ring slot and read the second KiB, requested after that wait, with no wait at all (round 5, ry_c2d_os<2,2,8,2,true>:
// NaNs in rows 4-7 of every tile on the MI355X, profiles/r05/n_xl.txt).  (The emulator runs the lanes one after the other: all of them past the copy first.)
// every LDS read of this wave has delivered its registers; nothing moves across (what follows may overwrite the LDS bytes just read, by DMA)
RY_DEV void ry_lds_reads_returned() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int AFTER>
RY_DEV void ry_own_dma_landed() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AFTER) : "memory"); }
// the instruction scheduler does not move anything across this point
RY_DEV void ry_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
RY_DEV float ry_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
// lane ^ MASK exchange with a compile-time mask: DPP modifiers inside a row of 16 lanes (a VALU-latency move instead of a trip
// through the LDS crossbar: ds_bpermute costs ~100 cycles of latency per dependent step), ds_bpermute across rows.
template <int CTRL>
RY_DEV float ry_dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int MASK>
RY_DEV float ry_shfl_xor_c(float v) {
    if constexpr (MASK == 1) return ry_dpp_mov<0xB1>(v);                       // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) return ry_dpp_mov<0x4E>(v);                  // quad_perm [2,3,0,1]
    else if constexpr (MASK == 4) return ry_dpp_mov<0x1B>(ry_dpp_mov<0x141>(v));   // row_half_mirror (i -> 7 - i), then quad_perm [3,2,1,0]: i -> i ^ 4
    else if constexpr (MASK == 8) return ry_dpp_mov<0x128>(v);                 // row_ror:8
    else return __shfl_xor(v, MASK, 64);
}
RY_DEV float ry_shfl(float v, int src) { return __shfl(v, src, 64); }
// separately rounded float32 product / sum (never contracted into an fma) and the correctly rounded square root: the silence gate
// has to reproduce numpy's float32 arithmetic bit for bit
RY_DEV float ry_mul_rn(float a, float b) { return __fmul_rn(a, b); }
RY_DEV float ry_add_rn(float a, float b) { return __fadd_rn(a, b); }
RY_DEV float ry_sqrt_rn(float a) { return __fsqrt_rn(a); }
RY_DEV int ry_lane() { return (int)(threadIdx.x & 63u); }

typedef hipStream_t ry_stream_t;
#define RY_LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#endif

RY_DEV f32x4 ry_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
RY_DEV void ry_st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// constants the host planner (ry_plan.cpp) and the kernels share: epilogue activations, forms of a stage-1 layer
enum { RY_ACT_NONE = 0, RY_ACT_LRELU = 1, RY_ACT_RELU = 2, RY_ACT_GLU = 3 };
enum { RY_C1D_S2 = 0, RY_C1D_S1 = 1, RY_C1D_DECONV = 2, RY_C1D_GEN = 3 };
