// ry_emu.h -- host-side SIMT emulator for the gfx950 kernel sources.  TEST INFRASTRUCTURE ONLY.
//
// Compiles realtime_yukarin_amd/csrc/*.h kernels as plain C++ (-DRY_HOST_EMU): every GPU thread is
// a cooperative fiber (own x86-64 context switch; ucontext elsewhere), a workgroup is a set of fibers on one OS thread, `__shared__` is a
// thread_local static shared by those fibers, `__syncthreads()` / wave collectives are fiber
// rendezvous, and v_mfma_f32_32x32x2_f32 is emulated with the fragment maps documented in
// /opt/skills/guides/cdna_hip_programming.md section 3.  Used by tests/ (-m "not gpu") to check kernel index
// arithmetic against the oracle where no GPU exists.  Never linked into libry355.so.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace ry_emu {
extern thread_local dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
void sync_block();
void wave_sync();
f32x16 mfma_32x32x2(float a, float b, f32x16 c);
f32x16 mfma_32x32x16_bf16(u16x8 a, u16x8 b, f32x16 c);
f32x4 mfma_4x4x1(float a, float b, f32x4 c);
unsigned short f2bf(float f);
float shfl_xor(float v, int mask);
float shfl(float v, int src);
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
}  // namespace ry_emu

#define threadIdx ry_emu::g_threadIdx
#define blockIdx ry_emu::g_blockIdx
#define blockDim ry_emu::g_blockDim
#define gridDim ry_emu::g_gridDim
#define __shared__ static thread_local
#define __syncthreads() ry_emu::sync_block()

#define RY_DEV static inline
#define RY_DEV_STATIC static inline
#define RY_KERNEL(...)

RY_DEV f32x16 ry_mfma_32x32x2(float a, float b, f32x16 c) { return ry_emu::mfma_32x32x2(a, b, c); }
RY_DEV f32x16 ry_mfma_32x32x16_bf16(u16x8 a, u16x8 b, f32x16 c) { return ry_emu::mfma_32x32x16_bf16(a, b, c); }
RY_DEV f32x4 ry_mfma_4x4x1(float a, float b, f32x4 c) { return ry_emu::mfma_4x4x1(a, b, c); }
RY_DEV unsigned short ry_f2bf(float f) { return ry_emu::f2bf(f); }
RY_DEV void ry_glds16(const float* gsrc_lane, float* lds_wave_base) {          // emulated global_load_lds_dwordx4
    memcpy(lds_wave_base + 4 * (threadIdx.x & 63u), gsrc_lane, 16);
}
RY_DEV void ry_glds16_off(const float* base_uniform, unsigned byte_off_lane, float* lds_wave_base) {
    ry_glds16(reinterpret_cast<const float*>(reinterpret_cast<const char*>(base_uniform) + byte_off_lane), lds_wave_base);
}
RY_DEV int ry_uniform(int v) { return v; }
RY_DEV void ry_wave_sync() { ry_emu::wave_sync(); }
RY_DEV void ry_lds_barrier() { ry_emu::sync_block(); }
template <int AFTER> RY_DEV void ry_own_dma_landed() { ry_emu::wave_sync(); }
RY_DEV void ry_lds_reads_returned() { ry_emu::wave_sync(); }
RY_DEV void ry_sched_fence() {}
RY_DEV float ry_mul_rn(float a, float b) { volatile float r = a * b; return r; }     // volatile: no contraction with a following add
RY_DEV float ry_add_rn(float a, float b) { volatile float r = a + b; return r; }
RY_DEV float ry_sqrt_rn(float a) { return sqrtf(a); }
RY_DEV float ry_shfl_xor(float v, int mask) { return ry_emu::shfl_xor(v, mask); }
template <int MASK> RY_DEV float ry_shfl_xor_c(float v) { return ry_emu::shfl_xor(v, MASK); }
RY_DEV float ry_shfl(float v, int src) { return ry_emu::shfl(v, src); }
RY_DEV int ry_lane() { return (int)(threadIdx.x & 63u); }

typedef void* ry_stream_t;
#define RY_LAUNCH(kernel, grid, block, stream, ...) \
    ry_emu::launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })
