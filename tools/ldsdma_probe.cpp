// ldsdma_probe.cpp -- where does global_load_lds_dwordx4 put its data when the LDS destination lies above 64 KiB (measurement aid, round 5)?
// One wave copies one KiB (lane l: floats 4 l .. 4 l + 3 of `src`) to LDS byte offset X by DMA, waits, and every lane reads its 16 bytes back from X
// and from X - 64 KiB.  "ok" = the data is at X.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ldsdma_probe.cpp -o tools/ldsdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

__global__ __launch_bounds__(64) void probe(const float* src, float* out, int x_bytes) {
    __shared__ __attribute__((aligned(16))) float lds[40 * 1024];             // 160 KiB
    const int lane = threadIdx.x;
    for (int i = lane; i < 40 * 1024; i += 64) lds[i] = -1.f;
    __syncthreads();
    float* dst = lds + __builtin_amdgcn_readfirstlane(x_bytes / 4);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 4), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const f32x4 a = *reinterpret_cast<const f32x4*>(lds + x_bytes / 4 + lane * 4);
    const int lo = (x_bytes & 0xffff) / 4;
    const f32x4 b = *reinterpret_cast<const f32x4*>(lds + lo + lane * 4);
    *reinterpret_cast<f32x4*>(out + lane * 4) = a;
    *reinterpret_cast<f32x4*>(out + 256 + lane * 4) = b;
}

int main() {
    float h[256], r[512], *src, *out;
    for (int i = 0; i < 256; ++i) h[i] = (float)(i + 1);
    CK(hipMalloc(&src, sizeof h)); CK(hipMalloc(&out, sizeof r)); CK(hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice));
    const int xs[] = {0, 1024, 32768, 64512, 65536, 66560, 98304, 131072, 162816};
    for (int x : xs) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, x);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(r, out, sizeof r, hipMemcpyDeviceToHost));
        int at_x = 0, at_lo = 0;
        for (int i = 0; i < 256; ++i) { at_x += r[i] == h[i]; at_lo += r[256 + i] == h[i]; }
        printf("LDS byte offset %6d: %3d / 256 floats at the offset, %3d / 256 at offset mod 64 KiB   %s\n", x, at_x, at_lo, at_x == 256 ? "ok" : "NOT at the offset");
    }
    return 0;
}
