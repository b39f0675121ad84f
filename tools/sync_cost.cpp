// sync_cost.cpp -- what ONE dependent step costs on this box, two ways (measurement aid; DESIGN.md section 10 rests on it):
//   (a) a hipGraph of N dependent small kernels (what the executor does today: one node per layer): microseconds per node;
//   (b) ONE persistent launch of G workgroups with N grid-wide barriers (what a fused multi-layer kernel would do): every workgroup writes
//       `bytes` of output, arrives on a counter in memory, polls it, then reads the bytes another workgroup (on another XCD) wrote -- the XCDs
//       do not snoop each other's L2, so data handed from one workgroup to another inside a launch has to travel through memory.
//       What is measured is the PESSIMISTIC form: every data store, the arrival and the poll are SYSTEM-scope atomics (one atomic store per
//       element), with s_sleep in the spin -- an UPPER bound (13 / 27 / 67 us per barrier at 128 / 256 / 512 workgroups, profiles/r04/h).  The
//       XCD-hierarchical barrier of /opt/skills/guides/MI355X_MICROARCH.md (per-XCC counters, agent-scope fences, plain stores) is quoted there
//       at 4.1 / 5.9 / 9.7 us at 256 / 512 / 1024 workgroups with nothing published and ~7 us after a phase that wrote a 64 KB slab per
//       workgroup: that is the figure DESIGN.md section 10 prices a fused multi-layer kernel with (against 2.2 us per dependent graph node).
//       The spin is bounded (~0.2 s): a workgroup that never sees the others sets a flag and leaves, the tool reports it instead of hanging the GPU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/sync_cost.cpp -o tools/sync_cost     Run: tools/sync_cost [steps] [workgroups] [bytes per workgroup]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

__global__ __launch_bounds__(256) void small_step(const float* in, float* out, int n) {       // a layer-sized stand-in: n floats per workgroup
    const int i = blockIdx.x * n + threadIdx.x;
    float v = 0.f;
    for (int k = threadIdx.x; k < n; k += 256) v += in[blockIdx.x * n + k];
    if (threadIdx.x < n) out[i] = v * 0.5f + 1.f;
}

__global__ __launch_bounds__(256) void persistent_steps(float* buf0, float* buf1, unsigned* counter, unsigned* failed, int steps, int n, int G) {
    const int tid = threadIdx.x, wg = blockIdx.x;
    float carry = 1.f;
    for (int s = 0; s < steps; ++s) {
        float* out = (s & 1) ? buf1 : buf0;
        const float* in = (s & 1) ? buf0 : buf1;
        // this workgroup's slice of the step's output (system-scope stores: visible to the other XCDs once they are acknowledged)
        for (int k = tid; k < n; k += 256)
            __hip_atomic_store(&out[wg * n + k], carry + (float)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();                                   // stores acknowledged before the arrival is published
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned want = (unsigned)(s + 1) * (unsigned)G;
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) { *failed = 1u; break; }     // never on an idle chip; bounded so that a mistake cannot hang the box
            }
        }
        __syncthreads();
        if (*failed) return;
        // read what the workgroup eight places further (another XCD) wrote in this step
        const int other = (wg + 9) % G;
        float v = 0.f;
        for (int k = tid; k < n; k += 256)
            v += __hip_atomic_load(&out[other * n + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        carry = v * 1e-9f + 1.f;
        (void)in;
    }
    if (tid == 0) buf0[wg * n] = carry;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 16, G = argc > 2 ? atoi(argv[2]) : 256, bytes = argc > 3 ? atoi(argv[3]) : 1024;
    const int n = bytes / 4 > 0 ? bytes / 4 : 1;
    float *a, *b; unsigned *counter, *failed;
    CK(hipMalloc(&a, (size_t)G * n * 4)); CK(hipMalloc(&b, (size_t)G * n * 4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&failed, 4));
    CK(hipMemset(a, 0, (size_t)G * n * 4)); CK(hipMemset(b, 0, (size_t)G * n * 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // ---- (a) graph of dependent nodes
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(small_step, dim3(G), dim3(256), 0, st, (s & 1) ? a : b, (s & 1) ? b : a, n);
    CK(hipStreamEndCapture(st, &graph)); CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    const int reps = 50;
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("(a) hipGraph of %d dependent nodes, %d workgroups x %d bytes each: %.2f us per graph, %.2f us per node\n", steps, G, bytes, ms * 1e3f / reps, ms * 1e3f / reps / steps);
    // eager launches of the same chain, for comparison
    for (int w = 0; w < 3; ++w) for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(small_step, dim3(G), dim3(256), 0, st, (s & 1) ? a : b, (s & 1) ? b : a, n);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(small_step, dim3(G), dim3(256), 0, st, (s & 1) ? a : b, (s & 1) ? b : a, n);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("    the same chain as eager launches on one stream: %.2f us per chain, %.2f us per launch\n", ms * 1e3f / reps, ms * 1e3f / reps / steps);
    // ---- (b) one persistent launch with grid-wide barriers
    for (int pass = 0; pass < 2; ++pass) {
        const int nsteps = pass == 0 ? steps : 4 * steps;          // two lengths: the slope is the cost of a step, the intercept the launch
        float tot = 0.f; unsigned bad = 0;
        for (int r = 0; r < reps + 3; ++r) {
            CK(hipMemsetAsync(counter, 0, 4, st)); CK(hipMemsetAsync(failed, 0, 4, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(persistent_steps, dim3(G), dim3(256), 0, st, a, b, counter, failed, nsteps, n, G);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 3) tot += ms;
            unsigned f = 0; CK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost)); bad |= f;
        }
        printf("(b) one launch, %d grid-wide barriers (write-through stores + atomic arrival + uncached poll + uncached read of another workgroup's %d bytes): "
               "%.2f us per launch, %.2f us per step%s\n", nsteps, bytes, tot * 1e3f / reps, tot * 1e3f / reps / nsteps, bad ? "   [SPIN LIMIT HIT: not all workgroups were resident]" : "");
    }
    return 0;
}
