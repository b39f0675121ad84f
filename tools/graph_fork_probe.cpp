// graph_fork_probe.cpp -- do the two branches of a captured fork-join hipGraph run side by side on this box (measurement aid, round 5)?
// Stream A and stream B each get a chain of `n` spin kernels of `us` microseconds (64 workgroups each: the chip has room for both);
//   eager:  the two chains on two streams, no graph                      -> wall ~ n * us when the streams overlap
//   graph:  A forks to B through an event inside one stream capture, joins -> the same when graph branches overlap, 2 n * us when they are serialised
//   chain:  both chains on one stream (the serial reference)
// Build: hipcc --offload-arch=gfx950 -O3 tools/graph_fork_probe.cpp -o tools/graph_fork_probe     Run: tools/graph_fork_probe [n] [us]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

__global__ void spin(unsigned long long ticks, unsigned long long* sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned long long t = t0;
    while (t - t0 < ticks) t = wall_clock64();
    if (sink && t == 1) sink[0] = t;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4, us = argc > 2 ? atoi(argv[2]) : 100;
    int khz = 100000;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0) != hipSuccess || khz <= 0) khz = 100000;
    const unsigned long long ticks = (unsigned long long)us * (unsigned long long)khz / 1000ull;
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    auto wall = [&](auto&& body) {
        body(); hipStreamSynchronize(a); hipStreamSynchronize(b);
        double best = 1e30;
        for (int r = 0; r < 5; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            body(); hipStreamSynchronize(a); hipStreamSynchronize(b);
            const double el = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (el < best) best = el;
        }
        return best;
    };
    const double chain = wall([&] { for (int i = 0; i < 2 * n; ++i) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, ticks, nullptr); });
    const double eager = wall([&] {
        for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, ticks, nullptr); hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, b, ticks, nullptr); }
    });
    const double eager_ev = wall([&] {        // the same with the fork / join events the executor would use (A -> B before, B -> A after)
        hipEventRecord(fork, a); hipStreamWaitEvent(b, fork, 0);
        for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, ticks, nullptr); hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, b, ticks, nullptr); }
        hipEventRecord(join, b); hipStreamWaitEvent(a, join, 0);
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, 1ull, nullptr);
    });
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    CK(hipStreamBeginCapture(a, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(fork, a)); CK(hipStreamWaitEvent(b, fork, 0));
    for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, ticks, nullptr); hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, b, ticks, nullptr); }
    CK(hipEventRecord(join, b)); CK(hipStreamWaitEvent(a, join, 0));
    hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, 1ull, nullptr);
    CK(hipStreamEndCapture(a, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const double graph = wall([&] { hipGraphLaunch(ge, a); });
    printf("# graph_fork_probe: two chains of %d spin kernels of %d us\n", n, us);
    printf("one stream (serial reference)             %9.1f us\n", chain);
    printf("two streams, eager                        %9.1f us\n", eager);
    printf("two streams, eager, fork / join events    %9.1f us\n", eager_ev);
    printf("one captured fork-join graph              %9.1f us   (%s)\n", graph, graph < 1.5 * n * us ? "branches overlap" : "branches are serialised");
    return 0;
}
