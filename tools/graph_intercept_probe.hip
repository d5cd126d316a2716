// Standalone reproducer for the SIGSEGV recorded in gpurun_out/sys.log (round 2) and gpurun_out/r3a/sys.log (round 3):
// a host fault inside librocprofiler-sdk.so (AQL packet walk of its queue-write interceptor), reached from
// hipGraphLaunch -> libamdhip64 -> libhsa-runtime64, under `rocprofv3 --kernel-trace`.  Nothing of libfemfct is
// involved here: one trivial kernel, captured G times into a graph, launched until P packets have gone through
// the (intercepted) queue.
//   hipcc -O2 --offload-arch=gfx950 tools/graph_intercept_probe.hip -o tools/graph_intercept_probe
//   rocprofv3 --kernel-trace --stats -d /tmp/gip -- tools/graph_intercept_probe 650 40000 [pre]
// `pre` plain launches go through the stream before the first graph launch (the library's sweeps enqueue a few
// ordinary kernels between graphs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Fat { double a[24]; const double* p[8]; int k[8]; };   // a by-value argument block like LoadSpec / ChebIO

__global__ void k_touch(int* counter, Fat f, int step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(counter, 1 + (f.k[0] & 0) + (step & 0));
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 650;          // kernel nodes per graph
    const long P = argc > 2 ? atol(argv[2]) : 40000;       // packets to push in total
    const int pre = argc > 3 ? atoi(argv[3]) : 0;
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    int* d = nullptr;
    CHECK(hipMalloc((void**)&d, sizeof(int)));
    CHECK(hipMemsetAsync(d, 0, sizeof(int), s));
    Fat f{};
    for (int i = 0; i < pre; ++i) hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, d, f, i);
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < G; ++i) hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s, d, f, i);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphDestroy(g));
    long sent = pre;
    int launches = 0;
    while (sent < P) {
        CHECK(hipGraphLaunch(ge, s));
        sent += G;
        ++launches;
        if (launches % 8 == 0) { printf("  %d graph launches, %ld packets\n", launches, sent); fflush(stdout); }
    }
    CHECK(hipStreamSynchronize(s));
    int h = 0;
    CHECK(hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost));
    printf("G=%d: %d graph launches, %d kernels ran (expected %ld): %s\n", G, launches, h, sent, h == sent ? "OK" : "MISMATCH");
    CHECK(hipGraphExecDestroy(ge));
    return h == sent ? 0 : 2;
}
