// Do captured graphs run as concurrently as eager launches?  16 streams, each issues the same work containing one or
// two 2 ms kernels (8 workgroups x 512 threads, 72 KB LDS: the layer-1 D-FPS shape), as
//   eager     plain launches                         graph1    a captured graph of the one kernel
//   graph3    captured [tiny, long, tiny]            graph2L   captured [long, long]
//   forkjoin  captured [tiny] -> helper stream [long] -> [tiny]   (what layers_util mode 6 does for the FPS pair)
// wall time of one round over all streams -> long kernels in flight.
// build: hipcc --offload-arch=gfx950 -O2 graph_concurrency.hip -o graph_concurrency ; run: GPU_MAX_HW_QUEUES=16 ./graph_concurrency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(float *p, long ticks) {
    extern __shared__ float lds[];
    const long t0 = wall_clock64();
    while ((long)wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f;
}
__global__ void tiny(float *p) { if (threadIdx.x == 0) p[1 + blockIdx.x] += 1.0f; }
static float *buf;
static const int LDS = 72 * 1024;
static void lng(hipStream_t s) { hipLaunchKernelGGL(spin, dim3(8), dim3(512), LDS, s, buf, 200000); }
static void tny(hipStream_t s) { hipLaunchKernelGGL(tiny, dim3(8), dim3(64), 0, s, buf); }

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 16;
    CK(hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipMalloc(&buf, 4096));
    std::vector<hipStream_t> st(N), hs(N);
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto &s : hs) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const char *names[] = {"eager", "graph1", "graph3", "graph2L", "forkjoin", "eager2L"};
    const int nlong[] = {1, 1, 1, 2, 1, 2};
    for (int v = 0; v < 6; ++v) {
        std::vector<hipGraphExec_t> ge(N);
        const bool graph = v >= 1 && v <= 4;
        if (graph) {
            for (int i = 0; i < N; ++i) {
                hipGraph_t g;
                CK(hipStreamBeginCapture(st[i], hipStreamCaptureModeThreadLocal));
                if (v == 1) lng(st[i]);
                if (v == 2) { tny(st[i]); lng(st[i]); tny(st[i]); }
                if (v == 3) { lng(st[i]); lng(st[i]); }
                if (v == 4) {
                    hipEvent_t e1, e2;
                    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
                    tny(st[i]);
                    CK(hipEventRecord(e1, st[i])); CK(hipStreamWaitEvent(hs[i], e1, 0));
                    lng(hs[i]);
                    CK(hipEventRecord(e2, hs[i])); CK(hipStreamWaitEvent(st[i], e2, 0));
                    tny(st[i]);
                }
                CK(hipStreamEndCapture(st[i], &g));
                CK(hipGraphInstantiate(&ge[i], g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge[i], st[i]));
            }
            CK(hipDeviceSynchronize());
        }
        for (int rounds : {1, 3}) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < rounds; ++r)
                for (int i = 0; i < N; ++i) {
                    if (graph) CK(hipGraphLaunch(ge[i], st[i]));
                    else { lng(st[i]); if (v == 5) lng(st[i]); }
                }
            CK(hipDeviceSynchronize());
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("%-9s streams %2d, %d round(s): wall %7.2f ms (ideal %5.1f) -> %.1f long kernels in flight\n", names[v], N, rounds, ms,
                   2.0 * nlong[v] * rounds, N * rounds * nlong[v] * 2.0 / ms);
        }
    }
    return 0;
}
