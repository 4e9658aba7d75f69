// Cost of a dependent kernel boundary with S chains in flight, by how the chains are issued:
//   graph   one hipGraph (K-kernel linear chain) per stream, replayed from one host thread
//   eager   the same chains launched kernel by kernel, one host thread per stream
// build: hipcc --offload-arch=gfx950 -O2 -pthread dispatch_queues.hip -o dispatch_queues ; run: ./dispatch_queues [K] [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <thread>
#include <vector>

__global__ void tiny(float *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 50, reps = argc > 2 ? atoi(argv[2]) : 6;
    const int n = 8 * 256 * 3;
    for (int S : {1, 4, 8, 12, 16, 24}) {
        std::vector<hipStream_t> st(S);
        std::vector<float *> buf(S);
        for (int i = 0; i < S; ++i) { CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); CK(hipMalloc(&buf[i], n * 4)); CK(hipMemset(buf[i], 0, n * 4)); }
        // ---- graphs
        std::vector<hipGraphExec_t> ge(S);
        for (int i = 0; i < S; ++i) {
            hipGraph_t g;
            CK(hipStreamBeginCapture(st[i], hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < K; ++k) hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, st[i], buf[i], n);
            CK(hipStreamEndCapture(st[i], &g));
            CK(hipGraphInstantiate(&ge[i], g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge[i], st[i]));
        }
        CK(hipDeviceSynchronize());
        std::vector<hipEvent_t> ea(S * reps), eb(S * reps);
        for (auto &e : ea) CK(hipEventCreate(&e));
        for (auto &e : eb) CK(hipEventCreate(&e));
        for (int r = 0; r < reps; ++r)
            for (int i = 0; i < S; ++i) {
                CK(hipEventRecord(ea[r * S + i], st[i]));
                CK(hipGraphLaunch(ge[i], st[i]));
                CK(hipEventRecord(eb[r * S + i], st[i]));
            }
        CK(hipDeviceSynchronize());
        double tg = 0;
        for (int j = 0; j < S * reps; ++j) { float ms; CK(hipEventElapsedTime(&ms, ea[j], eb[j])); tg += ms; }
        tg /= S * reps;
        // ---- eager, one host thread per stream
        auto worker = [&](int i, int r) {
            CK(hipEventRecord(ea[r * S + i], st[i]));
            for (int k = 0; k < K; ++k) hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, st[i], buf[i], n);
            CK(hipEventRecord(eb[r * S + i], st[i]));
        };
        double te = 0;
        for (int r = 0; r < reps; ++r) {
            std::vector<std::thread> th;
            for (int i = 0; i < S; ++i) th.emplace_back(worker, i, r);
            for (auto &t : th) t.join();
        }
        CK(hipDeviceSynchronize());
        for (int j = 0; j < S * reps; ++j) { float ms; CK(hipEventElapsedTime(&ms, ea[j], eb[j])); te += ms; }
        te /= S * reps;
        printf("streams %2d: graph chain %.3f ms (%.2f us/kernel) | eager threads chain %.3f ms (%.2f us/kernel)\n", S, tg,
               1e3 * tg / K, te, 1e3 * te / K);
        for (int i = 0; i < S; ++i) { CK(hipGraphExecDestroy(ge[i])); CK(hipFree(buf[i])); CK(hipStreamDestroy(st[i])); }
    }
    return 0;
}
