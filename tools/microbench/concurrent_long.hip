// How many long kernels run at the same time?  N streams, one 2 ms kernel each (8 workgroups x 512 threads, the layer-1
// D-FPS shape, with / without its 70 KB of LDS): wall time of the N launches.  N x 2 ms / wall = kernels in flight.
// build: hipcc --offload-arch=gfx950 -O2 concurrent_long.hip -o concurrent_long ; run: GPU_MAX_HW_QUEUES=16 ./concurrent_long
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
int main() {
    CK(hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    const int NS = 32;
    std::vector<hipStream_t> st(NS);
    float *buf;
    CK(hipMalloc(&buf, 4096));
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int lds : {0, 72 * 1024})
        for (int wgs : {8, 64})
            for (int N : {1, 2, 4, 6, 8, 10, 12, 16, 24, 32}) {
                for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(wgs), dim3(512), lds, st[i], buf, 1000);   // warm
                CK(hipDeviceSynchronize());
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(wgs), dim3(512), lds, st[i], buf, 200000);  // 2 ms
                CK(hipDeviceSynchronize());
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                printf("lds %5d  wgs %2d  streams %2d: wall %.2f ms -> %.1f kernels in flight\n", lds, wgs, N, ms, N * 2.0 / ms);
            }
    return 0;
}
