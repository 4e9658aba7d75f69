// What makes a dependent launch cost ~130 us when 16 steps of the backbone are in flight (tools/stage_trace.py)?
// V "victim" streams each replay a graph of K tiny dependent kernels; A "aggressor" streams replay chains of one of:
//   few         8 workgroups x 1024 threads spinning 3 ms            (the D-FPS shape: long, 8 CUs)
//   wide        8192 workgroups x 256 threads spinning 10 us each     (4 rounds over the chip: a tiled kernel)
//   persistent  2048 workgroups x 256 threads spinning 40 us each     (the same work, every workgroup resident at once)
//   wide_lds    8192 workgroups x 256 threads, 64 KB LDS, 10 us each  (2 per CU: 16 rounds)
// Reported: the victims' time per tiny kernel under each aggressor kind.
// build: hipcc --offload-arch=gfx950 -O2 dispatch_contention.hip -o dispatch_contention ; run: GPU_MAX_HW_QUEUES=16 ./dispatch_contention
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void tiny(float *p) { if (threadIdx.x == 0) p[blockIdx.x] = p[blockIdx.x] * 1.0001f + 1.0f; }
__global__ void spin(float *p, long ticks) {
    extern __shared__ float lds[];
    const long t0 = wall_clock64();
    while ((long)wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f;
}

struct Kind { const char *name; int grid, block, lds; long us; int reps; };

int main(int argc, char **argv) {
    const int K = 100, V = argc > 1 ? atoi(argv[1]) : 8, A = argc > 2 ? atoi(argv[2]) : 8;
    CK(hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const Kind kinds[] = {{"none", 0, 0, 0, 0, 0},        {"few", 8, 1024, 0, 3000, 8},     {"wide", 8192, 256, 0, 10, 400},
                          {"persistent", 2048, 256, 0, 40, 400}, {"wide_lds", 8192, 256, 65536, 10, 120}, {"wide_1round", 2048, 256, 0, 10, 1200}};
    std::vector<hipStream_t> st(V + A);
    std::vector<float *> buf(V + A);
    for (int i = 0; i < V + A; ++i) { CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); CK(hipMalloc(&buf[i], 4096)); CK(hipMemset(buf[i], 0, 4096)); }
    std::vector<hipGraphExec_t> vg(V);
    for (int i = 0; i < V; ++i) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(st[i], hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(tiny, dim3(8), dim3(64), 0, st[i], buf[i]);
        CK(hipStreamEndCapture(st[i], &g));
        CK(hipGraphInstantiate(&vg[i], g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(vg[i], st[i]));
    }
    CK(hipDeviceSynchronize());
    for (const Kind &kd : kinds) {
        std::vector<hipGraphExec_t> ag(A);
        if (kd.grid) {
            for (int i = 0; i < A; ++i) {
                hipGraph_t g;
                hipStream_t s = st[V + i];
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (int k = 0; k < kd.reps; ++k) hipLaunchKernelGGL(spin, dim3(kd.grid), dim3(kd.block), kd.lds, s, buf[V + i], kd.us * 100);
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ag[i], g, nullptr, nullptr, 0));
            }
        }
        hipEvent_t a0, a1;
        CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
        std::vector<hipEvent_t> ea(V), eb(V);
        for (auto &e : ea) CK(hipEventCreate(&e));
        for (auto &e : eb) CK(hipEventCreate(&e));
        CK(hipDeviceSynchronize());
        if (kd.grid) {
            CK(hipEventRecord(a0, st[V]));
            for (int i = 0; i < A; ++i) CK(hipGraphLaunch(ag[i], st[V + i]));
            CK(hipEventRecord(a1, st[V]));
        }
        for (int r = 0; r < 3; ++r)
            for (int i = 0; i < V; ++i) {
                if (r == 1) CK(hipEventRecord(ea[i], st[i]));
                CK(hipGraphLaunch(vg[i], st[i]));
                if (r == 1) CK(hipEventRecord(eb[i], st[i]));
            }
        for (int i = 0; i < V; ++i) CK(hipStreamSynchronize(st[i]));
        const bool still = kd.grid && hipEventQuery(a1) == hipErrorNotReady;
        CK(hipDeviceSynchronize());
        double t = 0;
        for (int i = 0; i < V; ++i) { float ms; CK(hipEventElapsedTime(&ms, ea[i], eb[i])); t += ms; }
        t /= V;
        float ams = 0;
        if (kd.grid) CK(hipEventElapsedTime(&ams, a0, a1));
        printf("%-12s victims %d x %d tiny kernels: %8.2f us per kernel | aggressors %d x %d kernels: %.1f us per kernel%s\n", kd.name, V, K,
               1e3 * t / K, A, kd.reps, kd.reps ? 1e3 * ams / kd.reps : 0.0, kd.grid ? (still ? " (still running at the end: ok)" : " (FINISHED EARLY)") : "");
        if (kd.grid) for (int i = 0; i < A; ++i) CK(hipGraphExecDestroy(ag[i]));
    }
    return 0;
}
