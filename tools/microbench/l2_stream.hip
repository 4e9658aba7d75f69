// What do the L2s hand to the CUs?  (the 96-row layer-4 MLP kernel and the 128-row aggregation
// kernel were declared bound by "the ~11 TB/s the L2s deliver" while MI355X_MICROARCH.md quotes ~34.5 TB/s aggregate.)
//
// Every workgroup streams an L2-resident buffer with 16-byte loads (and nothing else), several passes, in one of three
// access patterns:
//   distinct  every CU reads its OWN slice (slice > the 32 KB vector L1, sum of the slices of an XCD < its 4 MB L2):
//             all L2 channels of an XCD busy with different lines -- the guide's figure, if it is reachable at all
//   same      every CU reads the SAME buffer (1.4 MB = the weight image of the 96-row kernel) from the start, like the
//             MLP kernels walk their weights: the 32 CUs of an XCD ask for the same lines at about the same time
//   rotated   the same buffer, every workgroup starting at a different offset (de-phased walk)
// Usage: l2_stream [bytes_same=1441792] [slice_kb=96] [passes=64] [waves_per_cu=16] [loads_in_flight=8]
// Prints TB/s summed over the chip for each pattern.  Build: hipcc --offload-arch=gfx950 -O3 l2_stream.hip -o l2_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float vf4 __attribute__((ext_vector_type(4)));

// region: vf4 elements this workgroup walks, starting at `start` (wrapping inside the region), `passes` times.
template <int UN>
__global__ __launch_bounds__(256) void stream_kernel(const vf4 *__restrict__ buf, long region_elems, long wg_region_stride,
                                                     long start_stride, int passes, float *sink) {
    const vf4 *base = buf + (long)blockIdx.x * wg_region_stride;
    const long start = ((long)blockIdx.x * start_stride) % region_elems;
    vf4 acc[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) acc[u] = vf4{0.f, 0.f, 0.f, 0.f};
    const long steps = region_elems / (256 * UN);          // region_elems is a multiple of 256 * UN
    for (int p = 0; p < passes; ++p) {
        for (long s = 0; s < steps; ++s) {
            long e0 = start + s * (256 * UN) + threadIdx.x;
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                long e = e0 + u * 256;
                if (e >= region_elems) e -= region_elems;
                const vf4 v = base[e];
                acc[u] += v;
            }
        }
    }
    vf4 t = acc[0];
#pragma unroll
    for (int u = 1; u < UN; ++u) t += acc[u];
    if (t.x + t.y + t.z + t.w == 123.456f) sink[0] = 1.0f;
}

template <int UN>
static double run(const char *name, const vf4 *buf, long region_elems, long wg_stride, long start_stride, int passes, int wgs,
                  float *sink) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {                      // first run warms the L2s
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(stream_kernel<UN>, dim3(wgs), dim3(256), 0, 0, buf, region_elems, wg_stride, start_stride, passes, sink);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
    }
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)wgs * passes * (double)region_elems * 16.0;
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-34s wgs %5d  region %8.1f KB  passes %3d  UN %d  %8.3f ms  %7.2f TB/s\n", name, wgs, region_elems * 16.0 / 1024, passes, UN, ms, tbs);
    return tbs;
}

int main(int argc, char **argv) {
    long same_bytes = argc > 1 ? atol(argv[1]) : 1441792;
    long slice_kb = argc > 2 ? atol(argv[2]) : 96;
    int passes = argc > 3 ? atoi(argv[3]) : 64;
    int waves_per_cu = argc > 4 ? atoi(argv[4]) : 16;
    int un = argc > 5 ? atoi(argv[5]) : 8;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int wg_per_cu = waves_per_cu / 4 > 0 ? waves_per_cu / 4 : 1;
    const int wgs = cus * wg_per_cu;
    printf("%s: %d CUs, %d workgroups of 256 threads per CU, %d loads of 16 B in flight per lane\n", prop.name, cus, wg_per_cu, un);
    const long quantum = 256L * 8;                                       // elements: multiple of 256 * UN for UN <= 8
    long same_elems = (same_bytes / 16 + quantum - 1) / quantum * quantum;
    long slice_elems = (slice_kb * 1024 / 16 + quantum - 1) / quantum * quantum;
    // distinct: one slice per workgroup
    const size_t total = (size_t)(slice_elems * wgs > same_elems ? slice_elems * wgs : same_elems) * 16;
    vf4 *buf; float *sink;
    CHECK(hipMalloc(&buf, total)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 0, total));
    printf("distinct working set %.1f MB over the chip (%.2f MB per XCD), same-buffer %.2f MB\n", slice_elems * wgs * 16.0 / 1e6,
           slice_elems * wgs * 16.0 / 1e6 / 8, same_elems * 16.0 / 1e6);
#define RUN3(UN)                                                                                                    \
    run<UN>("distinct (own slice per workgroup)", buf, slice_elems, slice_elems, 0, passes, wgs, sink);              \
    run<UN>("same (all from offset 0)", buf, same_elems, 0, 0, passes > 8 ? passes / 8 : 1, wgs, sink);              \
    run<UN>("rotated (same buffer, de-phased)", buf, same_elems, 0, (same_elems / wgs / (256 * UN) + 1) * (256 * UN), \
            passes > 8 ? passes / 8 : 1, wgs, sink);
    if (un >= 8) { RUN3(8) } else if (un >= 4) { RUN3(4) } else { RUN3(2) }
    return 0;
}
