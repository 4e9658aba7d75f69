// Debug probe for tools/stage_trace.py: one thread appends (tag, device wall clock) to a trace buffer.  Launched after
// every C-ABI call of a step (also inside captured graphs), it shows where a step spends its time when 16 steps are
// in flight -- a regime rocprofv3 cannot observe (it serialises the streams).  Not part of the product library.
#include <hip/hip_runtime.h>

__global__ void stamp_kernel(unsigned long long *buf, unsigned long long cap, unsigned long long tag) {
    const unsigned long long i = atomicAdd(buf, 1ull);
    if (i < cap) { buf[2 + 2 * i] = tag; buf[3 + 2 * i] = wall_clock64(); }
}

extern "C" int stamp(unsigned long long *buf, unsigned long long cap, unsigned long long tag, hipStream_t stream) {
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, stream, buf, cap, tag);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
