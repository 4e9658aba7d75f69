// prob_sample (lib/utils/tf_ops/sampling/tf_sampling_g.cu:24-121,385-388): inverse-CDF sampling -- a cumulative sum of
// the per-row weights, then for every uniform number r the first position whose cumulative weight reaches r * total.
// The positions returned depend on the exact float values of the cumulative sum, so the reference's summation order
// is kept: 8192-element chunks; inside a chunk the four-element groups are summed as (v1, v1+v2, v3+(v1+v2),
// (v4+v3)+(v1+v2)), the group totals go through an up-sweep / down-sweep tree (Blelloch) over a bank-padded array,
// and the chunks are chained with a compensated running sum.  Only the thread <-> element assignment differs (one
// 1024-thread workgroup per row here, 512 threads striding over rows there); the tree's operations are the same set,
// each on its own operands, so every float comes out identical.
#include "sa_common.h"

namespace {

constexpr int kPsBlock = 2048;   // group totals per chunk (BlockSize of the reference)
constexpr int kPsPad = 5;        // paddingLevel
constexpr int kPsThreads = 1024;

__global__ __launch_bounds__(kPsThreads) void cumsum_kernel(int n, const float *__restrict__ inp, float *__restrict__ out) {
    __shared__ float buffer4[kPsBlock * 4];
    __shared__ float buffer[kPsBlock + (kPsBlock >> kPsPad)];
    const float *row = inp + (size_t)blockIdx.x * n;
    float *orow = out + (size_t)blockIdx.x * n;
    const int t = threadIdx.x;
    float runningsum = 0.0f, runningsum2 = 0.0f;
    for (int j = 0; j < n; j += kPsBlock * 4) {
        const int n24_i = min(n - j, kPsBlock * 4);
        const int n24 = (n24_i + 3) & ~3;
        const int n2 = n24 >> 2;
        for (int k = t * 4; k < n24_i; k += kPsThreads * 4) {
            if (k + 3 < n24_i) {
                float v1 = row[j + k], v2 = row[j + k + 1];
                v2 += v1;
                float v3 = row[j + k + 2], v4 = row[j + k + 3];
                v4 += v3;
                v3 += v2;
                v4 += v2;
                buffer4[k] = v1; buffer4[k + 1] = v2; buffer4[k + 2] = v3; buffer4[k + 3] = v4;
                buffer[(k >> 2) + (k >> (2 + kPsPad))] = v4;
            } else {
                float v = 0.0f;
                for (int k2 = k; k2 < n24_i; ++k2) { v += row[j + k2]; buffer4[k2] = v; }
                for (int k2 = n24_i; k2 < n24; ++k2) buffer4[k2] = v;
                buffer[(k >> 2) + (k >> (2 + kPsPad))] = v;
            }
        }
        int u = 0;
        for (; (2 << u) <= n2; ++u) {
            __syncthreads();
            for (int k = t; k < (n2 >> (u + 1)); k += kPsThreads) {
                int i1 = (((k << 1) + 2) << u) - 1, i2 = (((k << 1) + 1) << u) - 1;
                i1 += i1 >> kPsPad; i2 += i2 >> kPsPad;
                buffer[i1] += buffer[i2];
            }
        }
        --u;
        for (; u >= 0; --u) {
            __syncthreads();
            for (int k = t; k < ((n2 - (1 << u)) >> (u + 1)); k += kPsThreads) {
                int i1 = (((k << 1) + 3) << u) - 1, i2 = (((k << 1) + 2) << u) - 1;
                i1 += i1 >> kPsPad; i2 += i2 >> kPsPad;
                buffer[i1] += buffer[i2];
            }
        }
        __syncthreads();
        for (int k = t * 4; k < n24; k += kPsThreads * 4) {
            if (k != 0) {
                const int k2 = ((k >> 2) - 1) + (((k >> 2) - 1) >> kPsPad);
                const float add = buffer[k2];
                buffer4[k] += add; buffer4[k + 1] += add; buffer4[k + 2] += add; buffer4[k + 3] += add;
            }
        }
        __syncthreads();
        for (int k = t; k < n24_i; k += kPsThreads) orow[j + k] = buffer4[k] + runningsum;
        const float tt = buffer[(n2 - 1) + ((n2 - 1) >> kPsPad)] + runningsum2;
        const float r2 = runningsum + tt;
        runningsum2 = tt - (r2 - runningsum);
        runningsum = r2;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void binary_search_kernel(int n, int m, const float *__restrict__ dataset,
                                                            const float *__restrict__ query, int *__restrict__ result) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    int base = 1;
    while (base < n) base <<= 1;
    const float *d = dataset + (size_t)i * n;
    const float q = query[(size_t)i * m + j] * d[n - 1];
    int r = n - 1;
    for (int k = base; k >= 1; k >>= 1)
        if (r >= k && d[r - k] >= q) r -= k;
    result[(size_t)i * m + j] = r;
}

}  // namespace

// probsampleLauncher(b,n,m,inp_p,inp_r,temp,out) -- tf_sampling.cpp:102.  inp_p [b,n] weights, inp_r [b,m] uniform
// numbers, temp [b,n] (receives the cumulative sums), out [b,m] int.
extern "C" int sa_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out,
                              hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || !inp_p || !inp_r || !temp || !out) return SA_ERR_INVALID;
    if (b > 65535) return SA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(cumsum_kernel, dim3(b), dim3(kPsThreads), 0, stream, n, inp_p, temp);
    SA_CHECK_LAUNCH();
    hipLaunchKernelGGL(binary_search_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)b), dim3(256), 0, stream, n, m,
                       temp, inp_r, out);
    SA_CHECK_LAUNCH();
    return SA_OK;
}
