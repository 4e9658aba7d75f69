// Row gathers for gfx950: gather_point (tf_sampling_g.cu:320-331) and group_point
// (tf_grouping_g.cu:362-379).  Both are pure HBM traffic: the output is written once, fully
// coalesced (16 B per lane when the channel count allows), and the source rows are read in
// row-sized contiguous pieces.  The reference uses one thread per scalar on a fixed 512x64 grid.
#include "sa_common.h"

namespace {

// out[r, :] = src[batch(r), idx[r], :]   with r over b*rows_per_batch rows of c floats.
// NEG1_ZERO: idx == -1 produces a zero row (group_point, tf_grouping_g.cu:373-375).
template <typename VT, bool NEG1_ZERO>
__global__ __launch_bounds__(256) void gather_rows_kernel(long total_vec, int vec_per_row, int n,
                                                          long rows_per_batch,
                                                          const VT *__restrict__ src,
                                                          const int *__restrict__ idx,
                                                          VT *__restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total_vec;
         i += (long)gridDim.x * blockDim.x) {
        const long r = i / vec_per_row;
        const int v = (int)(i - r * vec_per_row);
        const long bi = r / rows_per_batch;
        const int a = idx[r];
        VT val;
        if (NEG1_ZERO && a == -1) {
            val = VT{};
        } else {
            val = src[((size_t)bi * n + a) * vec_per_row + v];
        }
        out[i] = val;
    }
}

template <bool NEG1_ZERO>
int launch_gather(int b, int n, int c, long rows_per_batch, const float *src, const int *idx, float *out,
                  hipStream_t stream) {
    const long rows = (long)b * rows_per_batch;
    const bool v4 = (c % 4 == 0) && (((uintptr_t)src | (uintptr_t)out) % 16 == 0);
    const int vpr = v4 ? c / 4 : c;
    const long total = rows * vpr;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (v4)
        hipLaunchKernelGGL((gather_rows_kernel<float4, NEG1_ZERO>), dim3((unsigned)blocks), dim3(256), 0,
                           stream, total, vpr, n, rows_per_batch, (const float4 *)src, idx, (float4 *)out);
    else
        hipLaunchKernelGGL((gather_rows_kernel<float, NEG1_ZERO>), dim3((unsigned)blocks), dim3(256), 0,
                           stream, total, vpr, n, rows_per_batch, src, idx, out);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

}  // namespace

// lib/utils/tf_ops/sampling/tf_sampling.cpp:235  gatherpointLauncher(b,n,m,c,inp,idx,out)
extern "C" int sa_gather_point(int b, int n, int m, int c, const float *inp, const int *idx, float *out,
                               hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || c <= 0 || !inp || !idx || !out) return SA_ERR_INVALID;
    return launch_gather<false>(b, n, c, m, inp, idx, out, stream);
}

// lib/utils/tf_ops/grouping/tf_grouping.cpp:446  groupPointLauncher(b,n,c,m,nsample,points,idx,out)
extern "C" int sa_group_point(int b, int n, int c, int m, int nsample, const float *points,
                              const int *idx, float *out, hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || c <= 0 || nsample <= 0 || !points || !idx || !out)
        return SA_ERR_INVALID;
    return launch_gather<true>(b, n, c, (long)m * nsample, points, idx, out, stream);
}
