// Pairwise squared distance in feature space, the F-FPS distance matrix of
// lib/utils/model_util.py:144-160 (calc_square_dist, norm=False):
//     out[b,i,j] = (|a_i|^2 + |b_j|^2) - 2 * (a_i . b_j)
// with |.|^2 and the dot product as fmaf chains over channels ascending from 0 (decision E of
// oracle/sa_oracle.c -- TensorFlow's own summation order is cuBLAS's and cannot be pinned).
//
// Row vectors may be given in two pieces [x0 | x1] (c0 + c1 channels) so that the SA layer's
// concat([xyz, features]) (layers_util.py:94,102) never has to be materialised.
//
// v1: fp32 VALU, 64x64 output tile per 256-thread workgroup, 4x4 outputs per thread, operands
// staged k-major through LDS.  Every output's chain runs over channels in ascending order, so the
// result is bit-identical to the oracle.
#include "sa_common.h"

namespace {

constexpr int kT = 64;    // tile edge
constexpr int kKC = 16;   // channels per LDS stage
constexpr int kLd = 68;   // padded leading dimension of a k-major stage row

struct RowSrc {
    const float *p0; int c0;   // first piece  [rows, c0]
    const float *p1; int c1;   // second piece [rows, c1] (may be null, c1 = 0)
};

__device__ __forceinline__ float load_ch(const RowSrc &s, long row, int ch) {
    if (ch < s.c0) return s.p0[row * s.c0 + ch];
    ch -= s.c0;
    if (ch < s.c1) return s.p1[row * s.c1 + ch];
    return 0.0f;
}

__global__ __launch_bounds__(256) void sqdist_kernel(int n, int m, RowSrc A, RowSrc Bm,
                                                     float *__restrict__ out) {
    __shared__ float As[kKC][kLd];
    __shared__ float Bs[kKC][kLd];
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * kT, j0 = blockIdx.x * kT;
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;        // 16 x 16 threads, 4 x 4 outputs each
    const int c = A.c0 + A.c1;

    float acc[4][4] = {};
    float sa[4] = {}, sb[4] = {};

    const int lrow = tid >> 2, lk = (tid & 3) * 4; // stage loader: row 0..63, 4 channels
    for (int k0 = 0; k0 < c; k0 += kKC) {
        const long ga = (long)b * n + min(i0 + lrow, n - 1);
        const long gb = (long)b * m + min(j0 + lrow, m - 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            As[lk + e][lrow] = load_ch(A, ga, k0 + lk + e);   // channels >= c read as 0
            Bs[lk + e][lrow] = load_ch(Bm, gb, k0 + lk + e);
        }
        __syncthreads();
        const int kend = min(kKC, c - k0);
        for (int kk = 0; kk < kend; ++kk) {
            const float4 av = *(const float4 *)&As[kk][ty * 4];
            const float4 bv = *(const float4 *)&Bs[kk][tx * 4];
            const float a[4] = {av.x, av.y, av.z, av.w};
            const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sa[r] = __builtin_fmaf(a[r], a[r], sa[r]);
                sb[r] = __builtin_fmaf(bb[r], bb[r], sb[r]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[r][q] = __builtin_fmaf(a[r], bb[q], acc[r][q]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty * 4 + r;
        if (i >= n) continue;
        float *o = out + ((size_t)b * n + i) * m + j0 + tx * 4;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (sa[r] + sb[q]) - 2.0f * acc[r][q];
        if (j0 + tx * 4 + 3 < m && (m % 4 == 0)) {
            *(float4 *)o = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (j0 + tx * 4 + q < m) o[q] = v[q];
        }
    }
}

}  // namespace

// a = [a0 | a1] rows [b,n,c0+c1], bb = [b0 | b1] rows [b,m,c0+c1]; out [b,n,m].
extern "C" int sa_calc_square_dist_split(int b, int n, int m, int c0, int c1, const float *a0,
                                         const float *a1, const float *b0, const float *b1, float *out,
                                         hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || c0 <= 0 || c1 < 0 || !a0 || !b0 || !out) return SA_ERR_INVALID;
    if (c1 > 0 && (!a1 || !b1)) return SA_ERR_INVALID;
    RowSrc A{a0, c0, a1, c1}, Bm{b0, c0, b1, c1};
    dim3 grid((m + kT - 1) / kT, (n + kT - 1) / kT, b);
    hipLaunchKernelGGL(sqdist_kernel, grid, dim3(256), 0, stream, n, m, A, Bm, out);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// model_util.calc_square_dist(a, b, norm=False): a [bs,n,c], bb [bs,m,c] -> [bs,n,m].
extern "C" int sa_calc_square_dist(int b, int n, int m, int c, const float *a, const float *bb,
                                   float *out, hipStream_t stream) {
    return sa_calc_square_dist_split(b, n, m, c, 0, a, nullptr, bb, nullptr, out, stream);
}
