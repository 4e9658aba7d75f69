// Pairwise squared distance in feature space, the F-FPS distance matrix of
// lib/utils/model_util.py:144-160 (calc_square_dist, norm=False):
//     out[b,i,j] = (|a_i|^2 + |b_j|^2) - 2 * (a_i . b_j)
// with |.|^2 and the dot product as fmaf chains over channels ascending from 0 (decision E of
// oracle/sa_oracle.c -- TensorFlow's own summation order is cuBLAS's and cannot be pinned).
//
// Row vectors may be given in two pieces [x0 | x1] (c0 + c1 channels) so that the SA layer's
// concat([xyz, features]) (layers_util.py:94,102) never has to be materialised.
//
// Main kernel: fp32-input MFMA (v_mfma_f32_32x32x2_f32).  On gfx950 that instruction is bitwise a
// k-ordered fp32 fmaf chain (one rounding per product, no wider accumulation; cdna_hip_programming.md
// section 3), so the matrix-core result is bit-identical to the oracle's scalar chain while running
// at the matrix pipe's fp32 rate.  128x128 output tile per 256-thread workgroup, 2x2 MFMA tiles per
// wave, operands staged k-major through LDS.  A plain VALU kernel (64x64 tile, 4x4 outputs per thread)
// is kept as the cross-check (SA_SQDIST_VALU=1); tests require the two to agree bit for bit.
#include <stdlib.h>

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

// ---- fp32 MFMA kernel ---------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMT = 128;   // tile edge
constexpr int kMK = 16;    // channels per LDS stage
constexpr int kMLd = 132;  // padded leading dimension

// SYM: a == b (the F-FPS case).  Only tiles with bj >= bi are computed; an off-diagonal tile is also
// written transposed (through a per-wave LDS patch, so both copies are coalesced).  The matrix is bitwise
// symmetric by construction: dot(i,j) and dot(j,i) are the same products in the same channel order, and
// the norm sum is a commutative fp32 add.
template <bool SYM>
__global__ __launch_bounds__(256) void sqdist_mfma_kernel(int n, int m, RowSrc A, RowSrc Bm,
                                                          float *__restrict__ out) {
    __shared__ float As[kMK][kMLd];
    __shared__ float Bs[kMK][kMLd];
    __shared__ float sA[kMT], sB[kMT];
    __shared__ float sT[SYM ? 4 : 1][SYM ? 64 : 1][SYM ? 33 : 1];   // per-wave transpose patch: 64 cols x 32 rows
    const int b = blockIdx.z;
    int bi = blockIdx.y, bj = blockIdx.x;
    if (SYM) {                                    // linear id over the upper triangle (row-major)
        const int T = (n + kMT - 1) / kMT;
        int rem = blockIdx.x;
        bi = 0;
        while (rem >= T - bi) { rem -= T - bi; ++bi; }
        bj = bi + rem;
    }
    const int i0 = bi * kMT, j0 = bj * kMT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int half = lane >> 5, col = lane & 31;
    const int c = A.c0 + A.c1;

    f32x16 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.0f;
    float nrm = 0.0f;   // |a_i|^2 (threads 0..127) or |b_j|^2 (threads 128..255), channel-ascending chain

    const int lrow = tid >> 1, lk = (tid & 1) * 8;
    const long ga = (long)b * n + min(i0 + lrow, n - 1);
    const long gb = (long)b * m + min(j0 + lrow, m - 1);
    float pa[8], pb[8];                            // next stage, fetched while the current one is multiplied
#pragma unroll
    for (int e = 0; e < 8; ++e) { pa[e] = load_ch(A, ga, lk + e); pb[e] = load_ch(Bm, gb, lk + e); }
    for (int k0 = 0; k0 < c; k0 += kMK) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { As[lk + e][lrow] = pa[e]; Bs[lk + e][lrow] = pb[e]; }
        __syncthreads();
        if (k0 + kMK < c) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pa[e] = load_ch(A, ga, k0 + kMK + lk + e);     // channels >= c read as 0
                pb[e] = load_ch(Bm, gb, k0 + kMK + lk + e);
            }
        }
        const int kend = min(kMK, c - k0);
        {
            const float(*S)[kMLd] = tid < kMT ? As : Bs;
            const int t = tid & (kMT - 1);
            for (int kk = 0; kk < kend; ++kk) nrm = __builtin_fmaf(S[kk][t], S[kk][t], nrm);
        }
#pragma unroll
        for (int ks = 0; ks < kMK / 2; ++ks) {
            if (ks * 2 < kend) {
                const int kk = ks * 2 + half;
                float a[2], bb[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[t] = As[kk][wr * 64 + t * 32 + col];
                    bb[t] = Bs[kk][wc * 64 + t * 32 + col];
                }
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti], bb[tj], acc[ti][tj], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (tid < kMT) sA[tid] = nrm; else sB[tid - kMT] = nrm;
    __syncthreads();
    const bool mirror = SYM && bi != bj;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int jl = wc * 64 + tj * 32 + col;
            const int j = j0 + jl;
            const float sb = sB[jl];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ir = (r & 3) + 8 * (r >> 2) + 4 * half;        // row inside the 32-row tile
                const int il = wr * 64 + ti * 32 + ir;
                const int i = i0 + il;
                const float v = (sA[il] + sb) - 2.0f * acc[ti][tj][r];
                if (i < n && j < m) out[((size_t)b * n + i) * m + j] = v;
                if (SYM) sT[w][tj * 32 + col][ir] = v;
            }
        }
        if (mirror) {                              // out[j][i] for rows j of this wave's patch, 32 columns i
            for (int r2 = 0; r2 < 32; ++r2) {
                const int r = r2 * 2 + half;       // two patch rows per step, 32 lanes (128 B) each
                const int j = j0 + wc * 64 + r, i = i0 + wr * 64 + ti * 32 + col;
                if (j < n && i < n) out[((size_t)b * n + j) * n + i] = sT[w][r][col];
            }
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
    static const bool use_valu = getenv("SA_SQDIST_VALU") && atoi(getenv("SA_SQDIST_VALU")) != 0;
    if (use_valu) {
        dim3 grid((m + kT - 1) / kT, (n + kT - 1) / kT, b);
        hipLaunchKernelGGL(sqdist_kernel, grid, dim3(256), 0, stream, n, m, A, Bm, out);
    } else if (n == m && a0 == b0 && a1 == b1) {   // the F-FPS case: symmetric, upper triangle only
        const int T = (n + kMT - 1) / kMT;
        dim3 grid(T * (T + 1) / 2, 1, b);
        hipLaunchKernelGGL(sqdist_mfma_kernel<true>, grid, dim3(256), 0, stream, n, m, A, Bm, out);
    } else {
        dim3 grid((m + kMT - 1) / kMT, (n + kMT - 1) / kMT, b);
        hipLaunchKernelGGL(sqdist_mfma_kernel<false>, grid, dim3(256), 0, stream, n, m, A, Bm, out);
    }
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// model_util.calc_square_dist(a, b, norm=False): a [bs,n,c], bb [bs,m,c] -> [bs,n,m].
extern "C" int sa_calc_square_dist(int b, int n, int m, int c, const float *a, const float *bb,
                                   float *out, hipStream_t stream) {
    return sa_calc_square_dist_split(b, n, m, c, 0, a, nullptr, bb, nullptr, out, stream);
}
