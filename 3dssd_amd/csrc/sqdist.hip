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
    int rs0 = 0, rs1 = 0;      // rows between consecutive frames of each piece (0: dense, the frame's own row count);
                               // honoured by the pack kernel only: a range slice of a larger tensor is read in place
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

// ---- fp32 MFMA kernel, second form (the one the F-FPS matrix takes) ---------------------------------------
// Same arithmetic as sqdist_mfma_kernel (same MFMA instruction, same k order, same norm chains, same final
// expression -> bit-identical output), reorganised around what an ablation of that kernel showed on MI355X:
// of 0.31 ms at the layer-2 shape, 0.15 ms were the operand staging (80 branchy dword loads + 80 ds_write_b32 +
// 67 ds_read_b32 per thread and tile) and 0.145 ms the epilogue (64-bit address arithmetic, an LDS norm read
// and a bounds check per element); the matrix instructions themselves ~0.02 ms.
//   * up to 68 channels (34 k-pairs) per stage -- the whole K of the layer-2 call in one stage, two barriers
//     per tile instead of ten;
//   * the second piece of the rows (the features) is read as float4, 16 consecutive lanes per 256-byte row;
//   * operands live in LDS as two PARITY PLANES per operand, plane[k & 1][row][k >> 1]: the MFMA lane
//     (row, half) needs exactly the k of parity `half`, so its 34 values are 9 ds_read_b128 (136 ds_read_b32 in
//     the first form);
//   * epilogue: full tiles take a path without bounds checks, 32-bit element offsets from one 64-bit tile base,
//     the 16 row norms of a lane fetched with 4 ds_read_b128 per row tile.
typedef float f32x4v __attribute__((ext_vector_type(4)));
#ifdef SA_SQ_TIMING
// debug build only (tools/sqdist_prof.py): phase clocks of wave 0 of every workgroup, summed
__device__ unsigned long long g_sq_prof[256][8];      // 256 slots (same-address atomics serialise: 67 000 workgroups x 7 words)
#define SQ_T0() unsigned long long t__ = __builtin_readcyclecounter(), acc__[6] = {0, 0, 0, 0, 0, 0}
#define SQ_TICK(i) { const unsigned long long n__ = __builtin_readcyclecounter(); acc__[i] += n__ - t__; t__ = n__; }
#define SQ_FLUSH() if (tid == 0) { unsigned long long *g__ = g_sq_prof[(blockIdx.x + 31u * blockIdx.y + 97u * blockIdx.z) & 255u]; for (int i__ = 0; i__ < 6; ++i__) atomicAdd(&g__[i__], acc__[i__]); atomicAdd(&g__[7], 1ull); }
#else
#define SQ_T0()
#define SQ_TICK(i)
#define SQ_FLUSH()
#endif
constexpr int kKS2 = 36;            // channels per stage (two stages for the 67-channel layer-2 call; LDS 41 KB -> 3 workgroups per CU)
constexpr int kPL = 20;             // floats per row in a parity plane (18 pairs + 2 padding; 80 B rows)

// Epilogue shared by the two second-form kernels: out = (|a|^2 + |b|^2) - 2 a.b for the wave's 64 x 64 part of the
// tile, the direct tile and (off the diagonal of a symmetric call) its mirror image.
template <bool SYM, bool NT>
__device__ __forceinline__ void sq_store_tile(float *lds, const float *sA, const float *sB, f32x16 (&acc)[2][2], int b, int n,
                                              int m, int i0, int j0, bool mirror, int w, int lane, float *__restrict__ out) {
    const int wr = w >> 1, wc = w & 1;
    const int half = lane >> 5, col = lane & 31;
    // Stores go through a per-wave LDS patch so that every lane writes 16 bytes: a (ti) strip of the wave's tile is
    // 32 rows x 64 columns; row-major in the patch it leaves as 4 rows x 256 B per store instruction, transposed
    // (64 patch rows x 32 columns) as 8 rows x 128 B for the mirrored tile.  (Straight from the accumulator layout
    // the stores are dword stores of 2 rows x 128 B.)  Edge tiles (n not a multiple of 128) take the scalar path.
    float *patch = lds + w * (32 * 68);                            // 32 x 68 floats, also viewed 64 x 34
    const bool full = i0 + kMT <= n && j0 + kMT <= m && (m & 3) == 0 && (!SYM || (n & 3) == 0);
    float *tile = out + ((size_t)b * n + i0) * m + j0;                           // (i0, j0) of this frame
    float *tileT = SYM ? out + ((size_t)b * n + j0) * n + i0 : nullptr;          // (j0, i0): the mirrored tile
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        float sa[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *(const float4 *)(sA + wr * 64 + ti * 32 + 8 * q + 4 * half);
            sa[4 * q + 0] = v.x; sa[4 * q + 1] = v.y; sa[4 * q + 2] = v.z; sa[4 * q + 3] = v.w;
        }
        float vv[2][16];
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const float sb = sB[wc * 64 + tj * 32 + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) vv[tj][r] = (sa[r] + sb) - 2.0f * acc[ti][tj][r];
        }
        if (full) {
            // ---- direct tile: patch[ir][tj*32 + col], read back as float4 rows
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * half) * 68 + tj * 32 + col] = vv[tj][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int pr = it * 4 + (lane >> 4), pc = (lane & 15) * 4;       // 4 rows x 16 lanes x 16 B
                const float4 q4 = *(const float4 *)(patch + pr * 68 + pc);
                f32x4v *dst = (f32x4v *)(tile + (unsigned)((wr * 64 + ti * 32 + pr) * m + wc * 64 + pc));
                const f32x4v qv = {q4.x, q4.y, q4.z, q4.w};
                if (NT) __builtin_nontemporal_store(qv, dst); else *dst = qv;
            }
            if (mirror) {
                // ---- mirrored tile: patch viewed as [64 columns j][34]: patchT[tj*32 + col][ir]
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) patch[(tj * 32 + col) * 34 + (r & 3) + 8 * (r >> 2) + 4 * half] = vv[tj][r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int pj = it * 8 + (lane >> 3), pi = (lane & 7) * 4;    // 8 rows x 8 lanes x 16 B
                    const float *src = patch + pj * 34 + pi;                     // 8-byte aligned rows: two float2 reads
                    const float2 lo2 = *(const float2 *)src, hi2 = *(const float2 *)(src + 2);
                    const float4 q4 = make_float4(lo2.x, lo2.y, hi2.x, hi2.y);
                    f32x4v *dst = (f32x4v *)(tileT + (unsigned)((wc * 64 + pj) * n + wr * 64 + ti * 32 + pi));
                    const f32x4v qv = {q4.x, q4.y, q4.z, q4.w};
                    if (NT) __builtin_nontemporal_store(qv, dst); else *dst = qv;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else {
            float (*sT)[34] = (float (*)[34])patch;
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const int jl = wc * 64 + tj * 32 + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ir = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int il = wr * 64 + ti * 32 + ir;
                    if (i0 + il < n && j0 + jl < m) tile[(unsigned)(il * m + jl)] = vv[tj][r];
                    if (SYM) sT[tj * 32 + col][ir] = vv[tj][r];
                }
            }
            if (mirror) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int r2 = 0; r2 < 32; ++r2) {
                    const int r = r2 * 2 + half;
                    const int jl = wc * 64 + r, il = wr * 64 + ti * 32 + col;
                    if (j0 + jl < n && i0 + il < n) tileT[(unsigned)(jl * n + il)] = sT[r][col];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
}

// Full tiles of the packed-operand kernel: the same values as sq_store_tile, with the MIRRORED tile written in 256-byte
// row pieces too.  sq_store_tile transposes a (ti) strip of 32 rows x 64 columns, whose image is 64 rows of 32 floats:
// 8 rows x 128 B per store instruction, 4.7 TB/s with nothing else running against 5.4 for the direct tile (store-only
// builds of this kernel).  Here the distances replace the accumulators in place, the direct tile leaves per (ti) strip as
// before and the mirrored tile per (tj) strip of 64 rows x 32 columns, whose image is 32 rows x 64 floats: 4 rows x 256 B
// per instruction like the direct one, and the patch is filled with 16-byte LDS writes (a lane's four consecutive
// accumulator registers are four consecutive rows i = one 16-byte piece of a transposed row).
template <bool SYM, bool NT>
__device__ __forceinline__ void sq_store_full(float *lds, const float *sA, const float *sB, f32x16 (&acc)[2][2], int b, int n,
                                              int m, int i0, int j0, bool mirror, int w, int lane, float *__restrict__ out) {
    const int wr = w >> 1, wc = w & 1;
    const int half = lane >> 5, col = lane & 31;
    float *patch = lds + w * (32 * 68);                                          // 32 x 68 floats
    float *tile = out + ((size_t)b * n + i0) * m + j0;
    float *tileT = SYM ? out + ((size_t)b * n + j0) * n + i0 : nullptr;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        float sa[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *(const float4 *)(sA + wr * 64 + ti * 32 + 8 * q + 4 * half);
            sa[4 * q + 0] = v.x; sa[4 * q + 1] = v.y; sa[4 * q + 2] = v.z; sa[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const float sb = sB[wc * 64 + tj * 32 + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = (sa[r] + sb) - 2.0f * acc[ti][tj][r];
        }
    }
    const int pr0 = lane >> 4, pc = (lane & 15) * 4;                             // read-back: 4 rows x 16 lanes x 16 B
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {                                             // direct tile: strip of 32 rows i x 64 columns j
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * half) * 68 + tj * 32 + col] = acc[ti][tj][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int pr = it * 4 + pr0;
            const float4 q4 = *(const float4 *)(patch + pr * 68 + pc);
            f32x4v *dst = (f32x4v *)(tile + (unsigned)((wr * 64 + ti * 32 + pr) * m + wc * 64 + pc));
            const f32x4v qv = {q4.x, q4.y, q4.z, q4.w};
            if (NT) __builtin_nontemporal_store(qv, dst); else *dst = qv;
        }
    }
    if (SYM && mirror) {
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {                                         // mirrored tile: strip of 64 rows i x 32 columns j
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4v v4 = {acc[ti][tj][4 * q], acc[ti][tj][4 * q + 1], acc[ti][tj][4 * q + 2], acc[ti][tj][4 * q + 3]};
                    *(f32x4v *)(patch + col * 68 + ti * 32 + 8 * q + 4 * half) = v4;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int pr = it * 4 + pr0;                                     // row j of the strip, 64 floats i
                const float4 q4 = *(const float4 *)(patch + pr * 68 + pc);
                f32x4v *dst = (f32x4v *)(tileT + (unsigned)((wc * 64 + tj * 32 + pr) * n + wr * 64 + pc));
                const f32x4v qv = {q4.x, q4.y, q4.z, q4.w};
                if (NT) __builtin_nontemporal_store(qv, dst); else *dst = qv;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// NT: non-temporal stores -- a matrix far larger than L2 + Infinity Cache (537 MB at the layer-2 shape, of which
// F-FPS later reads one row in eight) should not displace everything else (-5 %); the 8 MB layer-3 matrix is
// written normally so that the FPS kernel finds its rows on chip (non-temporal there: F-FPS +30 %).
template <bool SYM, bool NT>
__global__ __launch_bounds__(256, 4) void sqdist_mfma2_kernel(int n, int m, RowSrc A, RowSrc Bm,
                                                             float *__restrict__ out) {
    // plane[operand][parity][row][pair]; after the k loop the same memory is the per-wave transpose patches
    __shared__ __attribute__((aligned(16))) float s_pl[2][2][kMT][kPL];
    // the row norms live behind the four transpose patches in the same (by then dead) plane memory: 40 KiB of LDS
    // in total, four workgroups per CU
    static_assert(4 * 32 * 68 + 2 * kMT <= 2 * 2 * kMT * kPL, "patches + norms must fit the operand planes");
    float *sA = &s_pl[0][0][0][0] + 4 * 32 * 68, *sB = sA + kMT;
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
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int half = lane >> 5, col = lane & 31;
    const int c0 = A.c0, c1 = A.c1, c = c0 + c1;

    f32x16 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.0f;
    float nrm = 0.0f;   // |a_i|^2 (threads 0..127) or |b_j|^2 (threads 128..255), channel-ascending chain
    SQ_T0();

    for (int k0 = 0; k0 < c; k0 += kKS2) {
        const int cnt = min(kKS2, c - k0);                 // channels of this stage
        const int npairs = (cnt + 1) >> 1;
        if (k0 > 0) __syncthreads();                       // the previous stage is fully consumed
        // ---- stage the two operands: piece 0 (xyz) scalar, piece 1 (features) as float4
#pragma unroll
        for (int op = 0; op < 2; ++op) {
            const RowSrc &S = op == 0 ? A : Bm;
            const int r0 = op == 0 ? i0 : j0, nr = op == 0 ? n : m;
            float (*P0)[kPL] = s_pl[op][0], (*P1)[kPL] = s_pl[op][1];
            for (int e = tid; e < kMT * c0; e += 256) {    // piece 0
                const int row = e / c0, ch = e - row * c0;
                if (ch >= k0 && ch < k0 + cnt) {
                    const long g = (long)b * nr + min(r0 + row, nr - 1);
                    const float v = S.p0[g * c0 + ch];
                    const int kk = ch - k0;
                    ((kk & 1) ? P1 : P0)[row][kk >> 1] = v;
                }
            }
            if (c1 > 0) {                                  // piece 1: float4 q of a row covers channels c0+4q .. c0+4q+3
                const int Q = c1 >> 2;                     // (c1 % 4 == 0 checked on the host)
                const int q_lo = max(0, (k0 - c0) >> 2), q_hi = min(Q, (k0 + cnt - c0 + 3) >> 2);
                const int nq = q_hi - q_lo;                // <= 10 float4 per row and stage
                // all loads of the stage first, then the LDS stores: a rolled load -> store loop pays one full
                // memory round trip per iteration (measured: 145k of a tile's 230k cycles)
                constexpr int kMaxIt = (kMT * 10 + 255) / 256;
                float4 v4[kMaxIt];
                int rowv[kMaxIt], qv[kMaxIt];
#pragma unroll
                for (int it = 0; it < kMaxIt; ++it) {
                    const int e = tid + it * 256;
                    const bool in = e < kMT * nq;
                    const int row = in ? e / nq : 0, q = in ? q_lo + (e - row * nq) : q_lo;
                    rowv[it] = in ? row : -1;
                    qv[it] = q;
                    const long g = (long)b * nr + min(r0 + row, nr - 1);
                    v4[it] = *(const float4 *)(S.p1 + g * c1 + 4 * q);
                }
#pragma unroll
                for (int it = 0; it < kMaxIt; ++it) {
                    if (rowv[it] >= 0) {
                        const float v[4] = {v4[it].x, v4[it].y, v4[it].z, v4[it].w};
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const int kk = c0 + 4 * qv[it] + x - k0;
                            if (kk >= 0 && kk < cnt) ((kk & 1) ? P1 : P0)[rowv[it]][kk >> 1] = v[x];
                        }
                    }
                }
            }
            if (cnt & 1) {                                 // odd channel count: the last pair's second element is 0
                for (int row = tid; row < kMT; row += 256) P1[row][cnt >> 1] = 0.0f;
            }
        }
        SQ_TICK(0)
        __syncthreads();
        SQ_TICK(1)
        // ---- squared norms: one row per thread, channels ascending (even from plane 0, odd from plane 1)
        {
            const int op = tid >> 7, t = tid & (kMT - 1);
            const float *e0 = s_pl[op][0][t], *e1 = s_pl[op][1][t];
            for (int p4 = 0; p4 < npairs; p4 += 4) {
                const float4 a = *(const float4 *)(e0 + p4), d = *(const float4 *)(e1 + p4);
                const float ev[4] = {a.x, a.y, a.z, a.w}, od[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    if (p4 + x < npairs) {
                        nrm = __builtin_fmaf(ev[x], ev[x], nrm);
                        if (2 * (p4 + x) + 1 < cnt) nrm = __builtin_fmaf(od[x], od[x], nrm);
                    }
                }
            }
        }
        SQ_TICK(2)
        // ---- matrix work: four k-pairs per LDS round (one ds_read_b128 per 32-row / 32-column tile)
        const float *pa0 = s_pl[0][half][wr * 64 + col], *pa1 = s_pl[0][half][wr * 64 + 32 + col];
        const float *pb0 = s_pl[1][half][wc * 64 + col], *pb1 = s_pl[1][half][wc * 64 + 32 + col];
        for (int p4 = 0; p4 < npairs; p4 += 4) {
            const float4 a0 = *(const float4 *)(pa0 + p4), a1 = *(const float4 *)(pa1 + p4);
            const float4 b0 = *(const float4 *)(pb0 + p4), b1 = *(const float4 *)(pb1 + p4);
            const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
            const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                if (p4 + x < npairs) {
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                        for (int tj = 0; tj < 2; ++tj)
                            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ti][x], bv[tj][x], acc[ti][tj], 0, 0, 0);
                }
            }
        }
    }
    SQ_TICK(3)
    __syncthreads();                               // operand planes dead -> norms and transpose patches
    if (tid < kMT) sA[tid] = nrm; else sB[tid - kMT] = nrm;
    __syncthreads();

    sq_store_tile<SYM, NT>(&s_pl[0][0][0][0], sA, sB, acc, b, n, m, i0, j0, SYM && bi != bj, w, lane, out);
    SQ_TICK(4)
    SQ_FLUSH()
}

// ---- third form: operands packed once per call, tiles staged by plain 16-byte copies -------------------------
// The second form spends ~1 000 of its ~2 000 non-MFMA instructions per wave and tile on staging: every tile turns the
// same rows of [xyz | features] into parity planes again (index arithmetic, a parity select and a range check per
// element), 32 times per row band.  Here a pre-pass writes the operand ONCE in exactly the LDS image the k loop reads
//     pack[frame][stage][row tile][parity][128][kPL]            (20 KiB per (stage, row tile); zeros for padding)
// together with the row norms (same ascending fmaf chain), and the matrix kernel stages a tile with five float4 loads
// and five ds_write_b128 per thread and operand, the loads of stage s+1 in flight during the matrix work of stage s.
// Same MFMA instruction, k order, norm chains and final expression as the other forms: bit-identical output.
constexpr int kPackTile = 2 * kMT * kPL;       // floats per (stage, row tile)

// One 256-thread workgroup per row tile.  The tile's rows are first copied into LDS with lanes running over the
// channels of one row (coalesced; a thread-per-row read touches 64 cache lines per instruction and re-fetches each
// line ~70 times: 15 us measured), then every thread owns one row: per stage its 36 channels go through the norm chain
// in channel order and out as the two 80-byte plane rows.
__global__ __launch_bounds__(2 * kMT) void sqdist_pack_kernel(int n, int np, int S, int ldc, RowSrc A,
                                                             float *__restrict__ pack, float *__restrict__ norms) {
    extern __shared__ float s_rows[];             // [kMT][ldc], ldc odd: a thread's row reads are conflict-free
    const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const int T = np / kMT, c = A.c0 + A.c1;
    const int lo = tid & (kMT - 1), hi = tid >> 7;        // hi: which half of the rows (copy) / which stages (pack)
    for (int ch = lo; ch < c; ch += kMT) {
        // both pieces are read unconditionally (clamped addresses) and selected: no branch between the loads; 32 rows
        // are requested before the first LDS store (a rolled load -> store loop pays one round trip per row: 61 us)
        const int k0 = ch < A.c0 ? ch : A.c0 - 1;
        const int k1 = A.c1 > 0 ? min(max(ch - A.c0, 0), A.c1 - 1) : 0;
        const bool first = ch < A.c0;
        for (int r0 = hi * (kMT / 2); r0 < (hi + 1) * (kMT / 2); r0 += 32) {
            float x[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int row = t * kMT + r0 + u;
                const int rr = row < n ? row : n - 1;
                const long grow0 = (long)b * (A.rs0 ? A.rs0 : n) + rr, grow1 = (long)b * (A.rs1 ? A.rs1 : n) + rr;
                const float x0 = A.p0[grow0 * A.c0 + k0];
                const float x1 = A.c1 > 0 ? A.p1[grow1 * A.c1 + k1] : 0.0f;
                x[u] = row < n ? (first ? x0 : x1) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) s_rows[(r0 + u) * ldc + ch] = x[u];
        }
    }
    __syncthreads();
    const float *mine = s_rows + lo * ldc;
    if (hi == 0) {                                        // the row's norm: one ascending fmaf chain over all channels
        float nrm = 0.0f;
        for (int k = 0; k < c; ++k) nrm = __builtin_fmaf(mine[k], mine[k], nrm);
        norms[(size_t)b * np + t * kMT + lo] = nrm;
    }
    for (int st = hi; st < S; st += 2) {
        float v[kKS2];
#pragma unroll
        for (int i = 0; i < kKS2; ++i) {
            const int k = st * kKS2 + i;
            const float x = mine[k < c ? k : 0];
            v[i] = k < c ? x : 0.0f;
        }
        float *dst = pack + ((((size_t)b * S + st) * T + t) * 2) * (kMT * kPL) + (size_t)lo * kPL;
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            f32x4v *d4 = (f32x4v *)(dst + (size_t)par * kMT * kPL);
#pragma unroll
            for (int q = 0; q < kPL / 4; ++q) {
                f32x4v o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 4 * q + e;
                    o[e] = j < kKS2 / 2 ? v[2 * j + par] : 0.0f;
                }
                d4[q] = o;
            }
        }
    }
}

template <bool SYM, bool NT>
__global__ __launch_bounds__(256, 4) void sqdist_mfma3_kernel(int n, int m, int c, int npa, int npb,
                                                             const float *__restrict__ packA,
                                                             const float *__restrict__ packB,
                                                             const float *__restrict__ normA,
                                                             const float *__restrict__ normB, float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float s_pl[2][2][kMT][kPL];
    static_assert(4 * 32 * 68 + 2 * kMT <= 2 * 2 * kMT * kPL, "patches + norms must fit the operand planes");
    float *sA = &s_pl[0][0][0][0] + 4 * 32 * 68, *sB = sA + kMT;
    int b = blockIdx.z;
    int bi = blockIdx.y, bj = blockIdx.x;
    if (SYM) {                                    // linear id over the upper triangle (row-major)
        const int T = (n + kMT - 1) / kMT;
        int rem = blockIdx.x;
        if ((gridDim.x & 7) == 0 && (gridDim.z & 7) == 0) {
            // XCD-aware (block L is observed to run on XCD L % 8, sa_common.h): the tiles of frame f go to XCD f % 8, whose
            // L2 then holds ONE frame's packed operand (1.3 MB at the layer-2 shape) instead of fetching all eight
            const unsigned L = blockIdx.z * gridDim.x + blockIdx.x, G8 = 8u * gridDim.x;
            b = (int)(8u * (L / G8) + (L & 7u));
            // ... and every frame starts an eighth of the tile list further on: the frames are a power of two apart (64 MB at
            // the layer-2 shape), eight XCDs writing the SAME tile of eight frames at the same moment meet in the same
            // HBM channels (measured at the layer-2 shape: 156 us without this rotation, 150 with it, 149 with the plain
            // mapping -- whose eight L2s fetch 85 MB of packed operands per launch instead of 14)
            rem = (int)(((L % G8) >> 3) + (L & 7u) * (gridDim.x >> 3)) % (int)gridDim.x;
        }
        bi = 0;
        while (rem >= T - bi) { rem -= T - bi; ++bi; }
        bj = bi + rem;
    }
    const int i0 = bi * kMT, j0 = bj * kMT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int half = lane >> 5, col = lane & 31;
    const int S = (c + kKS2 - 1) / kKS2, Ta = npa / kMT, Tb = npb / kMT;

    SQ_T0();
#ifdef SA_SQ_TIMING
    const unsigned long long wall0__ = wall_clock64();         // 100 MHz: the workgroup's lifetime in real time
#endif
    f32x16 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.0f;
    const float nrm = tid < kMT ? normA[(size_t)b * npa + i0 + tid] : normB[(size_t)b * npb + j0 + tid - kMT];

    // (stage, tile) images of the two operands: kPackTile floats = 1280 float4 each, 5 per thread
    constexpr int kCp = kPackTile / 4 / 256;
    static_assert(kCp * 4 * 256 == kPackTile, "tile image must split evenly over the workgroup");
    const f32x4v *srcA = (const f32x4v *)(packA + (((size_t)b * S) * Ta + bi) * kPackTile) + tid;
    const f32x4v *srcB = (const f32x4v *)(packB + (((size_t)b * S) * Tb + bj) * kPackTile) + tid;
    f32x4v ra[kCp], rb[kCp];
#pragma unroll
    for (int i = 0; i < kCp; ++i) { ra[i] = srcA[i * 256]; rb[i] = srcB[i * 256]; }
    f32x4v *dA = (f32x4v *)&s_pl[0][0][0][0] + tid, *dB = (f32x4v *)&s_pl[1][0][0][0] + tid;

    for (int st = 0; st < S; ++st) {
        const int cnt = min(kKS2, c - st * kKS2);
        const int npairs = (cnt + 1) >> 1;
        if (st > 0) __syncthreads();                       // the previous stage is fully consumed
#pragma unroll
        for (int i = 0; i < kCp; ++i) { dA[i * 256] = ra[i]; dB[i * 256] = rb[i]; }
        __syncthreads();
        SQ_TICK(0)
        if (st + 1 < S) {                                  // next stage's images: in flight during the matrix work
            const f32x4v *nA = srcA + (size_t)(st + 1) * Ta * (kPackTile / 4), *nB = srcB + (size_t)(st + 1) * Tb * (kPackTile / 4);
#pragma unroll
            for (int i = 0; i < kCp; ++i) { ra[i] = nA[i * 256]; rb[i] = nB[i * 256]; }
        }
        const float *pa0 = s_pl[0][half][wr * 64 + col], *pa1 = s_pl[0][half][wr * 64 + 32 + col];
        const float *pb0 = s_pl[1][half][wc * 64 + col], *pb1 = s_pl[1][half][wc * 64 + 32 + col];
        for (int p4 = 0; p4 < npairs; p4 += 4) {
            const float4 a0 = *(const float4 *)(pa0 + p4), a1 = *(const float4 *)(pa1 + p4);
            const float4 b0 = *(const float4 *)(pb0 + p4), b1 = *(const float4 *)(pb1 + p4);
            const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
            const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                if (p4 + x < npairs) {
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                        for (int tj = 0; tj < 2; ++tj)
                            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ti][x], bv[tj][x], acc[ti][tj], 0, 0, 0);
                }
            }
        }
        SQ_TICK(1)
    }
    __syncthreads();                               // operand planes dead -> norms and transpose patches
    if (tid < kMT) sA[tid] = nrm; else sB[tid - kMT] = nrm;
    __syncthreads();
    // full tiles (wave-uniform): both images in 256-byte row pieces; edge tiles: the scalar path of sq_store_tile
    SQ_TICK(2)
    const bool full = i0 + kMT <= n && j0 + kMT <= m && (m & 3) == 0 && (!SYM || (n & 3) == 0);
    if (full) sq_store_full<SYM, NT>(&s_pl[0][0][0][0], sA, sB, acc, b, n, m, i0, j0, SYM && bi != bj, w, lane, out);
    else sq_store_tile<SYM, NT>(&s_pl[0][0][0][0], sA, sB, acc, b, n, m, i0, j0, SYM && bi != bj, w, lane, out);
#ifdef SA_SQ_TIMING
    SQ_TICK(3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the stores' drain, which the wave's end waits for anyway
    SQ_TICK(4)
    acc__[5] = wall_clock64() - wall0__;
    SQ_FLUSH()
#endif
}

#ifdef SA_SQ_TIMING
}  // namespace
extern "C" int sa_debug_sq_prof(unsigned long long *host8, int reset) {
    if (host8) {
        static unsigned long long all[256][8];
        if (hipMemcpyFromSymbol(all, HIP_SYMBOL(g_sq_prof), sizeof(g_sq_prof)) != hipSuccess) return SA_ERR_LAUNCH;
        for (int i = 0; i < 8; ++i) { host8[i] = 0; for (int sl = 0; sl < 256; ++sl) host8[i] += all[sl][i]; }
    }
    if (reset) {
        void *d = nullptr;
        if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_sq_prof)) != hipSuccess) return SA_ERR_LAUNCH;
        if (hipMemset(d, 0, sizeof(g_sq_prof)) != hipSuccess) return SA_ERR_LAUNCH;
    }
    return SA_OK;
}
namespace {
#endif

}  // namespace

// a = [a0 | a1] rows [b,n,c0+c1], bb = [b0 | b1] rows [b,m,c0+c1]; out [b,n,m].
extern "C" int sa_calc_square_dist_split(int b, int n, int m, int c0, int c1, const float *a0,
                                         const float *a1, const float *b0, const float *b1, float *out,
                                         hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || c0 <= 0 || c1 < 0 || !a0 || !b0 || !out) return SA_ERR_INVALID;
    if (c1 > 0 && (!a1 || !b1)) return SA_ERR_INVALID;
    RowSrc A{a0, c0, a1, c1}, Bm{b0, c0, b1, c1};
    static const bool use_valu = SA_KNOB("SA_SQDIST_VALU", 0) != 0;
    if (use_valu) {
        dim3 grid((m + kT - 1) / kT, (n + kT - 1) / kT, b);
        hipLaunchKernelGGL(sqdist_kernel, grid, dim3(256), 0, stream, n, m, A, Bm, out);
    } else {
        // second form: needs the feature piece readable as float4 and tile-relative offsets that fit 32 bits
        static const bool v1_only = SA_KNOB("SA_SQDIST_V1", 0) != 0;
        const bool v2 = !v1_only && (c1 % 4) == 0 && ((uintptr_t)a1 % 16) == 0 && ((uintptr_t)b1 % 16) == 0 &&
                        (long)kMT * (m > n ? m : n) + kMT < (1l << 31);
        const bool nt = (size_t)b * n * m * sizeof(float) > ((size_t)192 << 20);
        if (n == m && a0 == b0 && a1 == b1) {      // the F-FPS case: symmetric, upper triangle only
            const int T = (n + kMT - 1) / kMT;
            dim3 grid(T * (T + 1) / 2, 1, b);
            if (v2 && nt) hipLaunchKernelGGL((sqdist_mfma2_kernel<true, true>), grid, dim3(256), 0, stream, n, m, A, Bm, out);
            else if (v2) hipLaunchKernelGGL((sqdist_mfma2_kernel<true, false>), grid, dim3(256), 0, stream, n, m, A, Bm, out);
            else hipLaunchKernelGGL(sqdist_mfma_kernel<true>, grid, dim3(256), 0, stream, n, m, A, Bm, out);
        } else {
            dim3 grid((m + kMT - 1) / kMT, (n + kMT - 1) / kMT, b);
            if (v2 && nt) hipLaunchKernelGGL((sqdist_mfma2_kernel<false, true>), grid, dim3(256), 0, stream, n, m, A, Bm, out);
            else if (v2) hipLaunchKernelGGL((sqdist_mfma2_kernel<false, false>), grid, dim3(256), 0, stream, n, m, A, Bm, out);
            else hipLaunchKernelGGL(sqdist_mfma_kernel<false>, grid, dim3(256), 0, stream, n, m, A, Bm, out);
        }
    }
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// Workspace of the packed form: operand images (+ norms) of a, and of bb unless it is the same operand.
extern "C" size_t sa_calc_square_dist_ws_bytes(int b, int n, int m, int c, int symmetric) {
    if (b <= 0 || n <= 0 || m <= 0 || c <= 0) return 0;
    const size_t S = (size_t)(c + kKS2 - 1) / kKS2;
    const size_t npa = (size_t)(n + kMT - 1) / kMT * kMT, npb = (size_t)(m + kMT - 1) / kMT * kMT;
    size_t fl = (size_t)b * (S * (npa / kMT) * kPackTile + npa);
    if (!symmetric) fl += (size_t)b * (S * (npb / kMT) * kPackTile + npb);
    return fl * sizeof(float);
}

// sa_calc_square_dist_split with caller-owned scratch of sa_calc_square_dist_ws_bytes(b, n, m, c0 + c1, a == bb)
// bytes: the packed form (one pre-pass per operand, then plain-copy staging).  Same result, bit for bit.
// The symmetric F-FPS case on a range slice: a0 / a1 are frames rs0 / rs1 ROWS apart (0 = n: dense).  Only the
// packed form reads strided sources; when it cannot run, SA_ERR_UNSUPPORTED (the caller copies the slice).
extern "C" int sa_calc_square_dist_self_ws(int b, int n, int c0, int c1, const float *a0, int rs0, const float *a1,
                                           int rs1, float *out, void *workspace, hipStream_t stream);

static int sqdist_split_ws_impl(int b, int n, int m, int c0, int c1, const float *a0, const float *a1,
                                const float *b0, const float *b1, float *out, void *workspace, int rs0, int rs1,
                                hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || c0 <= 0 || c1 < 0 || !a0 || !b0 || !out) return SA_ERR_INVALID;
    if (c1 > 0 && (!a1 || !b1)) return SA_ERR_INVALID;
    const bool strided = (rs0 != 0 && rs0 != n) || (rs1 != 0 && rs1 != n);
    static const bool packed_on = SA_KNOB("SA_SQDIST_PACKED", 1) != 0;
    const long big = (long)kMT * (m > n ? m : n) + kMT;
    if (!workspace || !packed_on || big >= (1l << 31) || ((uintptr_t)workspace % 16) != 0 || b > 65535) {
        if (strided) return SA_ERR_UNSUPPORTED;
        return sa_calc_square_dist_split(b, n, m, c0, c1, a0, a1, b0, b1, out, stream);
    }
    const bool sym = n == m && a0 == b0 && a1 == b1;
    const int c = c0 + c1, S = (c + kKS2 - 1) / kKS2;
    const int npa = (n + kMT - 1) / kMT * kMT, npb = (m + kMT - 1) / kMT * kMT;
    float *packA = (float *)workspace, *normA = packA + (size_t)b * S * (npa / kMT) * kPackTile;
    float *packB = packA, *normB = normA;
    RowSrc A{a0, c0, a1, c1}, Bm{b0, c0, b1, c1};
    A.rs0 = rs0; A.rs1 = rs1;
    if (sym) { Bm.rs0 = rs0; Bm.rs1 = rs1; }
    const int ldc = c | 1;
    const size_t pack_lds = (size_t)kMT * ldc * sizeof(float);
    if (pack_lds > 150 * 1024) {
        if (strided) return SA_ERR_UNSUPPORTED;
        return sa_calc_square_dist_split(b, n, m, c0, c1, a0, a1, b0, b1, out, stream);
    }
    if (pack_lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)sqdist_pack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pack_lds);
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(sqdist_pack_kernel, dim3(npa / kMT, b), dim3(2 * kMT), pack_lds, stream, n, npa, S, ldc, A, packA, normA);
    SA_CHECK_LAUNCH();
    if (!sym) {
        packB = normA + (size_t)b * npa;
        normB = packB + (size_t)b * S * (npb / kMT) * kPackTile;
        hipLaunchKernelGGL(sqdist_pack_kernel, dim3(npb / kMT, b), dim3(2 * kMT), pack_lds, stream, m, npb, S, ldc, Bm, packB, normB);
        SA_CHECK_LAUNCH();
    }
    const bool nt = (size_t)b * n * m * sizeof(float) > ((size_t)192 << 20);
    if (sym) {
        const int T = npa / kMT;
        dim3 grid(T * (T + 1) / 2, 1, b);
        if (nt) hipLaunchKernelGGL((sqdist_mfma3_kernel<true, true>), grid, dim3(256), 0, stream, n, m, c, npa, npb, packA, packB, normA, normB, out);
        else hipLaunchKernelGGL((sqdist_mfma3_kernel<true, false>), grid, dim3(256), 0, stream, n, m, c, npa, npb, packA, packB, normA, normB, out);
    } else {
        dim3 grid(npb / kMT, npa / kMT, b);
        if (nt) hipLaunchKernelGGL((sqdist_mfma3_kernel<false, true>), grid, dim3(256), 0, stream, n, m, c, npa, npb, packA, packB, normA, normB, out);
        else hipLaunchKernelGGL((sqdist_mfma3_kernel<false, false>), grid, dim3(256), 0, stream, n, m, c, npa, npb, packA, packB, normA, normB, out);
    }
    SA_CHECK_LAUNCH();
    return SA_OK;
}

extern "C" int sa_calc_square_dist_split_ws(int b, int n, int m, int c0, int c1, const float *a0, const float *a1,
                                            const float *b0, const float *b1, float *out, void *workspace,
                                            hipStream_t stream) {
    return sqdist_split_ws_impl(b, n, m, c0, c1, a0, a1, b0, b1, out, workspace, 0, 0, stream);
}
extern "C" int sa_calc_square_dist_self_ws(int b, int n, int c0, int c1, const float *a0, int rs0, const float *a1,
                                           int rs1, float *out, void *workspace, hipStream_t stream) {
    return sqdist_split_ws_impl(b, n, n, c0, c1, a0, a1, a0, a1, out, workspace, rs0, rs1, stream);
}

// model_util.calc_square_dist(a, b, norm=False): a [bs,n,c], bb [bs,m,c] -> [bs,n,m].
extern "C" int sa_calc_square_dist(int b, int n, int m, int c, const float *a, const float *bb,
                                   float *out, hipStream_t stream) {
    return sa_calc_square_dist_split(b, n, m, c, 0, a, nullptr, bb, nullptr, out, stream);
}
