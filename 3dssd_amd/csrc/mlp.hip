// Fused grouped MLP for gfx950:  gather -> concat[features, rel-xyz] -> up to 3 x (1x1 conv + folded BN
// + ReLU) -> max over nsample -> empty-ball mask, one launch per (SA layer, radius scale).
//
// Reference op sequence (lib/utils/layers_util.py:157-181): group_point x2, subtract, concat,
// tf_util.conv2d x3 (lib/utils/tf_util.py:127-201, BN :424-444), reduce_max, mask -- each a separate
// TF op with a materialised [B,m,ns,C] intermediate.  Here the grouped tensor never exists in HBM:
// a workgroup gathers a 32-row tile straight into LDS, runs the whole stack on the matrix cores
// with the activations staying in LDS, and writes only the pooled [B,m,Cout] result.
//
// Arithmetic: "split bf16".  Every fp32 operand x is carried as hi = bf16(x), lo = bf16(x - hi);
// each product is the three MFMA passes hi*hi + lo*hi + hi*lo (v_mfma_f32_32x32x16_bf16, fp32
// accumulate).  Plain bf16 misses the 1e-3 parity bar of BASELINE.json by 3-5x through the three
// stacked layers (measured against the fp32 oracle); the split form is ~5e-6.
//
// MFMA fragment layout used (guides: cdna_hip_programming.md section 3):
//   A operand: lane l holds A[i = l&31][k = 8*(l>>5) + e], e = 0..7
//   B operand: lane l holds B[k = 8*(l>>5) + e][j = l&31]
//   C/D:       lane l, reg r holds D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31]
// Hidden layers compute D^T = W^T * X^T (weights as A, activations as B) so that a lane ends up with
// 4 consecutive output channels of ONE row -> packed 8-byte LDS stores in the row-major layout the
// next layer reads.  The last layer swaps the operands (D = X * W): a lane then holds 16 rows of one
// output channel and the max over nsample is an in-register max plus one half-wave swap.
//
// LDS activation layout: act[row][g][plane][8] bf16, g = channel/8, plane 0 = hi / 1 = lo; row stride
// = 4*width + 16 bytes (an odd number of 16-byte slots -> conflict-free ds_read_b128 fragment reads).
// Weights are pre-packed on the host in fragment order: uint4 index ((ct*KS + ks)*2 + plane)*64 + lane
// holds W[k = 16*ks + 8*(lane>>5) + e][cout = 32*ct + (lane&31)], so every weight load is one
// contiguous 1 KiB wave access (they stream from L2; they are never staged through LDS).
#include <stdlib.h>

#include "sa_common.h"
#include "mlp_plan.h"
#include "mlp_act.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kNW = 8;             // waves per workgroup
constexpr int kThreads = kNW * 64;
constexpr int kRows = 32;          // rows (= MFMA N) per pass
constexpr int kMaxLayers = 3;

struct LayerDesc {
    const uint4 *w;     // fragment-packed hi/lo weights
    const float *bias;  // [NT*32], zero padded
    int K, N;           // true input / output channels
    int KS, NT;         // k-steps of 16, output tiles of 32
};

struct MlpParams {
    const float *xyz, *feat, *new_xyz;
    const int *idx, *cnt;
    float *out;
    int n, m, ns, C;
    long nballs;
    int out_stride, out_off;
    const int *gran;    // row plan (mlp_plan.h): granule entries, 4 per 32-row tile
    const int *hdr;     // hdr[0] = number of granules
    int nl;
    LayerDesc L[kMaxLayers];
    int strideA, strideB, lds_bytes;  // bytes
    int *ovf;           // fp16 form: raised when a converted activation left the fp16 range (mlp_act.h); may be null
};

#ifdef SA_MLP_TIMING
// phase clocks (s_memtime) summed over workgroups: 0 gather, 1 hidden layers, 2 last layer + pooling,
// 3 write-out, 4 items, 5 passes.  Debug builds only (tools/mlp_phase_prof.py).
__device__ unsigned long long g_mlp_prof[65536 * 8];   // one row per workgroup, summed on the host
#define SA_T0() unsigned long long t__ = __builtin_readcyclecounter(); unsigned long long acc__[4] = {0, 0, 0, 0}; unsigned long long cnt__[2] = {0, 0}
#define SA_TICK(i) { unsigned long long n__ = __builtin_readcyclecounter(); acc__[i] += n__ - t__; t__ = n__; }
#define SA_COUNT(i) cnt__[i]++
#define SA_TFLUSH(cond) if (cond) { unsigned long long *r__ = g_mlp_prof + (size_t)(blockIdx.x & 65535) * 8; for (int i__ = 0; i__ < 4; ++i__) r__[i__] += acc__[i__]; r__[4] += cnt__[0]; r__[5] += cnt__[1]; }
#else
#define SA_T0()
#define SA_TICK(i)
#define SA_COUNT(i)
#define SA_TFLUSH(cond)
#endif

__device__ __forceinline__ f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_f16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// Operand precision of a scale, PR (chosen on the host per scale, utils/weights.py):
//   3  split bf16: hi/lo planes of activations and weights, three MFMA passes per k-step (~5e-6 of the fp32 oracle)
//   1  fp16: one plane, one pass (v_cvt_pk_f16_f32 rounds to nearest even, v_mfma_f32_32x32x16_f16, fp32 accumulate);
//      used for the scales whose every contraction is >= 128 wide (layer3 / layer4 of 3dssd.yaml), 4-7e-4 of the fp32
//      oracle through the three stacked layers against the 1e-3 bar (utils/weights.py has the measurements).  Half
//      the LDS, weight bytes and fragment registers, a third of the matrix passes, one converter instruction per pair
//      instead of six.
constexpr __host__ __device__ int grp_bytes(int PR) { return PR == 3 ? 32 : 16; }   // one 8-channel group of an LDS row
constexpr __host__ __device__ int wblk(int PR) { return PR == 3 ? 128 : 64; }       // uint4 per (tile, k-step) of weights

// fp16 conversions go through mlp_act.h: every converted fragment is range-checked into the scalar mask `det`
__device__ __forceinline__ uint4 cvt8_f16(const float (&v)[8], sa::f16_guard_t &det) {
    const uint4 r = make_uint4(sa::cvt2_f16(v[0], v[1]), sa::cvt2_f16(v[2], v[3]), sa::cvt2_f16(v[4], v[5]), sa::cvt2_f16(v[6], v[7]));
    sa::f16_guard_signed(r, det);
    return r;
}
// acc += W x for one k-step: a_* = the MFMA A operand planes, b_* = the B operand planes
template <int PR, bool WFIRST>
__device__ __forceinline__ void mma_step(f32x16 &acc, uint4 wh, uint4 wl, uint4 xh, uint4 xl) {
    if (PR == 3) {
        if (WFIRST) { acc = mfma_bf16(wh, xh, acc); acc = mfma_bf16(wl, xh, acc); acc = mfma_bf16(wh, xl, acc); }
        else { acc = mfma_bf16(xh, wh, acc); acc = mfma_bf16(xh, wl, acc); acc = mfma_bf16(xl, wh, acc); }
    } else {
        acc = WFIRST ? mfma_f16(wh, xh, acc) : mfma_f16(xh, wh, acc);
    }
}

// 8 fp32 -> hi/lo bf16 planes, 16 bytes each
__device__ __forceinline__ void split8(const float (&v)[8], uint4 &hi, uint4 &lo) {
    unsigned h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sa::bf16_split(v[e], h[e], l[e]);
    hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

// 8 consecutive channels of a row -> their LDS group (hi/lo planes, or one fp16 plane)
template <int PR>
__device__ __forceinline__ void store_group(unsigned char *dst, const float (&v)[8], sa::f16_guard_t &det) {
    if (PR == 3) {
        uint4 hi, lo;
        split8(v, hi, lo);
        *(uint4 *)dst = hi;
        *(uint4 *)(dst + 16) = lo;
    } else {
        *(uint4 *)dst = cvt8_f16(v, det);
    }
}
// 4 consecutive channels (one accumulator quad, ReLU applied HERE) -> half a group
template <int PR>
__device__ __forceinline__ void store_quad_relu(unsigned char *dst, const float (&a)[4], sa::f16_guard_t &det) {
    if (PR == 3) {
        unsigned h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) sa::bf16_split(sa::fmax_nn(a[e], 0.0f), h[e], l[e]);
        *(uint2 *)dst = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
        *(uint2 *)(dst + 16) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
    } else {
        const uint2 r = make_uint2(sa::cvt2_f16_relu(a[0], a[1]), sa::cvt2_f16_relu(a[2], a[3]));
        sa::f16_guard(r, det);
        *(uint2 *)dst = r;
    }
}
// the same without an activation (values already final): split-bf16 only (vote tail)
__device__ __forceinline__ void store_quad_bf16(unsigned char *dst, const float (&v)[4]) {
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) sa::bf16_split(v[e], h[e], l[e]);
    *(uint2 *)dst = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    *(uint2 *)(dst + 16) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
}

// acc[tt] += sum_ks (hi*hi + lo*hi + hi*lo) for the TG output tiles starting at gb.  WFIRST: weights are the
// MFMA A operand (D^T form) else the activations are (D form).  Weight fragments (global, L2-resident) and
// activation fragments (LDS) of k-step ks+1 are requested before the MFMAs of k-step ks are issued, so the
// matrix pipe works under the load latency instead of behind it.
#define SA_MLP_PREFETCH 1
template <int TG, bool WFIRST, int PR>
__device__ __forceinline__ void mma_k_loop(f32x16 (&acc)[TG], const unsigned char *arow, const LayerDesc &L,
                                           int gb, int lane) {
    constexpr int WB = wblk(PR), KSB = 2 * grp_bytes(PR);     // uint4 per weight block, LDS bytes per k-step of a row
    const uint4 *wbase[TG];
#pragma unroll
    for (int tt = 0; tt < TG; ++tt)
        wbase[tt] = L.w + ((size_t)(min(gb + tt, L.NT - 1) * L.KS)) * WB + lane;
    // Batched k-steps: the weight (L2) and activation (LDS) fragments of KB consecutive k-steps are requested
    // together, then their MFMAs run back to back: one L2 round trip (~600-700 cycles measured) per KB k-steps
    // instead of one per k-step (a k-step is only 96 x TG cycles of MFMA issue).  The sched_barrier keeps hipcc
    // from sinking the later loads back down to their uses, which silently re-serialises the loop.
#ifndef SA_MLP_KB1
#define SA_MLP_KB1 4
#endif
    constexpr int KB = (TG >= 4 ? 1 : (TG == 2 ? (SA_MLP_KB1 >= 2 ? 2 : 1) : SA_MLP_KB1)) * (PR == 1 ? 2 : 1);
    for (int ks0 = 0; ks0 < L.KS; ks0 += KB) {
        uint4 wh[KB][TG], wl[KB][TG], ah[KB], al[KB];
#pragma unroll
        for (int d = 0; d < KB; ++d) {
            const int ks = ks0 + d < L.KS ? ks0 + d : L.KS - 1;
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) {
                wh[d][tt] = wbase[tt][ks * WB];
                if (PR == 3) wl[d][tt] = wbase[tt][ks * WB + 64];
            }
            ah[d] = *(const uint4 *)(arow + ks * KSB);
            if (PR == 3) al[d] = *(const uint4 *)(arow + ks * KSB + 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < KB; ++d) {
            if (ks0 + d < L.KS) {
#pragma unroll
                for (int tt = 0; tt < TG; ++tt)
                    if (gb + tt < L.NT) mma_step<PR, WFIRST>(acc[tt], wh[d][tt], wl[d][tt], ah[d], al[d]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}


// 8 consecutive channels [c0, c0+8) of the grouped row (features first, then xyz - centre, then zero
// padding).  The mixed / unaligned case issues every load unconditionally at a clamped address and
// selects afterwards: one round trip, no per-element branch + wait.
__device__ __forceinline__ void gather8(const MlpParams &P, long pt, long ball, int c0, float (&v)[8]) {
    if ((P.C & 3) == 0 && c0 + 8 <= P.C) {
        const float4 f0 = *(const float4 *)(P.feat + pt * P.C + c0);
        const float4 f1 = *(const float4 *)(P.feat + pt * P.C + c0 + 4);
        v[0] = f0.x; v[1] = f0.y; v[2] = f0.z; v[3] = f0.w;
        v[4] = f1.x; v[5] = f1.y; v[6] = f1.z; v[7] = f1.w;
        return;
    }
    const float px = P.xyz[pt * 3 + 0] - P.new_xyz[ball * 3 + 0];
    const float py = P.xyz[pt * 3 + 1] - P.new_xyz[ball * 3 + 1];
    const float pz = P.xyz[pt * 3 + 2] - P.new_xyz[ball * 3 + 2];
    float fv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        fv[e] = P.C > 0 ? P.feat[pt * P.C + (c < P.C ? c : P.C - 1)] : 0.0f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int x = c0 + e - P.C;
        v[e] = x < 0 ? fv[e] : (x == 0 ? px : (x == 1 ? py : (x == 2 ? pz : 0.0f)));
    }
}


// Gather one 32/64-row input tile into `buf` (hi/lo bf16 planes), NTHR threads, no barrier inside.
//  * every wave resolves (ball, source point) of each tile row once (lane r -> row r) and hands it to the
//    slot that needs it with a wave shuffle -- idx/cnt are not re-read per 8-channel group;
//  * the full 8-channel feature groups (two aligned float4 loads, identical for every lane: no divergence)
//    are separated from the short tail (left-over feature channels + relative xyz + zero padding), which
//    only 32..128 slots execute.
template <int ROWS, int NTHR, int PR>
__device__ __forceinline__ void gather_tile(const MlpParams &P, unsigned char *buf, int stride, int item,
                                            int ngran, int G0, int tid, sa::f16_guard_t &det) {
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    // rows lane and lane+64 (ROWS == 64) / lane&31 (ROWS == 32) resolved by this lane
    int r_pt[ROWS / 32 > 1 ? 1 : 1], r_ball[1];
    {
        const int row = ROWS == 64 ? lane : (lane & 31);
        // granule row>>3 of the item -> plan entry -> (ball, sample); entries past the end read ball 0
        const int ent = sa::plan_entry(P.gran, ngran, item * (ROWS / 8) + (row >> 3));
        const long ball = ent >= 0 ? sa::plan_ball(ent) : 0;
        const int s = sa::plan_sample(ent, row & 7, P.ns);
        const int a_raw = P.idx[ball * P.ns + s];           // both loads issue together
        const int a = P.cnt[ball] > 0 ? a_raw : 0;          // layers_util.py:157-159
        r_ball[0] = (int)ball;
        r_pt[0] = (int)((ball / P.m) * P.n + a);
    }
    const int GF = (P.C & 3) == 0 ? (P.C >> 3) : 0;         // full feature groups per row
    const int totF = ROWS * GF;
    // uniform trip counts: every lane stays active for the shuffles, out-of-range slots repeat the last one
#pragma unroll 4
    for (int it0 = 0; it0 < totF; it0 += NTHR) {
        const bool live = it0 + tid < totF;
        const int it = live ? it0 + tid : totF - 1;
        const int row = it / GF, g = it - row * GF;
        const long pt = __shfl(r_pt[0], row);
        const float4 f0 = *(const float4 *)(P.feat + pt * P.C + g * 8);
        const float4 f1 = *(const float4 *)(P.feat + pt * P.C + g * 8 + 4);
        const float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
        if (live) store_group<PR>(buf + row * stride + g * grp_bytes(PR), v, det);
    }
    const int GT = G0 - GF;                                 // tail groups per row (1 or 2; all of them if C%4 != 0)
    const int totT = ROWS * GT;
    for (int it0 = 0; it0 < totT; it0 += NTHR) {
        const bool live = it0 + tid < totT;
        const int it = live ? it0 + tid : totT - 1;
        const int row = it / GT, g = GF + (it - row * GT);
        const long pt = __shfl(r_pt[0], row);
        const long ball = __shfl(r_ball[0], row);
        const float px = P.xyz[pt * 3 + 0] - P.new_xyz[ball * 3 + 0];
        const float py = P.xyz[pt * 3 + 1] - P.new_xyz[ball * 3 + 1];
        const float pz = P.xyz[pt * 3 + 2] - P.new_xyz[ball * 3 + 2];
        // feature channels that did not fill a whole group (nleft of them after the GF full groups; wave-
        // uniform): element e of a tail group can only be a feature if e < nleft, so the branch is uniform and
        // the loads are independent; the address is clamped, the value selected per lane
        const int nleft = P.C - GF * 8;
        float fv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            fv[e] = 0.0f;
            if (e < nleft) {
                const int c = g * 8 + e;
                fv[e] = P.feat[pt * P.C + (c < P.C ? c : P.C - 1)];
            }
        }
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int x = g * 8 + e - P.C;
            v[e] = x < 0 ? fv[e] : (x == 0 ? px : (x == 1 ? py : (x == 2 ? pz : 0.0f)));
        }
        if (live) store_group<PR>(buf + row * stride + g * grp_bytes(PR), v, det);
    }
}

// ---- hidden layer, D^T form: out[row][cout] = relu(bias + sum_k W[k][cout] * in[row][k]) ----------
template <int TG, int NW, int PR>
__device__ __forceinline__ void layer_hidden(const unsigned char *in, int strideIn, unsigned char *outb,
                                             int strideOut, const LayerDesc &L, int lane, int w, sa::f16_guard_t &det) {
    constexpr int GB = grp_bytes(PR);
    const int half = lane >> 5, col = lane & 31;
    const unsigned char *arow = in + col * strideIn + half * GB;
    for (int gb = w * TG; gb < L.NT; gb += NW * TG) {
        f32x16 acc[TG];
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) {
            const int ct = min(gb + tt, L.NT - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *(const float4 *)(L.bias + ct * 32 + 8 * q + 4 * half);
                acc[tt][4 * q + 0] = bv.x; acc[tt][4 * q + 1] = bv.y;
                acc[tt][4 * q + 2] = bv.z; acc[tt][4 * q + 3] = bv.w;
            }
        }
        mma_k_loop<TG, true, PR>(acc, arow, L, gb, lane);
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) {
            if (gb + tt < L.NT) {
                unsigned char *orow = outb + col * strideOut + (gb + tt) * 4 * GB + 8 * half;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v[4] = {acc[tt][4 * q], acc[tt][4 * q + 1], acc[tt][4 * q + 2], acc[tt][4 * q + 3]};
                    store_quad_relu<PR>(orow + q * GB, v, det);
                }
            }
        }
    }
}

// ---- last layer, D form + max over the rows of each ball; bias/ReLU/mask are applied at write-out ---
template <int TG, int NW, int PR>
__device__ __forceinline__ void layer_last(const unsigned char *in, int strideIn, const LayerDesc &L,
                                           const MlpParams &P, const int (&ent)[4], const int (&cn)[4], int lane,
                                           int w) {
    const int half = lane >> 5, col = lane & 31;
    const unsigned char *arow = in + col * strideIn + half * grp_bytes(PR);
    for (int gb = w * TG; gb < L.NT; gb += NW * TG) {
        f32x16 acc[TG];
#pragma unroll
        for (int tt = 0; tt < TG; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tt][r] = 0.0f;
        mma_k_loop<TG, false, PR>(acc, arow, L, gb, lane);
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) {
            if (gb + tt < L.NT) {
                // granule maxima (rows 8q..8q+7), combined per ball run, relu(max + bias) written (mlp_plan.h)
                float qm[4];
                sa::granule_max(acc[tt], qm);
                const int c = (gb + tt) * 32 + col;
                sa::pool_write_tile(qm, ent, cn, L.bias[c], c, L.N, P.out, P.out_stride, P.out_off, lane);
            }
        }
    }
}

#ifndef SA_MLP_MAXTG
#define SA_MLP_MAXTG 2
#endif
#ifndef SA_MLP_WPE8
#define SA_MLP_WPE8 4      // min waves per SIMD the 8-wave kernel is compiled for (4 -> 128 VGPRs, 2 -> 256)
#endif
template <int NW>
__device__ __forceinline__ int pick_tg(int NT) {
    // the one-wave kernel keeps TG <= 2 under register prefetch (4 tiles x double-buffered fragments spill)
    constexpr int cap = (NW == 1 && SA_MLP_PREFETCH) ? 2 : SA_MLP_MAXTG;
    return (cap >= 4 && NT >= 4 * NW) ? 4 : (NT >= 2 * NW ? 2 : 1);
}

// NW = waves cooperating on one 32-row item.  NW = 8: the wide layers (weights split across waves,
// one 512-thread workgroup per item).  NW = 1: narrow layers (layer1/layer2 of 3dssd.yaml) where one
// wave runs the whole stack on its own item -- no cross-wave barrier, 8x more items in flight per CU to
// hide the idx -> point gather latency.
template <int NW, int PR>
// (the fp16 form with its range guard is compiled for 2 waves per SIMD: at 128 registers it spilled 60 of them)
__global__ __launch_bounds__(NW * 64, PR == 1 ? 2 : (NW == 8 ? SA_MLP_WPE8 : 4)) void group_mlp_max_kernel(MlpParams P) {
    constexpr int kThr = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *bufA = smem;
    unsigned char *bufB = smem + kRows * P.strideA;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ngran = __builtin_amdgcn_readfirstlane(P.hdr[0]);
    const int nitems = (ngran + 3) >> 2;               // 32-row tiles of the plan
    const LayerDesc &LL = P.L[P.nl - 1];
    const int G0 = P.L[0].KS * 2;                     // 8-channel groups of the input tile
    SA_T0();

    int istride;
    const int item0 = sa::xcd_block(blockIdx.x, gridDim.x, nitems, istride);
    if (item0 < 0) return;
    sa::f16_guard_t det = 0;                           // fp16 range guard (mlp_act.h), scalar registers
    for (int item = item0; item < nitems; item += istride) {
        SA_COUNT(0);
        SA_COUNT(1);
        // ---- gather the [32 rows x cin] input tile (features first, then relative xyz:
        //      layers_util.py:160-165) into bufA as hi/lo bf16
        gather_tile<kRows, kThr, PR>(P, bufA, P.strideA, item, ngran, G0, tid, det);
        // the tile's plan entries and ball counts, wave-uniform
        int ent[4], cn[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int e = sa::plan_entry(P.gran, ngran, item * 4 + g);
            ent[g] = __builtin_amdgcn_readfirstlane(e);
            cn[g] = __builtin_amdgcn_readfirstlane(e >= 0 ? P.cnt[sa::plan_ball(e)] : 0);
        }
        __syncthreads();
        SA_TICK(0)
        // ---- hidden layers (ping-pong A -> B -> A)
        for (int l = 0; l + 1 < P.nl; ++l) {
            const unsigned char *in = (l & 1) ? bufB : bufA;
            unsigned char *ob = (l & 1) ? bufA : bufB;
            const int si = (l & 1) ? P.strideB : P.strideA, so = (l & 1) ? P.strideA : P.strideB;
            switch (pick_tg<NW>(P.L[l].NT)) {
#if SA_MLP_MAXTG >= 4
                case 4: layer_hidden<4, NW, PR>(in, si, ob, so, P.L[l], lane, w, det); break;
#endif
                case 2: layer_hidden<2, NW, PR>(in, si, ob, so, P.L[l], lane, w, det); break;
                default: layer_hidden<1, NW, PR>(in, si, ob, so, P.L[l], lane, w, det); break;
            }
            __syncthreads();
        }
        SA_TICK(1)
        // ---- last layer + pooling + write-out: relu(max + bias), zero for empty balls (layers_util.py:178-181)
        {
            const int l = P.nl - 1;
            const unsigned char *in = (l & 1) ? bufB : bufA;
            const int si = (l & 1) ? P.strideB : P.strideA;
            switch (pick_tg<NW>(LL.NT)) {
#if SA_MLP_MAXTG >= 4
                case 4: layer_last<4, NW, PR>(in, si, LL, P, ent, cn, lane, w); break;
#endif
                case 2: layer_last<2, NW, PR>(in, si, LL, P, ent, cn, lane, w); break;
                default: layer_last<1, NW, PR>(in, si, LL, P, ent, cn, lane, w); break;
            }
        }
        __syncthreads();
        SA_TICK(2)
    }
    if (PR == 1) sa::f16_overflow_report(det, P.ovf, lane);
    SA_TFLUSH(tid == 0)
}


// =====================================================================================================
// Wide-layer kernel: 64 rows per item.  Every weight fragment fetched from L2 feeds TWO 32-row tiles
// (6 MFMA passes per 2 KiB of weights instead of 3), which halves the L2 weight traffic that bounds the
// 32-row kernel on the 256..1024-channel layers.  The last layer's accumulators (all of a wave's output
// tiles x 2 row tiles) stay in registers while the last HIDDEN layer is produced in column chunks, so its
// full-width output never has to sit in LDS (512 channels x 64 rows x hi/lo would not fit next to the
// 256-channel buffer): chunk c of layer nl-2 is written, consumed as k-range c of layer nl-1, overwritten.
// =====================================================================================================
constexpr int kWRows = 64;

struct WideParams {
    MlpParams M;          // strideA/strideB/lds_bytes are for 64-row buffers here
    int tiles_per_chunk;  // output tiles of the last hidden layer per chunk (all of them when nl == 1)
    int nchunks;
};

// hidden layer (D^T form) for output tiles [tile_lo, tile_hi), written at column (tile - tile_lo) * 32
template <int TG, int PR>
__device__ __forceinline__ void wide_hidden(const unsigned char *in, int strideIn, unsigned char *outb,
                                            int strideOut, const LayerDesc &L, int tile_lo, int tile_hi,
                                            int lane, int w, sa::f16_guard_t &det) {
    asm volatile("" : "+v"(lane));
    constexpr int GB = grp_bytes(PR), WB = wblk(PR), KSB = 2 * GB;
    const int half = lane >> 5, col = lane & 31;
    const unsigned char *arow0 = in + col * strideIn + half * GB;
    const unsigned char *arow1 = arow0 + 32 * strideIn;
    for (int gb = tile_lo + w * TG; gb < tile_hi; gb += kNW * TG) {
        f32x16 acc[TG][2];
        const uint4 *wbase[TG];
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) {
            const int ct = min(gb + tt, tile_hi - 1);
            wbase[tt] = L.w + ((size_t)(ct * L.KS)) * WB + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *(const float4 *)(L.bias + ct * 32 + 8 * q + 4 * half);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[tt][j][4 * q + 0] = bv.x; acc[tt][j][4 * q + 1] = bv.y;
                    acc[tt][j][4 * q + 2] = bv.z; acc[tt][j][4 * q + 3] = bv.w;
                }
            }
        }
        constexpr int D = (TG == 2 ? 2 : 4) * (PR == 1 ? 2 : 1);     // weight fragments in flight (k-steps ahead)
        uint4 wq[D][TG][2];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int kd = d < L.KS ? d : L.KS - 1;
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) {
                wq[d][tt][0] = wbase[tt][kd * WB];
                if (PR == 3) wq[d][tt][1] = wbase[tt][kd * WB + 64];
            }
        }
        uint4 ah0 = *(const uint4 *)(arow0), al0, ah1 = *(const uint4 *)(arow1), al1;
        if (PR == 3) { al0 = *(const uint4 *)(arow0 + 16); al1 = *(const uint4 *)(arow1 + 16); }
        for (int ks0 = 0; ks0 < L.KS; ks0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int ks = ks0 + d;
                if (ks < L.KS) {
                    uint4 wh[TG], wl[TG];
#pragma unroll
                    for (int tt = 0; tt < TG; ++tt) { wh[tt] = wq[d][tt][0]; wl[tt] = wq[d][tt][1]; }
                    const int kw = ks + D < L.KS ? ks + D : L.KS - 1;
#pragma unroll
                    for (int tt = 0; tt < TG; ++tt) {
                        wq[d][tt][0] = wbase[tt][kw * WB];
                        if (PR == 3) wq[d][tt][1] = wbase[tt][kw * WB + 64];
                    }
                    const int kn = ks + 1 < L.KS ? ks + 1 : ks;
                    const uint4 nah0 = *(const uint4 *)(arow0 + kn * KSB), nah1 = *(const uint4 *)(arow1 + kn * KSB);
                    uint4 nal0, nal1;
                    if (PR == 3) { nal0 = *(const uint4 *)(arow0 + kn * KSB + 16); nal1 = *(const uint4 *)(arow1 + kn * KSB + 16); }
#pragma unroll
                    for (int tt = 0; tt < TG; ++tt) {
                        if (gb + tt < tile_hi) {
                            if (PR == 3) {
                                acc[tt][0] = mfma_bf16(wh[tt], ah0, acc[tt][0]);
                                acc[tt][1] = mfma_bf16(wh[tt], ah1, acc[tt][1]);
                                acc[tt][0] = mfma_bf16(wl[tt], ah0, acc[tt][0]);
                                acc[tt][1] = mfma_bf16(wl[tt], ah1, acc[tt][1]);
                                acc[tt][0] = mfma_bf16(wh[tt], al0, acc[tt][0]);
                                acc[tt][1] = mfma_bf16(wh[tt], al1, acc[tt][1]);
                            } else {
                                acc[tt][0] = mfma_f16(wh[tt], ah0, acc[tt][0]);
                                acc[tt][1] = mfma_f16(wh[tt], ah1, acc[tt][1]);
                            }
                        }
                    }
                    ah0 = nah0; ah1 = nah1;
                    if (PR == 3) { al0 = nal0; al1 = nal1; }
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) {
            if (gb + tt < tile_hi) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    unsigned char *orow = outb + (j * 32 + col) * strideOut + (gb + tt - tile_lo) * 4 * GB + 8 * half;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v[4] = {acc[tt][j][4 * q], acc[tt][j][4 * q + 1], acc[tt][j][4 * q + 2], acc[tt][j][4 * q + 3]};
                        store_quad_relu<PR>(orow + q * GB, v, det);
                    }
                }
            }
        }
    }
}

// last layer (D form), k-steps [ks_lo, ks_hi) of the input whose columns start at k-step ks_lo in `in`;
// wave w owns output tiles w, w+8, ... (TGL of them), accumulators persist across calls
template <int TGL, int PR>
__device__ __forceinline__ void wide_last_partial(f32x16 (&acc)[TGL][2], const unsigned char *in, int strideIn,
                                                  const LayerDesc &L, int ks_lo, int ks_hi, int lane, int w) {
    asm volatile("" : "+v"(lane));
    constexpr int GB = grp_bytes(PR), WB = wblk(PR), KSB = 2 * GB;
    const int half = lane >> 5, col = lane & 31;
    const unsigned char *arow0 = in + col * strideIn + half * GB;
    const unsigned char *arow1 = arow0 + 32 * strideIn;
    const uint4 *wbase[TGL];
#pragma unroll
    for (int tt = 0; tt < TGL; ++tt)
        wbase[tt] = L.w + ((size_t)(min(w + kNW * tt, L.NT - 1) * L.KS)) * WB + lane;
    constexpr int D = (TGL >= 4 ? 1 : (TGL == 2 ? 2 : 4)) * (PR == 1 ? 2 : 1);
    uint4 wq[D][TGL][2];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int kd = ks_lo + d < ks_hi ? ks_lo + d : ks_hi - 1;
#pragma unroll
        for (int tt = 0; tt < TGL; ++tt) {
            wq[d][tt][0] = wbase[tt][kd * WB];
            if (PR == 3) wq[d][tt][1] = wbase[tt][kd * WB + 64];
        }
    }
    uint4 ah0 = *(const uint4 *)(arow0), al0, ah1 = *(const uint4 *)(arow1), al1;
    if (PR == 3) { al0 = *(const uint4 *)(arow0 + 16); al1 = *(const uint4 *)(arow1 + 16); }
    for (int ks0 = ks_lo; ks0 < ks_hi; ks0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int ks = ks0 + d;
            if (ks < ks_hi) {
                uint4 wh[TGL], wl[TGL];
#pragma unroll
                for (int tt = 0; tt < TGL; ++tt) { wh[tt] = wq[d][tt][0]; wl[tt] = wq[d][tt][1]; }
                const int kw = ks + D < ks_hi ? ks + D : ks_hi - 1;
#pragma unroll
                for (int tt = 0; tt < TGL; ++tt) {
                    wq[d][tt][0] = wbase[tt][kw * WB];
                    if (PR == 3) wq[d][tt][1] = wbase[tt][kw * WB + 64];
                }
                const int kn = (ks + 1 < ks_hi ? ks + 1 : ks) - ks_lo;
                const uint4 nah0 = *(const uint4 *)(arow0 + kn * KSB), nah1 = *(const uint4 *)(arow1 + kn * KSB);
                uint4 nal0, nal1;
                if (PR == 3) { nal0 = *(const uint4 *)(arow0 + kn * KSB + 16); nal1 = *(const uint4 *)(arow1 + kn * KSB + 16); }
#pragma unroll
                for (int tt = 0; tt < TGL; ++tt) {
                    if (w + kNW * tt < L.NT) {
                        if (PR == 3) {
                            acc[tt][0] = mfma_bf16(ah0, wh[tt], acc[tt][0]);
                            acc[tt][1] = mfma_bf16(ah1, wh[tt], acc[tt][1]);
                            acc[tt][0] = mfma_bf16(ah0, wl[tt], acc[tt][0]);
                            acc[tt][1] = mfma_bf16(ah1, wl[tt], acc[tt][1]);
                            acc[tt][0] = mfma_bf16(al0, wh[tt], acc[tt][0]);
                            acc[tt][1] = mfma_bf16(al1, wh[tt], acc[tt][1]);
                        } else {
                            acc[tt][0] = mfma_f16(ah0, wh[tt], acc[tt][0]);
                            acc[tt][1] = mfma_f16(ah1, wh[tt], acc[tt][1]);
                        }
                    }
                }
                ah0 = nah0; ah1 = nah1;
                if (PR == 3) { al0 = nal0; al1 = nal1; }
            }
        }
    }
}

template <int TGL, int PR>
__device__ __forceinline__ void wide_item_layers(const WideParams &WP, unsigned char *bufA, unsigned char *bufB,
                                                 const int (&ent)[2][4], const int (&cn)[2][4], int lane, int w,
                                                 sa::f16_guard_t &det) {
    const MlpParams &P = WP.M;
    const int nl = P.nl;
    const LayerDesc &LL = P.L[nl - 1];
    // hidden layers that are stored whole
    for (int l = 0; l + 2 < nl; ++l) {
        const unsigned char *in = (l & 1) ? bufB : bufA;
        unsigned char *ob = (l & 1) ? bufA : bufB;
        const int si = (l & 1) ? P.strideB : P.strideA, so = (l & 1) ? P.strideA : P.strideB;
        if (P.L[l].NT >= 2 * kNW) wide_hidden<2, PR>(in, si, ob, so, P.L[l], 0, P.L[l].NT, lane, w, det);
        else wide_hidden<1, PR>(in, si, ob, so, P.L[l], 0, P.L[l].NT, lane, w, det);
        __syncthreads();
    }
    f32x16 acc[TGL][2];
#pragma unroll
    for (int tt = 0; tt < TGL; ++tt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tt][j][r] = 0.0f;
    if (nl == 1) {
        wide_last_partial<TGL, PR>(acc, bufA, P.strideA, LL, 0, LL.KS, lane, w);
    } else {
        const int l = nl - 2;                         // last hidden layer, produced in column chunks
        const unsigned char *in = (l & 1) ? bufB : bufA;
        unsigned char *ob = (l & 1) ? bufA : bufB;
        const int si = (l & 1) ? P.strideB : P.strideA, so = (l & 1) ? P.strideA : P.strideB;
        for (int c = 0; c < WP.nchunks; ++c) {
            const int t_lo = c * WP.tiles_per_chunk;
            const int t_hi = min(P.L[l].NT, t_lo + WP.tiles_per_chunk);
            if (c > 0) __syncthreads();               // previous chunk fully consumed before it is overwritten
            // with 128 accumulator registers live (TGL == 4) only the one-tile form fits the register file
            if (TGL < 4 && t_hi - t_lo >= 2 * kNW) wide_hidden<2, PR>(in, si, ob, so, P.L[l], t_lo, t_hi, lane, w, det);
            else wide_hidden<1, PR>(in, si, ob, so, P.L[l], t_lo, t_hi, lane, w, det);
            __syncthreads();
            const int ks_lo = t_lo * 2, ks_hi = min(LL.KS, t_hi * 2);
            if (ks_lo < ks_hi) wide_last_partial<TGL, PR>(acc, ob, so, LL, ks_lo, ks_hi, lane, w);
        }
    }
    // pooling + write-out: row tile j covers rows 32j..32j+31 of the item = plan tile 2*item + j
    asm volatile("" : "+v"(lane));
    const int col = lane & 31;
#pragma unroll
    for (int tt = 0; tt < TGL; ++tt) {
        const int ct = w + kNW * tt;
        if (ct < LL.NT) {
            const int c = ct * 32 + col;
            const float bc = LL.bias[c];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float qm[4];
                sa::granule_max(acc[tt][j], qm);
                sa::pool_write_tile(qm, ent[j], cn[j], bc, c, LL.N, P.out, P.out_stride, P.out_off, lane);
            }
        }
    }
}

template <int PR>
__global__ __launch_bounds__(kThreads, 2) void group_mlp_wide_kernel(WideParams WP) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const MlpParams &P = WP.M;
    unsigned char *bufA = smem;
    unsigned char *bufB = smem + kWRows * P.strideA;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ngran = __builtin_amdgcn_readfirstlane(P.hdr[0]);
    const int nitems = (ngran + 7) >> 3;               // 64-row items = 8 granules of the plan
    const LayerDesc &LL = P.L[P.nl - 1];
    const int G0 = P.L[0].KS * 2;
    const int tgl = (LL.NT + kNW - 1) / kNW;

    // XCD x takes a contiguous eighth of every pass over the items (sa::xcd_block): a frame's feature rows go to one
    // or two L2s, not eight
    int istride;
    const int item0 = sa::xcd_block(blockIdx.x, gridDim.x, nitems, istride);
    if (item0 < 0) return;
    sa::f16_guard_t det = 0;                           // fp16 range guard (mlp_act.h), scalar registers
    for (int item = item0; item < nitems; item += istride) {
        gather_tile<kWRows, kThreads, PR>(P, bufA, P.strideA, item, ngran, G0, tid, det);
        int ent[2][4], cn[2][4];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int e = sa::plan_entry(P.gran, ngran, item * 8 + g);
            ent[g >> 2][g & 3] = __builtin_amdgcn_readfirstlane(e);
            cn[g >> 2][g & 3] = __builtin_amdgcn_readfirstlane(e >= 0 ? P.cnt[sa::plan_ball(e)] : 0);
        }
        __syncthreads();
        if (tgl <= 1) wide_item_layers<1, PR>(WP, bufA, bufB, ent, cn, lane, w, det);
        else if (tgl <= 2) wide_item_layers<2, PR>(WP, bufA, bufB, ent, cn, lane, w, det);
        else wide_item_layers<4, PR>(WP, bufA, bufB, ent, cn, lane, w, det);
        __syncthreads();
    }
    if (PR == 1) sa::f16_overflow_report(det, P.ovf, lane);
}

// ---- dense layer on rows: y = act(x W + b), x [rows,K] fp32 -> y [rows,N] fp32 (conv1d 1x1:
//      the "ensemble" aggregation of layers_util.py:183-185 and vote_layer, layers_util.py:17-19) ----
struct DenseParams {
    const float *x;
    float *y;
    long rows;
    int relu;
    LayerDesc L;
    int stride;   // LDS row stride in bytes for a KC-wide chunk
    int KC;       // channels staged per chunk (multiple of 16)
    int tg;       // output tiles per wave (1, 2 or 4)
};

constexpr int kDW = 4;               // waves per dense workgroup
constexpr int kDThreads = kDW * 64;

constexpr int kDStageIt = (kRows * 32 + kDThreads - 1) / kDThreads;   // KC <= 256 -> at most 32 eight-float groups per row

// requests the fp32 values of chunk k0 (32 rows x KC channels, 8 per item) into registers; no wait here
__device__ __forceinline__ void dense_stage_load(const DenseParams &P, float (&sv)[kDStageIt][8], long r0, int k0, int Kp,
                                                 int tid) {
    const LayerDesc &L = P.L;
    const int kc = min(P.KC, Kp - k0);
    const int G = kc / 8;
#pragma unroll
    for (int i = 0; i < kDStageIt; ++i) {
        const int it = tid + i * kDThreads;
        const int itc = it < kRows * G ? it : 0;
        const int row = itc / G, g = itc - row * G;
        long r = r0 + row;
        if (r >= P.rows) r = P.rows - 1;
        const int c0 = k0 + g * 8;
        if ((L.K & 3) == 0 && c0 + 8 <= L.K) {
            const float4 f0 = *(const float4 *)(P.x + r * L.K + c0);
            const float4 f1 = *(const float4 *)(P.x + r * L.K + c0 + 4);
            sv[i][0] = f0.x; sv[i][1] = f0.y; sv[i][2] = f0.z; sv[i][3] = f0.w;
            sv[i][4] = f1.x; sv[i][5] = f1.y; sv[i][6] = f1.z; sv[i][7] = f1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = P.x[r * L.K + (c0 + e < L.K ? c0 + e : L.K - 1)];
                sv[i][e] = c0 + e < L.K ? x : 0.0f;
            }
        }
    }
}

// One workgroup: 32 rows x (kDW*TG output tiles starting at tile blockIdx.y*kDW*TG).  Splitting the output
// channels over blockIdx.y keeps >= 256 workgroups in flight for the short, wide aggregation layers
// (2048 rows x 1536 -> 512 is only 64 row tiles).
// dense_tile_accumulate: acc[tt] = bias + x[r0 .. r0+31, :] W for the wave's TG output tiles (D^T form: a lane holds row
// r0 + (lane & 31), channels 8q + 4 * (lane >> 5) + e of the tile).  Ends with a barrier: `buf` is free on return.
template <int TG>
__device__ __forceinline__ void dense_tile_accumulate(const DenseParams &P, unsigned char *buf, long r0, int gb, int lane,
                                                      int tid, f32x16 (&acc)[TG]) {
    const int half = lane >> 5, col = lane & 31;
    const LayerDesc &L = P.L;
    const int Kp = L.KS * 16;
#pragma unroll
    for (int tt = 0; tt < TG; ++tt) {
        const int ct = min(gb + tt, L.NT - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bv = *(const float4 *)(L.bias + ct * 32 + 8 * q + 4 * half);
            acc[tt][4 * q + 0] = bv.x; acc[tt][4 * q + 1] = bv.y;
            acc[tt][4 * q + 2] = bv.z; acc[tt][4 * q + 3] = bv.w;
        }
    }
    // The fp32 rows of chunk k0 + KC are requested (into registers) before the matrix work of chunk k0 starts, so
    // their HBM/L2 latency overlaps it; one LDS buffer is enough (it is rewritten after the closing barrier).
    float sv[kDStageIt][8];
    dense_stage_load(P, sv, r0, 0, Kp, tid);
    for (int k0 = 0; k0 < Kp; k0 += P.KC) {
        const int kc = min(P.KC, Kp - k0);
        const int G = kc / 8;
#pragma unroll
        for (int i = 0; i < kDStageIt; ++i) {
            const int it = tid + i * kDThreads;
            if (it < kRows * G) {
                const int row = it / G, g = it - row * G;
                uint4 hi, lo;
                split8(sv[i], hi, lo);
                unsigned char *dst = buf + row * P.stride + g * 32;
                *(uint4 *)dst = hi;
                *(uint4 *)(dst + 16) = lo;
            }
        }
        __syncthreads();
        if (k0 + P.KC < Kp) dense_stage_load(P, sv, r0, k0 + P.KC, Kp, tid);
        if (gb < L.NT) {
            // k-steps batched behind a scheduling barrier like mma_k_loop: one L2 round trip per KB k-steps
            const unsigned char *arow = buf + col * P.stride + half * 32;
            const int ks0 = k0 / 16, nks = kc / 16;
            constexpr int KB = TG >= 4 ? 1 : (TG == 2 ? 2 : 4);
            const uint4 *wb[TG];
#pragma unroll
            for (int tt = 0; tt < TG; ++tt)
                wb[tt] = L.w + ((size_t)(min(gb + tt, L.NT - 1) * L.KS + ks0)) * 128 + lane;
            for (int ksb = 0; ksb < nks; ksb += KB) {
                uint4 wh[KB][TG], wl[KB][TG], ah[KB], al[KB];
#pragma unroll
                for (int d = 0; d < KB; ++d) {
                    const int ks = ksb + d < nks ? ksb + d : nks - 1;
#pragma unroll
                    for (int tt = 0; tt < TG; ++tt) { wh[d][tt] = wb[tt][ks * 128]; wl[d][tt] = wb[tt][ks * 128 + 64]; }
                    ah[d] = *(const uint4 *)(arow + ks * 64);
                    al[d] = *(const uint4 *)(arow + ks * 64 + 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d = 0; d < KB; ++d) {
                    if (ksb + d < nks) {
#pragma unroll
                        for (int tt = 0; tt < TG; ++tt) {
                            if (gb + tt < L.NT) {
                                acc[tt] = mfma_bf16(wh[d][tt], ah[d], acc[tt]);
                                acc[tt] = mfma_bf16(wl[d][tt], ah[d], acc[tt]);
                                acc[tt] = mfma_bf16(wh[d][tt], al[d], acc[tt]);
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
}

// (ReLU and) write the wave's TG tiles of rows r0 .. r0+31; leaves the activated values in acc
template <int TG>
__device__ __forceinline__ void dense_tile_store(const DenseParams &P, long r0, int gb, int lane, f32x16 (&acc)[TG]) {
    const int half = lane >> 5, col = lane & 31;
    const LayerDesc &L = P.L;
    const long r = r0 + col;
#pragma unroll
    for (int tt = 0; tt < TG; ++tt) {
        if (gb + tt < L.NT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = (gb + tt) * 32 + 8 * q + 4 * half;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[tt][4 * q + e];
                    if (P.relu) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                    acc[tt][4 * q + e] = v[e];
                }
                if (r < P.rows) {
                    float *o = P.y + r * L.N + c0;
                    if ((L.N & 3) == 0 && c0 + 3 < L.N) {
                        *(float4 *)o = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c0 + e < L.N) o[e] = v[e];
                    }
                }
            }
        }
    }
}

template <int TG>
__device__ __forceinline__ void dense_body(const DenseParams &P, unsigned char *buf, int lane, int w,
                                           int tid) {
    const int gb = (blockIdx.y * kDW + w) * TG;
    for (long r0 = (long)blockIdx.x * kRows; r0 < P.rows; r0 += (long)gridDim.x * kRows) {
        f32x16 acc[TG];
        dense_tile_accumulate<TG>(P, buf, r0, gb, lane, tid, acc);
        dense_tile_store<TG>(P, r0, gb, lane, acc);
    }
}

// TG (tiles per wave) is chosen on the host; one kernel per TG so that each gets its own register allocation
template <int TG>
__global__ __launch_bounds__(kDThreads) void dense_kernel(DenseParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    dense_body<TG>(P, smem, lane, w, tid);
}

// ---- dense layer on 128-ROW blocks for the wide aggregation layers of a coalesced replay (32 frames: 8192 x 1536 -> 512,
//      16384 x 768 -> 256, 32768 x 384 -> 128).  dense_kernel streams ALL weights of its column tiles for every 32 rows
//      with one L2 round trip per couple of k-steps: 805 MB of weight fragments and 48 dependent round trips per
//      workgroup at 8192 x 1536 -> 512 (75 us, MFMA busy 0.20).  Here a workgroup owns 128 rows x 128 columns: eight
//      waves, wave = (column tile, row half), a weight fragment feeds two row tiles; the contraction runs in chunks of
//      64 channels through a double-buffered LDS image of the rows (chunk c + 1 staged in front of the matrix work on
//      chunk c, one LDS-only barrier per chunk), and both streams -- the fp32 rows and the weight fragments -- are
//      requested PD chunks ahead into register rings (branch-free blocks, prologue values laundered: see
//      mlp_wide128.hip; a ring slot is refilled AFTER its MFMAs so that no fragment is copied).  Same arithmetic as
//      dense_kernel (bias in the accumulator, k ascending, the three split-bf16 passes in the same order, D^T form):
//      bit-identical output.  75 / 47 / 32 us -> 54 / 35 / 24 us for the three layers above.  What bounds it now: the
//      workgroups draw 604 MB from the L2s per call at 8192 x 1536 -> 512 (rows 786 KB + weights 2 x 786 KB per
//      workgroup: both row halves fetch the fragments) = 11 TB/s, the L2 -> CU rate every streaming kernel of this
//      library tops out at; the four-wave form (each fragment fetched once, RT = 4) has one wave per SIMD and nobody to
//      overlap its LDS round trips with: 58 us.  Measured on the way: removing the MFMAs saves 11 us, either stream
//      12 us; hipcc left alone reuses one register pair for the LDS operand fragments (a round trip in front of every
//      three MFMAs) -- the sched_group_barrier pipeline below fixes the ISA but not the time, the L2 does.
constexpr int kD128KC = 64;
constexpr int kD128Stride = kD128KC * 4 + 16;                 // bytes of a row in one LDS buffer (hi / lo planes, padded)
constexpr size_t kD128Lds = (size_t)2 * 128 * kD128Stride;
typedef unsigned d128_u32x4 __attribute__((ext_vector_type(4)));
typedef float d128_f32x4 __attribute__((ext_vector_type(4)));

// NW waves, CT column tiles per workgroup (4: 128 columns, 2: 64), PD chunks of lookahead (K % (64 * PD) == 0)
template <int NW, int CT, int PD>
__global__ __launch_bounds__(NW * 64, NW / 4) void dense128_kernel(DenseParams P) {
    constexpr int RT = 4 * CT / NW;                           // row tiles per wave
    static_assert(RT >= 1 && RT * NW == 4 * CT, "waves = column tiles x row parts");
    constexpr int NI = 1024 / (NW * 64);                      // (row, 8-channel group) items a thread stages per chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col = lane & 31;
    const LayerDesc &L = P.L;
    // (row block, column block) of this workgroup.  The column blocks of one row block read the SAME 128 input rows: in
    // launch order (x fastest) they were a whole pass over the rows apart and every column block fetched the input from
    // HBM again -- 898 MB of counter traffic against 272 algorithmic at 32768 x 1536 -> 512 (profiles/r05_rooflines_128f).
    // Now they are consecutive workgroups of ONE XCD (block L is observed to run on XCD L % 8, sa_common.h): they walk the
    // row block's chunks together and the later ones hit that XCD's L2.  A wrong guess about placement only costs the saving.
    int rb = blockIdx.x, cb = blockIdx.y;
    if (gridDim.y > 1 && (gridDim.x & 7) == 0) {
        const unsigned Lid = blockIdx.y * gridDim.x + blockIdx.x, per = 8u * gridDim.y;
        const unsigned within = Lid % per;
        rb = (int)((Lid / per) * 8u + (within & 7u));
        cb = (int)(within >> 3);
    }
    const int ct = cb * CT + (w % CT);                        // this wave's column tile (host: NT % CT == 0)
    const int rt0 = (w / CT) * RT;                            // ... and its first row tile of the block
    const long r0 = (long)rb * 128;
    const int nch = L.K / kD128KC;                            // host: K % (64 * PD) == 0

    // this thread's (row, 8-channel group) items of a chunk: rows past the end read the last row, never stored
    const float *sp[NI];
    int soff[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int item = tid + NW * 64 * j, row = item >> 3, g = item & 7;
        long r = r0 + row;
        if (r >= P.rows) r = P.rows - 1;
        sp[j] = P.x + r * L.K + g * 8;
        soff[j] = row * kD128Stride + g * 32;
    }
    f32x16 acc[RT];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 bv = *(const float4 *)(L.bias + ct * 32 + 8 * q + 4 * half);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            acc[rt][4 * q + 0] = bv.x; acc[rt][4 * q + 1] = bv.y; acc[rt][4 * q + 2] = bv.z; acc[rt][4 * q + 3] = bv.w;
        }
    }
    const d128_u32x4 *wb = (const d128_u32x4 *)L.w + (size_t)ct * L.KS * 128 + lane;
    d128_f32x4 xa[PD][NI][2];
    d128_u32x4 wq[PD][4][2];
#pragma unroll
    for (int d = 0; d < PD; ++d) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            xa[d][j][0] = *(const d128_f32x4 *)(sp[j] + d * kD128KC);
            xa[d][j][1] = *(const d128_f32x4 *)(sp[j] + d * kD128KC + 4);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            wq[d][ks][0] = wb[(d * 4 + ks) * 128];
            wq[d][ks][1] = wb[(d * 4 + ks) * 128 + 64];
        }
    }
#pragma unroll
    for (int d = 0; d < PD; ++d) {
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(xa[d][j][0]), "+v"(xa[d][j][1]));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(wq[d][ks][0]), "+v"(wq[d][ks][1]));
    }
    // stage(slot, chunk to refill with, buffer): the ring slot's fp32 rows -> split bf16 planes -> LDS, then the slot is
    // requested again PD chunks ahead (clamped at the end: harmless re-reads)
    auto stage = [&](d128_f32x4 (&xs)[NI][2], int refill, unsigned char *dst) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const float v[8] = {xs[j][0][0], xs[j][0][1], xs[j][0][2], xs[j][0][3], xs[j][1][0], xs[j][1][1], xs[j][1][2], xs[j][1][3]};
            uint4 hi, lo;
            split8(v, hi, lo);
            *(uint4 *)(dst + soff[j]) = hi;
            *(uint4 *)(dst + soff[j] + 16) = lo;
        }
        const int cn = refill < nch ? refill : nch - 1;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            xs[j][0] = *(const d128_f32x4 *)(sp[j] + cn * kD128KC);
            xs[j][1] = *(const d128_f32x4 *)(sp[j] + cn * kD128KC + 4);
        }
    };
    stage(xa[0], PD, smem);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");              // LDS-only barrier: the rings stay in flight
    // Chunk c + 1 is staged into the other buffer in front of the matrix work on chunk c: one barrier per chunk.
    for (int c0 = 0; c0 < nch; c0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int c = c0 + d;
            const int dn = (d + 1) % PD;                 // ring slot of chunk c + 1
            const unsigned char *buf = smem + (c & 1) * (128 * kD128Stride);
            unsigned char *bufn = smem + ((c + 1) & 1) * (128 * kD128Stride);
            stage(xa[dn], c + 1 + PD, bufn);
            // The matrix work of the chunk as its own scheduling region with an explicit pipeline: left alone, hipcc reuses
            // ONE pair of registers for the operand fragments (ds_read x2 -> wait -> 3 MFMA -> ds_read x2 ...: an LDS round
            // trip in front of every three MFMAs, ~2 000 cycles per chunk against 770 of matrix issue); here the fragments
            // of group g + 1 are requested before the MFMAs of group g.
            __builtin_amdgcn_sched_barrier(0);
            {
                const unsigned char *arow = buf + (rt0 * 32 + col) * kD128Stride + half * 32;
                uint4 ah[4 * RT], al[4 * RT];
#pragma unroll
                for (int g = 0; g < 4 * RT; ++g) {
                    ah[g] = *(const uint4 *)(arow + (g % RT) * 32 * kD128Stride + (g / RT) * 64);
                    al[g] = *(const uint4 *)(arow + (g % RT) * 32 * kD128Stride + (g / RT) * 64 + 16);
                }
#pragma unroll
                for (int g = 0; g < 4 * RT; ++g) {
                    const int ks = g / RT, rt = g % RT;
                    const uint4 wh = __builtin_bit_cast(uint4, wq[d][ks][0]), wl = __builtin_bit_cast(uint4, wq[d][ks][1]);
                    acc[rt] = mfma_bf16(wh, ah[g], acc[rt]);
                    acc[rt] = mfma_bf16(wl, ah[g], acc[rt]);
                    acc[rt] = mfma_bf16(wh, al[g], acc[rt]);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);            // DS reads of groups 0, 1
#pragma unroll
                for (int g = 0; g + 2 < 4 * RT; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);        // MFMAs of group g
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        // DS reads of group g + 2
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // the ring slot of the weight fragments is requested again (PD chunks ahead, clamped) once its MFMAs are
            // issued: no copy of the fragments, and the loads of a slot are consumed in the order they were issued
            {
                const int cn = c + PD < nch ? c + PD : nch - 1;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    wq[d][ks][0] = wb[(cn * 4 + ks) * 128];
                    wq[d][ks][1] = wb[(cn * 4 + ks) * 128 + 64];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const long r = r0 + (rt0 + rt) * 32 + col;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[rt][4 * q + e];
                if (P.relu) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
            }
            if (r < P.rows) *(float4 *)(P.y + r * L.N + ct * 32 + 8 * q + 4 * half) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// ---- vote_layer tail in ONE launch (layers_util.py:17-23): the last hidden conv1d (K -> H <= 128, BN folded, ReLU),
//      the offset conv1d (H -> 3, no activation) and out = xyz + clip(offsets) -- three launches before.  A workgroup
//      owns 32 rows: its four waves produce the 32 x H hidden tile exactly as dense_kernel<1> does (and write it, it is
//      the layer's feature output), park it in LDS as the split-bf16 B operand of the next layer (what the second launch
//      re-created from the fp32 tensor: the same bits), and wave 0 runs the H -> 3 layer with the same MFMA sequence
//      and finishes the translation.  Bit-identical to the three-launch form.
struct VoteTailParams {
    DenseParams D;          // hidden layer (relu = 1)
    LayerDesc L2;           // offset layer, N = 3
    float *offsets;         // [rows, 3]
    const float *xyz;       // [rows, 3]
    float *out;             // [rows, 3]
    float lo[3];
};
__global__ __launch_bounds__(kDThreads) void vote_tail_kernel(VoteTailParams V) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const DenseParams &P = V.D;
    const int stride2 = P.L.NT * 32 * 4 + 16;          // hidden tile rows in LDS: NT * 32 channels, hi/lo planes
    for (long r0 = (long)blockIdx.x * kRows; r0 < P.rows; r0 += (long)gridDim.x * kRows) {
        f32x16 acc[1];
        dense_tile_accumulate<1>(P, smem, r0, w, lane, tid, acc);
        dense_tile_store<1>(P, r0, w, lane, acc);
        if (w < P.L.NT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v[4] = {acc[0][4 * q + 0], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]};
                store_quad_bf16(smem + col * stride2 + (4 * w + q) * 32 + half * 8, v);
            }
        }
        __syncthreads();
        if (w == 0) {
            const LayerDesc &L2 = V.L2;
            f32x16 a2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *(const float4 *)(L2.bias + 8 * q + 4 * half);
                a2[4 * q + 0] = bv.x; a2[4 * q + 1] = bv.y; a2[4 * q + 2] = bv.z; a2[4 * q + 3] = bv.w;
            }
            const unsigned char *arow = smem + col * stride2 + half * 32;
            const uint4 *wb = L2.w + lane;
            for (int ks = 0; ks < L2.KS; ++ks) {
                const uint4 wh = wb[ks * 128], wl = wb[ks * 128 + 64];
                const uint4 ah = *(const uint4 *)(arow + ks * 64), al = *(const uint4 *)(arow + ks * 64 + 16);
                a2 = mfma_bf16(wh, ah, a2);
                a2 = mfma_bf16(wl, ah, a2);
                a2 = mfma_bf16(wh, al, a2);
            }
            const long r = r0 + col;
            if (half == 0 && r < P.rows) {
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    const float o = a2[e];
                    V.offsets[r * 3 + e] = o;
                    V.out[r * 3 + e] = V.xyz[r * 3 + e] + sa::fmin_nn(sa::fmax_nn(o, V.lo[e]), -V.lo[e]);
                }
            }
        }
        __syncthreads();
    }
}

// vote_layer tail (layers_util.py:21-23): out = xyz + clip(off, lo, -lo), lo = MAX_TRANSLATE_RANGE (< 0)
__global__ void vote_translate_kernel(long total, const float *__restrict__ xyz,
                                      const float *__restrict__ off, float lx, float ly, float lz,
                                      float *__restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const float lo = c == 0 ? lx : (c == 1 ? ly : lz);
        float o = off[i];
        o = sa::fmin_nn(sa::fmax_nn(o, lo), -lo);
        out[i] = xyz[i] + o;
    }
}

// ---- row plan (mlp_plan.h): ball -> ceil(clamp(cnt, 1, ns) / GR) granules of GR rows (GR = 8, or 4) ----------------
// Packing is "next fit" into 32-row tiles (GPT = 32 / GR granules): a ball of <= GPT granules never straddles a tile
// boundary (the rest of the tile is padded with invalid entries), so its maximum is complete inside one wave and is
// written with a plain store; only balls of more than 32 distinct rows are split (atomic max on a row zeroed here).
// Next fit is a sequential rule; it is evaluated in parallel as a scan over FUNCTIONS phase -> (phase, advance) (phase =
// fill of the current tile, 0..GPT-1): a thread folds its 8 balls for each of the GPT start phases, waves scan by
// composition.
#ifndef SA_PLAN_THREADS
#define SA_PLAN_THREADS 512
#endif
#ifndef SA_PLAN_BPT
#define SA_PLAN_BPT 8
#endif
constexpr int kPlanThreads = SA_PLAN_THREADS, kPlanBallsPerThread = SA_PLAN_BPT;
constexpr int kPlanChunk = kPlanThreads * kPlanBallsPerThread;      // 4096 balls per workgroup

template <int GPT> struct PlanFn { int t[GPT]; };     // t[p] = advance << SH | end phase, for start phase p
template <int GPT> struct PlanSh { static constexpr int SH = GPT == 4 ? 2 : 3; };
template <int GPT>
__device__ __forceinline__ int plan_fn_at(const PlanFn<GPT> &f, int p) {
    int v = f.t[0];
#pragma unroll
    for (int i = 1; i < GPT; ++i) v = p == i ? f.t[i] : v;
    return v;
}
// first a, then b
template <int GPT>
__device__ __forceinline__ PlanFn<GPT> plan_fn_compose(const PlanFn<GPT> &a, const PlanFn<GPT> &b) {
    constexpr int SH = PlanSh<GPT>::SH;
    PlanFn<GPT> c;
#pragma unroll
    for (int p = 0; p < GPT; ++p) {
        const int x = a.t[p], y = plan_fn_at<GPT>(b, x & (GPT - 1));
        c.t[p] = (((x >> SH) + (y >> SH)) << SH) | (y & (GPT - 1));
    }
    return c;
}
// state (advance << SH | phase) followed by function y
template <int GPT>
__device__ __forceinline__ int plan_state_apply(int st, int y) {
    constexpr int SH = PlanSh<GPT>::SH;
    return (((st >> SH) + (y >> SH)) << SH) | (y & (GPT - 1));
}
// one ball of g granules placed from phase ph: returns the padding in front of it; updates ph.  nextfit (rounds 2-5): a
// ball of <= GPT granules never straddles a tile boundary -- the open tile is padded instead.  Round 6 default: TIGHT --
// no padding at all; a ball that crosses a tile boundary is a split ball (its partial maxima meet through the atomic
// max that balls of more than 32 rows always used).  On the generator's frames next fit padded 5-22 % of the rows of
// the wide scales (layer 4 scale 1: 11 296 rows per two frames against 9 280; tools/plan_padding_sim.py).
template <int GPT>
__device__ __forceinline__ int plan_place(int g, int &ph, int nextfit) {
    int pad = 0;
    if (nextfit && g <= GPT && ph + g > GPT) { pad = GPT - ph; ph = 0; }
    ph = (ph + g) & (GPT - 1);
    return pad;
}
template <int GPT>
__device__ __forceinline__ int plan_granules_of(const int *cnt, int ball, int nballs, int ns, int dense, int &rows) {
    constexpr int GR = 32 / GPT;
    if (ball >= nballs) return 0;
    int c = cnt[ball];
    c = c < 1 ? 1 : (c > ns ? ns : c);
    if (dense) c = ns;
    rows += c;
    return (c + GR - 1) / GR;
}
// inclusive scan by composition over the 64 lanes of a wave
template <int GPT>
__device__ __forceinline__ PlanFn<GPT> plan_wave_scan(PlanFn<GPT> incl, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        PlanFn<GPT> o;
#pragma unroll
        for (int p = 0; p < GPT; ++p) o.t[p] = __shfl_up(incl.t[p], d);
        if (lane >= d) incl = plan_fn_compose<GPT>(o, incl);
    }
    return incl;
}

// A scale's balls are cut into chunks of 4096; workgroup (chunk, scale) packs ITS chunk.  The packing state in front of
// the chunk (granules placed so far, fill of the open tile) is a function of all earlier balls.  Two launches, no
// atomics, no flags to reset between calls, the same plan on every run:
//   mlp_plan_summary_kernel   workgroup (chunk, scale) folds its 4096 balls into ONE phase -> (phase, advance) function
//                             (+ distinct rows, split balls) and stores the words in the scale's scratch;
//   mlp_plan_kernel           composes the summaries of the chunks in front of it (a thread folds a run of them, a
//                             wave scan and an 8-entry serial composition give the total) and packs its chunk.
// History: until round 3 one workgroup per scale walked its chunks one after the other (61 us for layer1's 3 x 32 768
// balls); then every workgroup re-folded all BALLS in front of it -- quadratic, 22 us per call at 8 frames but 50 us at
// the 32 frames of a coalesced replay (pipeline.py), 0.2 ms of plans per replay.
constexpr int kPlanMaxScales = 4;
constexpr int kPlanSumInts = 12;     // per chunk: t[0..7] (advance << SH | end phase for start phase p), rows, split balls, 2 unused
struct PlanJob {
    const int *cnt;
    int *hdr, *gran, *sum;
    int ns, out_off, N;
    int gr4;                         // 0: granules of 8 rows (4 per tile), 1: of 4 rows (8 per tile), 2: of 2 rows (16 per tile; tight packing only)
};
struct PlanJobs {
    PlanJob j[kPlanMaxScales];
    int nballs, dense, out_stride;
    int nextfit;                     // 1: the next-fit packing of rounds 2-5 (flags bit 7; A/B measurements, tests)
    float *out;
};

// this thread's 8 balls of the chunk: granules, distinct rows, and their fold for the GPT start phases
template <int GPT>
__device__ __forceinline__ PlanFn<GPT> plan_fold_thread(const PlanJob &job, int ball0, int nballs, int dense, int nextfit,
                                                        int (&g)[kPlanBallsPerThread], int &rows, int &nsp) {
    constexpr int SH = PlanSh<GPT>::SH;
#pragma unroll
    for (int k = 0; k < kPlanBallsPerThread; ++k) {
        g[k] = plan_granules_of<GPT>(job.cnt, ball0 + k, nballs, job.ns, dense, rows);
        nsp += g[k] > GPT;
    }
    PlanFn<GPT> f;
#pragma unroll
    for (int p = 0; p < GPT; ++p) {
        int ph = p, adv = 0;
#pragma unroll
        for (int k = 0; k < kPlanBallsPerThread; ++k)
            if (g[k] > 0) { adv += plan_place<GPT>(g[k], ph, nextfit); adv += g[k]; }
        f.t[p] = (adv << SH) | ph;
    }
    return f;
}

// TIGHT packing needs no phase functions: a ball's position is the plain prefix sum of the granule counts in front of it
// (the phase is position mod GPT).  Summary of a chunk: o[0] = its granules, o[8] = rows, o[9] = balls of > GPT granules.
template <int GPT>
__device__ __forceinline__ void plan_summary_tight(const PlanJobs &J, const PlanJob &job, int (*wsum)[2], int (*wg)[8]) {
    constexpr int NWV = kPlanThreads / 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ball0 = blockIdx.x * kPlanChunk + tid * kPlanBallsPerThread;
    int gs = 0, rows = 0, nsp = 0;
#pragma unroll
    for (int k = 0; k < kPlanBallsPerThread; ++k) {
        const int g = plan_granules_of<GPT>(job.cnt, ball0 + k, J.nballs, job.ns, J.dense, rows);
        gs += g;
        nsp += g > GPT;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { gs += __shfl_xor(gs, d); rows += __shfl_xor(rows, d); nsp += __shfl_xor(nsp, d); }
    if (lane == 0) { wsum[w][0] = rows; wsum[w][1] = nsp; wg[w][0] = gs; }
    __syncthreads();
    if (tid == 0) {
        int G = 0, r = 0, n = 0;
#pragma unroll
        for (int i = 0; i < NWV; ++i) { G += wg[i][0]; r += wsum[i][0]; n += wsum[i][1]; }
        int *o = job.sum + (size_t)blockIdx.x * kPlanSumInts;
        o[0] = G; o[8] = r; o[9] = n;
    }
}

template <int GPT>
__device__ __forceinline__ void plan_summary_body(const PlanJobs &J, const PlanJob &job, int (*wfn)[8], int (*wsum)[2]) {
    constexpr int NWV = kPlanThreads / 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int chunk0 = blockIdx.x * kPlanChunk;
    int g[kPlanBallsPerThread], rows = 0, nsp = 0;
    const PlanFn<GPT> f = plan_fold_thread<GPT>(job, chunk0 + tid * kPlanBallsPerThread, J.nballs, J.dense, J.nextfit, g, rows, nsp);
    const PlanFn<GPT> incl = plan_wave_scan<GPT>(f, lane);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { rows += __shfl_xor(rows, d); nsp += __shfl_xor(nsp, d); }
    if (lane == 63) {
#pragma unroll
        for (int p = 0; p < GPT; ++p) wfn[w][p] = incl.t[p];
    }
    if (lane == 0) { wsum[w][0] = rows; wsum[w][1] = nsp; }
    __syncthreads();
    if (tid < GPT) {                                         // thread p: the chunk's function at start phase p
        int st = tid, r = 0, n = 0;
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            st = plan_state_apply<GPT>(st, wfn[i][st & (GPT - 1)]);
            r += wsum[i][0]; n += wsum[i][1];
        }
        int *o = job.sum + (size_t)blockIdx.x * kPlanSumInts;
        o[tid] = st;                                         // advance << SH | end phase (the start phase's bits carry no advance)
        if (tid == 0) { o[8] = r; o[9] = n; }
    }
}

__global__ __launch_bounds__(kPlanThreads) void mlp_plan_summary_kernel(PlanJobs J) {
    __shared__ int wfn[kPlanThreads / 64][8];
    __shared__ int wsum[kPlanThreads / 64][2];
    const PlanJob job = J.j[blockIdx.y];
    if (blockIdx.x * kPlanChunk + kPlanChunk >= J.nballs) return;   // nobody reads the last chunk's summary
    if (!J.nextfit) {
        if (job.gr4 == 2) plan_summary_tight<16>(J, job, wsum, wfn);
        else if (job.gr4) plan_summary_tight<8>(J, job, wsum, wfn);
        else plan_summary_tight<4>(J, job, wsum, wfn);
        return;
    }
    if (job.gr4) plan_summary_body<8>(J, job, wfn, wsum);
    else plan_summary_body<4>(J, job, wfn, wsum);
}

template <int GPT>
__device__ __forceinline__ void plan_pack_tight(const PlanJobs &J, const PlanJob &job, int (*wg)[8], int (*wsum)[2], int &nsplit_s) {
    constexpr int NWV = kPlanThreads / 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nballs = J.nballs;
    const int chunk0 = blockIdx.x * kPlanChunk;
    const bool last_chunk = chunk0 + kPlanChunk >= nballs;
    if (tid == 0) nsplit_s = 0;
    // ---- granules, rows and > GPT balls of the chunks in front of this one: plain sums of their summaries
    int base = 0, rows_before = 0, nsplit_before = 0;
    if (blockIdx.x > 0) {
        int G = 0, r = 0, n = 0;
        for (int c = tid; c < (int)blockIdx.x; c += kPlanThreads) {
            const int *o = job.sum + (size_t)c * kPlanSumInts;
            G += o[0]; r += o[8]; n += o[9];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { G += __shfl_xor(G, d); r += __shfl_xor(r, d); n += __shfl_xor(n, d); }
        if (lane == 0) { wg[w][0] = G; wsum[w][0] = r; wsum[w][1] = n; }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NWV; ++i) { base += wg[i][0]; rows_before += wsum[i][0]; nsplit_before += wsum[i][1]; }
        __syncthreads();
    }
    // ---- this chunk: exclusive prefix sum of the threads' granule counts
    const int ball0 = chunk0 + tid * kPlanBallsPerThread;
    int g[kPlanBallsPerThread], gs = 0, rows = 0;
#pragma unroll
    for (int k = 0; k < kPlanBallsPerThread; ++k) { g[k] = plan_granules_of<GPT>(job.cnt, ball0 + k, nballs, job.ns, J.dense, rows); gs += g[k]; }
    int incl = gs;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    int rsum = rows;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) rsum += __shfl_xor(rsum, d);
    if (lane == 63) wg[w][0] = incl;
    if (lane == 0) wsum[w][0] = rsum;
    __syncthreads();
    int pos = base + incl - gs, used = base, rows_chunk = 0;
#pragma unroll
    for (int i = 0; i < NWV; ++i) { if (i < w) pos += wg[i][0]; used += wg[i][0]; rows_chunk += wsum[i][0]; }
    int nsp_mine = 0;
#pragma unroll
    for (int k = 0; k < kPlanBallsPerThread; ++k) {
        if (g[k] > 0) {
            const int ball = ball0 + k;
            const int split = (pos & (GPT - 1)) + g[k] > GPT ? 1 : 0;     // its granules lie in more than one tile
            for (int j = 0; j < g[k]; ++j) job.gran[pos + j] = (ball << 7) | (j << 1) | split;
            pos += g[k];
            nsp_mine += split;
        }
    }
    if (last_chunk) {                                              // the header needs the chunk's split count
        if (nsp_mine) atomicAdd(&nsplit_s, nsp_mine);
        __syncthreads();
        if (tid == 0) {
            const int total = (used + GPT - 1) & ~(GPT - 1);
            for (int q = used; q < total; ++q) job.gran[q] = -1;
            job.hdr[0] = total; job.hdr[1] = nsplit_before + nsplit_s; job.hdr[2] = rows_before + rows_chunk; job.hdr[3] = 32 / GPT;
        }
    }
}

template <int GPT>
__device__ __forceinline__ void plan_pack_body(const PlanJobs &J, const PlanJob &job, int (*wfn)[8], int (*wsum)[2],
                                               int &nsplit_s, int *split_ball) {
    constexpr int NWV = kPlanThreads / 64;
    constexpr int SH = PlanSh<GPT>::SH;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nballs = J.nballs;
    const int chunk0 = blockIdx.x * kPlanChunk;
    const bool last_chunk = chunk0 + kPlanChunk >= nballs;
    if (tid == 0) nsplit_s = 0;

    // ---- the chunks in front of this one: packing state `base`, distinct rows and split balls so far, from their
    //      summaries (mlp_plan_summary_kernel, earlier on the stream)
    int base = 0, rows_before = 0, nsplit_before = 0;
    if (blockIdx.x > 0) {
        const int nprev = blockIdx.x;
        const int per = (nprev + kPlanThreads - 1) / kPlanThreads;
        const int c0 = tid * per, c1 = min(c0 + per, nprev);
        PlanFn<GPT> f;
        int rows = 0, nsp = 0;
#pragma unroll
        for (int p = 0; p < GPT; ++p) f.t[p] = p;
        for (int c = c0; c < c1; ++c) {
            const int *o = job.sum + (size_t)c * kPlanSumInts;
            PlanFn<GPT> h;
#pragma unroll
            for (int p = 0; p < GPT; p += 4) {
                const int4 t4 = *(const int4 *)(o + p);
                h.t[p] = t4.x; h.t[p + 1] = t4.y; h.t[p + 2] = t4.z; h.t[p + 3] = t4.w;
            }
            const int2 r2 = *(const int2 *)(o + 8);
            f = plan_fn_compose<GPT>(f, h);
            rows += r2.x; nsp += r2.y;
        }
        const PlanFn<GPT> incl = plan_wave_scan<GPT>(f, lane);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { rows += __shfl_xor(rows, d); nsp += __shfl_xor(nsp, d); }
        if (lane == 63) {
#pragma unroll
            for (int p = 0; p < GPT; ++p) wfn[w][p] = incl.t[p];
        }
        if (lane == 0) { wsum[w][0] = rows; wsum[w][1] = nsp; }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            base = plan_state_apply<GPT>(base, wfn[i][base & (GPT - 1)]);
            rows_before += wsum[i][0];
            nsplit_before += wsum[i][1];
        }
        __syncthreads();
    }

    // ---- this chunk
    const int ball0 = chunk0 + tid * kPlanBallsPerThread;
    int g[kPlanBallsPerThread], rows = 0, nsp_unused = 0;
    const PlanFn<GPT> f = plan_fold_thread<GPT>(job, ball0, nballs, J.dense, J.nextfit, g, rows, nsp_unused);
    const PlanFn<GPT> incl = plan_wave_scan<GPT>(f, lane);
    int rsum = rows;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) rsum += __shfl_xor(rsum, d);
    if (lane == 63) {
#pragma unroll
        for (int p = 0; p < GPT; ++p) wfn[w][p] = incl.t[p];
    }
    if (lane == 0) wsum[w][0] = rsum;
    __syncthreads();
    int st = base, bend = base, rows_chunk = 0;       // state in front of this wave / behind the chunk
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
        bend = plan_state_apply<GPT>(bend, wfn[i][bend & (GPT - 1)]);
        if (i < w) st = bend;
        rows_chunk += wsum[i][0];
    }
    {
        PlanFn<GPT> excl;                               // exclusive prefix of this thread inside its wave
#pragma unroll
        for (int p = 0; p < GPT; ++p) { const int v = __shfl_up(incl.t[p], 1); excl.t[p] = lane == 0 ? p : v; }
        st = plan_state_apply<GPT>(st, plan_fn_at<GPT>(excl, st & (GPT - 1)));
    }
    int pos = st >> SH, ph = st & (GPT - 1);
#pragma unroll
    for (int k = 0; k < kPlanBallsPerThread; ++k) {
        if (g[k] > 0) {
            const int ball = ball0 + k;
            const int ph0 = ph;
            const int pad = plan_place<GPT>(g[k], ph, J.nextfit);
            for (int j = 0; j < pad; ++j) job.gran[pos + j] = -1;
            pos += pad;
            // split: the ball's granules lie in more than one tile (always for > GPT granules; tight packing: whenever it
            // crosses a tile boundary) -- its rows are zeroed below and its partial maxima meet through atomic max
            const int split = (J.nextfit ? g[k] > GPT : ph0 + g[k] > GPT) ? 1 : 0;
            for (int j = 0; j < g[k]; ++j) job.gran[pos + j] = (ball << 7) | (j << 1) | split;
            pos += g[k];
            if (split) atomicAdd(&nsplit_s, 1);
        }
    }
    __syncthreads();
    // (the rows of split balls are zeroed by mlp_plan_zero_kernel, over the whole chip: with tight packing a ball is split
    //  at almost every tile boundary of the wide scales -- 150 MB of rows per 128 frames -- and the few workgroups of this
    //  kernel took 0.37 ms for them)
    const int nsp = nsplit_s;
    (void)split_ball;
    // header (the workgroup of the last chunk): granules padded to whole tiles with invalid entries, split balls, rows
    if (last_chunk && tid == 0) {
        const int used = bend >> SH, total = (used + GPT - 1) & ~(GPT - 1);
        for (int q = used; q < total; ++q) job.gran[q] = -1;
        job.hdr[0] = total; job.hdr[1] = nsplit_before + nsp; job.hdr[2] = rows_before + rows_chunk; job.hdr[3] = 32 / GPT;
    }
}

__global__ __launch_bounds__(kPlanThreads) void mlp_plan_kernel(PlanJobs J) {
    __shared__ int wfn[kPlanThreads / 64][8];
    __shared__ int wsum[kPlanThreads / 64][2];
    __shared__ int nsplit_s;
    __shared__ int split_ball[kPlanChunk];
    const PlanJob job = J.j[blockIdx.y];
    if (blockIdx.x * kPlanChunk >= J.nballs) return;
    if (!J.nextfit) {
        if (job.gr4 == 2) plan_pack_tight<16>(J, job, wfn, wsum, nsplit_s);
        else if (job.gr4) plan_pack_tight<8>(J, job, wfn, wsum, nsplit_s);
        else plan_pack_tight<4>(J, job, wfn, wsum, nsplit_s);
        return;
    }
    if (job.gr4) plan_pack_body<8>(J, job, wfn, wsum, nsplit_s, split_ball);
    else plan_pack_body<4>(J, job, wfn, wsum, nsplit_s, split_ball);
}

// Rows of split balls are zeroed: their partial maxima meet through an atomic max (mlp_plan.h).  One wave per 64 plan
// entries; the first granule of a split ball (ordinal 0) names the row.
__global__ __launch_bounds__(256) void mlp_plan_zero_kernel(PlanJobs J) {
    const PlanJob job = J.j[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int ngran = __builtin_amdgcn_readfirstlane(job.hdr[0]);
    for (int base = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; base < ngran; base += gridDim.x * 256) {
        const int e = base + lane < ngran ? job.gran[base + lane] : -1;
        unsigned long long m = __ballot(e >= 0 && (e & 1) && ((e >> 1) & (sa::kPlanMaxOrd - 1)) == 0);
        while (m != 0ull) {
            const int l = (int)__builtin_ctzll(m);
            m &= m - 1ull;
            const int ball = __builtin_amdgcn_readlane(e, l) >> 7;
            float *row = J.out + (size_t)ball * J.out_stride + job.out_off;
            for (int c = lane; c < job.N; c += 64) row[c] = 0.0f;
        }
    }
}

int roundup(int x, int q) { return (x + q - 1) / q * q; }

}  // namespace

// mlp_rowwave.hip: LDS-resident-weight / register-resident-activation kernel for the narrow and mid scales
int sa_rowwave_try(int b, int n, int m, int ns, int c, const float *xyz, const float *feat, const float *new_xyz,
                   const int *idx, const int *cnt, int nl, const int *dims, const void *const *wpack,
                   const float *const *bias, float *out, int out_stride, int out_off, const int *plan_hdr,
                   const int *plan_gran, long max_tiles, int fp16, int gr4, int dry, int *overflow, hipStream_t stream, int *st);

// mlp_wide128.hip: the widest fp16 scales on 128-row items (a weight fragment feeds four MFMA tiles)
int sa_wide128_try(int b, int n, int m, int ns, int c, const float *xyz, const float *feat, const float *new_xyz,
                   const int *idx, const int *cnt, int nl, const int *dims, const void *const *wpack,
                   const float *const *bias, float *out, int out_stride, int out_off, const int *plan_hdr,
                   const int *plan_gran, long max_tiles, int fp16, int force, int dry, int *overflow, hipStream_t stream, int *st);

// Upper bound of the plan length in granules of 8 rows: next fit never leaves two consecutive tiles with a combined fill
// <= one tile, so the list is shorter than twice the granules (+ the padding of the last tile).  (max + 3) / 4 bounds
// the TILES of a plan of either granule size: a 4-row plan of the same balls never has more tiles than the 8-row bound
// (ceil(ns / 4) <= 2 ceil(ns / 8)).
static long sa_plan_max_granules(long nballs, int ns) {
    const long g = nballs * ((ns + 7) / 8);
    return (ns <= 8 ? g : 2 * g) + 8;
}
// ... and in granules of 4 rows: what the entry list of a scale's scratch is sized for (any granule size fits: a tight
// 2-row plan holds nballs * ceil(ns / 2) <= 2 nballs * ceil(ns / 4) entries + the padding of the last tile)
static long sa_plan_max_entries(long nballs, int ns) {
    const long g = nballs * ((ns + 3) / 4);
    return 2 * g + 16;
}

// ints in front of the chunk summaries of a scale's scratch: header + one int per granule of the densest plan, 16-byte
// aligned
static size_t sa_plan_sum_offset_ints(long nballs, int ns) {
    return ((size_t)sa::kPlanHeaderInts + (size_t)sa_plan_max_entries(nballs, ns) + 8 + 3) & ~(size_t)3;
}
// Bytes of caller-owned scratch sa_group_mlp_max needs for the row plan of one scale (header + one int per granule
// of the densest plan + the summaries of its 4096-ball chunks).
extern "C" size_t sa_group_mlp_max_ws_bytes(int b, int m, int ns) {
    if (b <= 0 || m <= 0 || ns <= 0) return 0;
    const long nballs = (long)b * m;
    const size_t chunks = (size_t)((nballs + kPlanChunk - 1) / kPlanChunk);
    return (sa_plan_sum_offset_ints(nballs, ns) + chunks * kPlanSumInts) * sizeof(int);
}
// the two launches of a layer's (or a scale's) row plans
static void launch_plan(const PlanJobs &J, int nscale, hipStream_t stream) {
    const unsigned chunks = (unsigned)((J.nballs + kPlanChunk - 1) / kPlanChunk);
    if (chunks > 1) hipLaunchKernelGGL(mlp_plan_summary_kernel, dim3(chunks - 1, nscale), dim3(kPlanThreads), 0, stream, J);
    hipLaunchKernelGGL(mlp_plan_kernel, dim3(chunks, nscale), dim3(kPlanThreads), 0, stream, J);
    // the rows of split balls, zeroed over the whole chip (the entry list is at most 2 x the 4-row granules of the balls)
    long zg = ((long)J.nballs * 4 + 255) / 256;
    if (zg > 2048) zg = 2048;
    hipLaunchKernelGGL(mlp_plan_zero_kernel, dim3((unsigned)zg, nscale), dim3(256), 0, stream, J);
}

// mlp_gemm.hip: the wide scales as a chain of three large-tile GEMMs over packed fp16 intermediates
size_t sa_mlp_gemm_scratch_bytes(long max_tiles, const int *dims);
bool sa_mlp_gemm_eligible(int c, int nl, const int *dims, int fp16);
int sa_mlp_gemm_launch(int nscale, int b, int n, int m, const int *ns, int c, const float *xyz, const float *feat,
                       const float *new_xyz, const int *const *idx, const int *const *cnt, const int *dims,
                       const void *const *wpack, const float *const *bias, float *out, int out_stride, const int *out_off,
                       const int *const *plan_hdr, const int *const *plan_gran, const long *max_tiles,
                       void *const *scratch, int *overflow, hipStream_t stream);

static size_t plan_bytes_aligned(int b, int m, int ns) { return (sa_group_mlp_max_ws_bytes(b, m, ns) + 255) & ~(size_t)255; }

// Scratch for a scale that may take the GEMM chain: the row plan, then (256-byte aligned) the packed hidden activations
// H1 | H2 of the densest plan.  A caller that passes only sa_group_mlp_max_ws_bytes gets the fused kernels.
extern "C" size_t sa_group_mlp_gemm_ws_bytes(int b, int m, int ns, int c, int nl, const int *dims) {
    if (b <= 0 || m <= 0 || ns <= 0 || !dims) return 0;
    if (!sa_mlp_gemm_eligible(c, nl, dims, 1)) return sa_group_mlp_max_ws_bytes(b, m, ns);
    const long max_tiles = (sa_plan_max_granules((long)b * m, ns) + 3) / 4;
    return plan_bytes_aligned(b, m, ns) + sa_mlp_gemm_scratch_bytes(max_tiles, dims);
}
// does this call take the chain?  Opt-in: flags bit 4 (16).  Measured on layer4 of 3dssd.yaml (batch 8, both scales):
// chain 157 us (phases 40 / 37 / 80) against 146 us for the fused kernels -- the packed intermediates H1 / H2 (18 + 37 MB
// per scale) do not stay in the 4 MB per-XCD L2 between launches, every phase streams them through Infinity Cache / HBM
// (~300 MB per layer call), and its stores stall at the memory write rate; DESIGN.md section 6 has the breakdown.
static bool use_gemm_chain(int b, int m, int ns, int c, int nl, const int *dims, size_t ws_bytes, int flags) {
    if (!(flags & 16) || (flags & 1) || !(flags & 4)) return false;
    if (!sa_mlp_gemm_eligible(c, nl, dims, 1)) return false;
    return ws_bytes >= sa_group_mlp_gemm_ws_bytes(b, m, ns, c, nl, dims);
}

// Row plans of ALL scales of an SA layer (two launches: chunk summaries, then the packing; workgroups = chunks x scales).  cnt[i]: pts_cnt of scale i
// [b, m]; ws[i]: scratch of sa_group_mlp_max_ws_bytes(b, m, ns[i]) bytes; out / out_stride / out_off[i] / nout[i]:
// where scale i's pooled channels go (rows of balls with more than 32 distinct rows are zeroed here).  The
// sa_group_mlp_max calls of the layer then pass the same ws[i] and flags | 2.  scale_flags (may be null): bit 6 (64) of
// scale_flags[i] asks for granules of 4 rows for scale i (mlp_plan.h; ask sa_group_mlp_granule_rows which scales take
// them) -- the same bit then goes into that scale's flags of sa_group_mlp_max / sa_group_mlp_max_layer.
extern "C" int sa_group_mlp_plan2(int b, int m, int nscale, const int *ns, const int *const *cnt, void *const *ws,
                                  float *out, int out_stride, const int *out_off, const int *nout, int flags,
                                  const int *scale_flags, hipStream_t stream) {
    if (b <= 0 || m <= 0 || nscale < 1 || nscale > kPlanMaxScales || !ns || !cnt || !ws || !out || !out_off || !nout)
        return SA_ERR_INVALID;
    const long nballs = (long)b * m;
    if (nballs >= (1l << 24)) return SA_ERR_UNSUPPORTED;
    PlanJobs J{};
    for (int i = 0; i < nscale; ++i) {
        if (ns[i] <= 0 || !cnt[i] || !ws[i] || nout[i] <= 0) return SA_ERR_INVALID;
        // granule size of the scale's plan: bit 6 (64) = 4 rows, bit 8 (256) = 2 rows (round 6; tight packing only)
        const int gr4 = !scale_flags ? 0 : (scale_flags[i] & 256) ? 2 : (scale_flags[i] & 64) ? 1 : 0;
        if (gr4 == 2 && (flags & 128)) return SA_ERR_UNSUPPORTED;
        if (ns[i] > (8 >> gr4) * sa::kPlanMaxOrd) return SA_ERR_UNSUPPORTED;
        J.j[i].cnt = cnt[i]; J.j[i].hdr = (int *)ws[i]; J.j[i].gran = (int *)ws[i] + sa::kPlanHeaderInts;
        J.j[i].sum = (int *)ws[i] + sa_plan_sum_offset_ints(nballs, ns[i]);
        J.j[i].ns = ns[i]; J.j[i].out_off = out_off[i]; J.j[i].N = nout[i]; J.j[i].gr4 = gr4;
    }
    J.nballs = (int)nballs; J.dense = flags & 1; J.nextfit = (flags >> 7) & 1; J.out_stride = out_stride; J.out = out;
    launch_plan(J, nscale, stream);
    SA_CHECK_LAUNCH();
    return SA_OK;
}
extern "C" int sa_group_mlp_plan(int b, int m, int nscale, const int *ns, const int *const *cnt, void *const *ws,
                                 float *out, int out_stride, const int *out_off, const int *nout, int flags,
                                 hipStream_t stream) {
    return sa_group_mlp_plan2(b, m, nscale, ns, cnt, ws, out, out_stride, out_off, nout, flags, nullptr, stream);
}

// which kernel family takes one scale with these flags (the dispatch order of sa_group_mlp_max, nothing launched):
// the row-wave kernels of mlp_rowwave.hip read plans of either granule size and are fastest on 4-row granules;
// everything else reads 8-row plans only
static bool scale_takes_rowwave(int b, int n, int m, int ns, int c, int nl, const int *dims, const void *const *wpack,
                                size_t ws_bytes, int flags) {
    const bool fp16 = (flags & 4) != 0;
    const float *const bias3[3] = {nullptr, nullptr, nullptr};
    if (flags & 1) return false;                                      // dense A/B plans stay on 8-row granules
    if (nl != 3 || ns > 4 * sa::kPlanMaxOrd) return false;
    if (use_gemm_chain(b, m, ns, c, nl, dims, ws_bytes, flags)) return false;
    const long max_tiles = (sa_plan_max_granules((long)b * m, ns) + 3) / 4;
    int st = SA_OK;
    if (fp16 && !(flags & 8) && ((flags & 32) || (long)b * m * ns >= 4096) &&
        sa_wide128_try(b, n, m, ns, c, nullptr, nullptr, nullptr, nullptr, nullptr, nl, dims, wpack, bias3, nullptr, 0, 0,
                       nullptr, nullptr, max_tiles, 1, (flags & 32) ? 1 : 0, 1, nullptr, nullptr, &st))
        return false;
    return sa_rowwave_try(b, n, m, ns, c, nullptr, nullptr, nullptr, nullptr, nullptr, nl, dims, wpack, bias3, nullptr, 0, 0,
                          nullptr, nullptr, max_tiles, fp16 ? 1 : 0, 1, 1, nullptr, nullptr, &st) != 0;
}
// Rows per granule (4 or 8) the plan of this scale should be built with: 4 when a row-wave kernel will take the scale
// (pass flags | 64 to sa_group_mlp_plan2 and to the MLP call then), 8 otherwise.  wpack: the scale's packed layers (the
// streamed kernels need them back to back); ws_bytes: size of the scale's scratch; flags: as for sa_group_mlp_max.
extern "C" int sa_group_mlp_granule_rows(int b, int n, int m, int ns, int c, int nl, const int *dims, const void *const *wpack,
                                         size_t ws_bytes, int flags) {
    if (b <= 0 || n <= 0 || m <= 0 || ns <= 0 || c < 0 || nl < 1 || nl > kMaxLayers || !dims || !wpack) return 8;
    return scale_takes_rowwave(b, n, m, ns, c, nl, dims, wpack, ws_bytes, flags) ? 4 : 8;
}

// One scale of an SA layer.  Layer l: wpack[l] (device, fragment-packed hi/lo bf16, see header),
// bias[l] (device, fp32, zero-padded to a multiple of 32), dims[0] = C+3, dims[l+1] = output channels.
// out[(b*m + j)*out_stride + out_off + c] receives the pooled channel c.  Additional to the reference
// API (the reference has no fused op).  ws: sa_group_mlp_max_ws_bytes(b, m, ns) bytes of device scratch (the row
// plan; contents are private to the call).  flags bit 0: dense plan -- every ball is evaluated on all nsample rows
// like the reference does (A/B measurements); default: only the distinct rows of a ball (mlp_plan.h), same results.
// flags bit 1: the plan in ws was built by sa_group_mlp_plan for this layer (skips the per-scale plan launch).
// flags bit 2: wpack[] holds single-plane fp16 fragments (utils/weights.py precision "fp16") and the scale is
// evaluated with one fp16 MFMA pass per k-step instead of the three split-bf16 passes.
extern "C" int sa_group_mlp_max(int b, int n, int m, int ns, int c, const float *xyz, const float *feat,
                                const float *new_xyz, const int *idx, const int *cnt, int nl,
                                const int *dims, const void *const *wpack, const float *const *bias,
                                float *out, int out_stride, int out_off, void *ws, size_t ws_bytes, int flags,
                                int *overflow, hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || ns <= 0 || c < 0 || nl < 1 || nl > kMaxLayers) return SA_ERR_INVALID;
    if (!xyz || !new_xyz || !idx || !cnt || !out || !dims || !wpack || !bias) return SA_ERR_INVALID;
    if (c > 0 && !feat) return SA_ERR_INVALID;
    if (dims[0] != c + 3) return SA_ERR_INVALID;
    for (int l = 0; l < nl; ++l)
        if (dims[l + 1] <= 0 || !wpack[l] || !bias[l]) return SA_ERR_INVALID;
    if (!ws || ws_bytes < sa_group_mlp_max_ws_bytes(b, m, ns)) return SA_ERR_INVALID;
    const long nballs = (long)b * m;
    if (nballs >= (1l << 24) || ns > 8 * sa::kPlanMaxOrd) return SA_ERR_UNSUPPORTED;   // plan entry fields
    const long gmax = sa_plan_max_granules(nballs, ns);
    const long max_tiles = (gmax + 3) / 4;
    if (max_tiles > 0x0FFFFFFFl) return SA_ERR_UNSUPPORTED;
    int *hdr = (int *)ws, *gran = hdr + sa::kPlanHeaderInts;
    // granule size: flags bit 6 = 4-row granules (opt-in: mlp_plan.h; only the row-wave kernels read such plans), for a
    // plan built by the caller (flags bit 1) and for the plan of this call alike
    if (flags & 256) return SA_ERR_UNSUPPORTED;             // 2-row plans: sa_group_mlp_max_layer's one-launch kernels only
    const bool gr4 = (flags & 64) != 0;
    if (gr4 && !(flags & 2) && !scale_takes_rowwave(b, n, m, ns, c, nl, dims, wpack, ws_bytes, flags & ~64)) return SA_ERR_UNSUPPORTED;
    // ---- the row plan of this call (unless sa_group_mlp_plan built the plans of the whole layer already)
    if (!(flags & 2)) {
        PlanJobs J{};
        J.j[0].cnt = cnt; J.j[0].hdr = hdr; J.j[0].gran = gran; J.j[0].ns = ns; J.j[0].out_off = out_off; J.j[0].N = dims[nl];
        J.j[0].gr4 = gr4 ? 1 : 0;
        J.j[0].sum = hdr + sa_plan_sum_offset_ints(nballs, ns);
        J.nballs = (int)nballs; J.dense = flags & 1; J.nextfit = (flags >> 7) & 1; J.out_stride = out_stride; J.out = out;
        launch_plan(J, 1, stream);
        SA_CHECK_LAUNCH();
    }
    const bool fp16 = (flags & 4) != 0;
    if (gr4) {                                  // 4-row granules: only the row-wave kernels read them
        int st = SA_OK;
        if (sa_rowwave_try(b, n, m, ns, c, xyz, feat, new_xyz, idx, cnt, nl, dims, wpack, bias, out, out_stride,
                           out_off, hdr, gran, max_tiles, fp16 ? 1 : 0, 1, 0, overflow, stream, &st))
            return st;
        return SA_ERR_UNSUPPORTED;
    }
    if (use_gemm_chain(b, m, ns, c, nl, dims, ws_bytes, flags)) {
        const int *hdr_c = hdr, *gran_c = gran;
        void *scratch = (char *)ws + plan_bytes_aligned(b, m, ns);
        return sa_mlp_gemm_launch(1, b, n, m, &ns, c, xyz, feat, new_xyz, &idx, &cnt, dims, wpack, bias, out, out_stride,
                                  &out_off, &hdr_c, &gran_c, &max_tiles, &scratch, overflow, stream);
    }
    if (fp16 && !(flags & 8) && ((flags & 32) || (long)b * m * ns >= 4096)) {   // the widest scales on 96-row items (mlp_wide128.hip)
        int st = SA_OK;
        if (sa_wide128_try(b, n, m, ns, c, xyz, feat, new_xyz, idx, cnt, nl, dims, wpack, bias, out, out_stride, out_off,
                           hdr, gran, max_tiles, 1, (flags & 32) ? 1 : 0, 0, overflow, stream, &st))
            return st;
    }
    {
        int st = SA_OK;
        if (sa_rowwave_try(b, n, m, ns, c, xyz, feat, new_xyz, idx, cnt, nl, dims, wpack, bias, out, out_stride,
                           out_off, hdr, gran, max_tiles, fp16 ? 1 : 0, 0, 0, overflow, stream, &st))
            return st;
    }
    const int abytes = fp16 ? 2 : 4;               // LDS bytes per activation channel (one fp16 plane / hi + lo bf16)
    MlpParams P{};
    P.xyz = xyz; P.feat = feat; P.new_xyz = new_xyz; P.idx = idx; P.cnt = cnt; P.out = out;
    P.n = n; P.m = m; P.ns = ns; P.C = c; P.nballs = nballs;
    P.out_stride = out_stride; P.out_off = out_off; P.nl = nl;
    P.hdr = hdr; P.gran = gran; P.ovf = overflow;
    int wA = roundup(dims[0], 16), wB = 0;
    for (int l = 0; l < nl; ++l) {
        P.L[l].w = (const uint4 *)wpack[l];
        P.L[l].bias = bias[l];
        P.L[l].K = dims[l];
        P.L[l].N = dims[l + 1];
        P.L[l].KS = roundup(dims[l], 16) / 16;
        P.L[l].NT = roundup(dims[l + 1], 32) / 32;
        if (l + 1 < nl) {                      // output of layer l is stored: l even -> B, l odd -> A
            const int wd = P.L[l].NT * 32;
            if (l & 1) { if (wd > wA) wA = wd; } else { if (wd > wB) wB = wd; }
        }
    }
    P.strideA = wA * abytes + 16;
    P.strideB = wB * abytes + 16;
    P.lds_bytes = kRows * (P.strideA + P.strideB);
    const size_t lds = (size_t)P.lds_bytes;
    if (lds > 160 * 1024) return SA_ERR_UNSUPPORTED;
    if (lds > 48 * 1024) {   // opt in to large dynamic LDS; a refusal surfaces at the launch check below
        (void)hipFuncSetAttribute(fp16 ? (const void *)group_mlp_max_kernel<kNW, 1> : (const void *)group_mlp_max_kernel<kNW, 3>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipGetLastError();
    }
    const long nitems = max_tiles;             // the densest plan; workgroups past the planned tiles leave at once
    int max_nt = 0;
    for (int l = 0; l < nl; ++l) if (P.L[l].NT > max_nt) max_nt = P.L[l].NT;
    static const int narrow_nt = SA_KNOB("SA_MLP_NARROW_NT", 4);  // tuning knob
    const bool narrow = max_nt <= narrow_nt && lds <= 40 * 1024;   // default: <= 128 output channels everywhere
    if (narrow) {
        const int grid = (int)(nitems < 65536 ? nitems : 65536);
        if (fp16) hipLaunchKernelGGL((group_mlp_max_kernel<1, 1>), dim3(grid), dim3(64), lds, stream, P);
        else hipLaunchKernelGGL((group_mlp_max_kernel<1, 3>), dim3(grid), dim3(64), lds, stream, P);
        SA_CHECK_LAUNCH();
        return SA_OK;
    }
    // ---- wide path: 64-row items, last hidden layer chunked until the buffers fit
    // 64-row items halve the L2 weight traffic; measured faster only where that traffic is the bound -- the
    // 1024-channel last layer of layer4 (0.344 -> 0.307 ms); SA_MLP_WIDE=1/0 forces it on / off for every shape
    static const int wide_env = SA_KNOB("SA_MLP_WIDE", -1);
    const bool use_wide = wide_env >= 0 ? wide_env != 0 : P.L[nl - 1].NT >= 32;
    if (use_wide && P.L[nl - 1].NT <= 4 * kNW) {
        WideParams WP{};
        WP.M = P;
        const int nt_h = nl >= 2 ? P.L[nl - 2].NT : 0;            // tiles of the last hidden layer
        for (int nch = 1; nch <= (nt_h > 0 ? nt_h : 1); nch *= 2) {
            const int tpc = nt_h > 0 ? (nt_h + nch - 1) / nch : 0;
            int wwA = roundup(dims[0], 16), wwB = 0;
            for (int l = 0; l + 1 < nl; ++l) {                    // act_{l+1}: l even -> B, l odd -> A
                const int wd = (l == nl - 2 ? tpc : P.L[l].NT) * 32;
                if (l & 1) { if (wd > wwA) wwA = wd; } else { if (wd > wwB) wwB = wd; }
            }
            WP.M.strideA = wwA * abytes + 16;
            WP.M.strideB = wwB * abytes + 16;
            WP.M.lds_bytes = kWRows * (WP.M.strideA + WP.M.strideB);
            const size_t wlds = (size_t)WP.M.lds_bytes;
            if (wlds <= 156 * 1024) {
                WP.tiles_per_chunk = tpc;
                WP.nchunks = nt_h > 0 ? (nt_h + tpc - 1) / tpc : 1;
                (void)hipFuncSetAttribute(fp16 ? (const void *)group_mlp_wide_kernel<1> : (const void *)group_mlp_wide_kernel<3>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds);
                (void)hipGetLastError();
                const long witems = (max_tiles + 1) / 2;
                const int grid = (int)(witems < 16384 ? witems : 16384);
                if (fp16) hipLaunchKernelGGL(group_mlp_wide_kernel<1>, dim3(grid), dim3(kThreads), wlds, stream, WP);
                else hipLaunchKernelGGL(group_mlp_wide_kernel<3>, dim3(grid), dim3(kThreads), wlds, stream, WP);
                SA_CHECK_LAUNCH();
                return SA_OK;
            }
            if (nt_h == 0) break;
        }
    }
    {
        const int grid = (int)(nitems < 16384 ? nitems : 16384);
        if (fp16) hipLaunchKernelGGL((group_mlp_max_kernel<kNW, 1>), dim3(grid), dim3(kThreads), lds, stream, P);
        else hipLaunchKernelGGL((group_mlp_max_kernel<kNW, 3>), dim3(grid), dim3(kThreads), lds, stream, P);
    }
    SA_CHECK_LAUNCH();
    return SA_OK;
}

int sa_rowwave_try_layer(int b, int n, int m, const int *ns, int c, const float *xyz, const float *feat,
                         const float *new_xyz, const int *const *idx, const int *const *cnt, const int *dims,
                         const void *const *wpack, const float *const *bias, float *out, int out_stride,
                         const int *out_off, const int *const *plan_hdr, const int *const *plan_gran,
                         const long *max_tiles, const int *fp16, const int *gr, int *overflow, hipStream_t stream, int *st);

// All scales of one SA layer (layers_util.py:134-181): scale i has nsample ns[i], index / count tensors idx[i] /
// cnt[i], layer widths dims[i*(nl+1) ..], weights wpack[i*nl ..] / bias[i*nl ..], output slice out_off[i], plan
// scratch ws[i] and flags[i] (as sa_group_mlp_max).  The three-scale layers of 3dssd.yaml whose plans were built by
// sa_group_mlp_plan run as ONE launch (mlp_rowwave.hip, mlp_multi_kernel); anything else is the per-scale loop.
extern "C" int sa_group_mlp_max_layer(int nscale, int b, int n, int m, const int *ns, int c, const float *xyz,
                                      const float *feat, const float *new_xyz, const int *const *idx,
                                      const int *const *cnt, int nl, const int *dims, const void *const *wpack,
                                      const float *const *bias, float *out, int out_stride, const int *out_off,
                                      void *const *ws, const size_t *ws_bytes, const int *flags, int *overflow,
                                      hipStream_t stream) {
    if (nscale < 1 || !ns || !idx || !cnt || !dims || !wpack || !bias || !out_off || !ws || !ws_bytes || !flags)
        return SA_ERR_INVALID;
    if (nscale == 3 && nl == 3 && b > 0 && m > 0 && c >= 0 && xyz && new_xyz && out) {
        bool ok = true;
        const int *hdr[3], *gran[3];
        long max_tiles[3];
        int fp16[3];
        for (int i = 0; i < 3 && ok; ++i) {
            ok = (flags[i] & 2) && !(flags[i] & 1) && ns[i] > 0 && idx[i] && cnt[i] && ws[i] &&
                 ws_bytes[i] >= sa_group_mlp_max_ws_bytes(b, m, ns[i]) && dims[4 * i] == c + 3 && (c == 0 || feat);
            for (int l = 0; l < 3 && ok; ++l) ok = dims[4 * i + l + 1] > 0 && wpack[3 * i + l] && bias[3 * i + l];
            if (!ok) break;
            hdr[i] = (const int *)ws[i];
            gran[i] = hdr[i] + sa::kPlanHeaderInts;
            max_tiles[i] = (sa_plan_max_granules((long)b * m, ns[i]) + 3) / 4;
            fp16[i] = (flags[i] & 4) ? 1 : 0;
        }
        if (ok) {
            // granule size per scale (bit 6 of a scale's flags = 4 rows, bit 8 = 2 rows): the instantiated combinations of
            // mlp_rowwave.hip run as ONE launch, any other falls through to the scale-by-scale calls (8 / 4 rows only)
            int gr[3];
            for (int i = 0; i < 3; ++i) gr[i] = (flags[i] & 256) ? 2 : (flags[i] & 64) ? 4 : 8;
            int st = SA_OK;
            if (sa_rowwave_try_layer(b, n, m, ns, c, xyz, feat, new_xyz, idx, cnt, dims, wpack, bias, out, out_stride,
                                     out_off, hdr, gran, max_tiles, fp16, gr, overflow, stream, &st))
                return st;
            if (gr[0] == 2 || gr[1] == 2 || gr[2] == 2) return SA_ERR_UNSUPPORTED;    // 2-row plans: the one-launch layer kernels only
        }
    }
    // two wide scales (layer4 of 3dssd.yaml) whose plans are built: the GEMM chain of both in three launches
    if (nscale == 2 && nl == 3 && b > 0 && m > 0 && xyz && new_xyz && out && feat) {
        bool ok = true;
        const int *hdr[2], *gran[2];
        long max_tiles[2];
        void *scratch[2];
        for (int i = 0; i < 2 && ok; ++i) {
            ok = (flags[i] & 2) && ns[i] > 0 && idx[i] && cnt[i] && ws[i] && dims[4 * i] == c + 3 &&
                 use_gemm_chain(b, m, ns[i], c, nl, dims + 4 * i, ws_bytes[i], flags[i]);
            for (int l = 0; l < 3 && ok; ++l) ok = dims[4 * i + l + 1] > 0 && wpack[3 * i + l] && bias[3 * i + l];
            if (!ok) break;
            hdr[i] = (const int *)ws[i];
            gran[i] = hdr[i] + sa::kPlanHeaderInts;
            max_tiles[i] = (sa_plan_max_granules((long)b * m, ns[i]) + 3) / 4;
            scratch[i] = (char *)ws[i] + plan_bytes_aligned(b, m, ns[i]);
        }
        if (ok)
            return sa_mlp_gemm_launch(2, b, n, m, ns, c, xyz, feat, new_xyz, idx, cnt, dims, wpack, bias, out, out_stride,
                                      out_off, hdr, gran, max_tiles, scratch, overflow, stream);
    }
    for (int i = 0; i < nscale; ++i) {
        const int st = sa_group_mlp_max(b, n, m, ns[i], c, xyz, feat, new_xyz, idx[i], cnt[i], nl, dims + (nl + 1) * i,
                                        wpack + nl * i, bias + nl * i, out, out_stride, out_off[i], ws[i], ws_bytes[i],
                                        flags[i], overflow, stream);
        if (st != SA_OK) return st;
    }
    return SA_OK;
}

// y[rows,N] = act(x[rows,K] W + b): tf_util.conv1d 1x1 with folded BN (tf_util.py:51-124).
extern "C" int sa_dense(long rows, int K, int N, const float *x, const void *wpack, const float *bias,
                        int relu, float *y, hipStream_t stream) {
    if (rows <= 0 || K <= 0 || N <= 0 || !x || !wpack || !bias || !y) return SA_ERR_INVALID;
    DenseParams P{};
    P.x = x; P.y = y; P.rows = rows; P.relu = relu;
    P.L.w = (const uint4 *)wpack; P.L.bias = bias; P.L.K = K; P.L.N = N;
    P.L.KS = roundup(K, 16) / 16;
    P.L.NT = roundup(N, 32) / 32;
    // layers with enough rows to give every CU a 128-row block: the 128-row kernel (same bits).  128-column blocks with
    // three chunks of lookahead where the shape allows, 64-column blocks with two (four waves) for the narrow 128 -> 64 layer
    if (rows < (1l << 31) && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {      // 16-byte row pieces in and out
        const long blocks = (rows + 127) / 128;
        // 256-column blocks (round 6; a wave = one column tile x FOUR row tiles, two chunks of lookahead: 212 registers): a
        // weight fragment is fetched once per 128 rows instead of twice and the rows are converted to split bf16 for half
        // as many column blocks -- 65536 x 768 -> 256 126 -> 95 us, 32768 x 1536 -> 512 204 -> 154 us (128 frames; three
        // chunks of lookahead: 100 / 160), bit-identical
        static const int wide_knob = SA_KNOB("SA_D128_WIDE", 1);
        if (wide_knob && K % (kD128KC * 2) == 0 && N % 256 == 0 && blocks * (N / 256) >= 192) {
            auto kern = dense128_kernel<8, 8, 2>;
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kD128Lds);
            (void)hipGetLastError();
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks, N / 256), dim3(512), kD128Lds, stream, P);
            SA_CHECK_LAUNCH();
            return SA_OK;
        }
        if (K % (kD128KC * 3) == 0 && N % 128 == 0 && blocks * (N / 128) >= 192) {
            auto kern = dense128_kernel<8, 4, 3>;
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kD128Lds);
            (void)hipGetLastError();
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks, N / 128), dim3(512), kD128Lds, stream, P);
            SA_CHECK_LAUNCH();
            return SA_OK;
        }
        if (K % (kD128KC * 2) == 0 && N % 64 == 0 && N % 128 != 0 && blocks * (N / 64) >= 512) {
            // FOUR waves (a wave = one column tile x two row tiles) since round 6: the eight-wave form needs 200 registers per
            // lane, i.e. ONE workgroup per CU, and with two chunks per block (K = 128) nothing overlaps one block's loads with
            // another's matrix work and stores; two independent four-wave workgroups per CU do: 140 -> 109 us at 524288 x 128
            // -> 64 (128 frames), bit-identical.  (Wider layers: the 64-column form reads and converts the rows once per 64
            // columns -- 131072 x 384 -> 128 100 -> 104 us with it: they keep the eight-wave forms above.)
            auto kern = dense128_kernel<4, 2, 2>;
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kD128Lds);
            (void)hipGetLastError();
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks, N / 64), dim3(256), kD128Lds, stream, P);
            SA_CHECK_LAUNCH();
            return SA_OK;
        }
    }
    P.KC = P.L.KS * 16 < 256 ? P.L.KS * 16 : 256;
    P.stride = P.KC * 4 + 16;
    const size_t lds = (size_t)kRows * P.stride;
    const long tiles = (rows + kRows - 1) / kRows;
    // tiles per wave: as few as keeps >= ~512 workgroups in flight, at most 4
    P.tg = 4;
    while (P.tg > 1 && tiles * ((P.L.NT + kDW * P.tg - 1) / (kDW * P.tg)) < 512) P.tg >>= 1;
    while (P.tg > 1 && kDW * P.tg / 2 >= P.L.NT) P.tg >>= 1;     // no more tile slots per workgroup than the layer has tiles
    const int ysplit = (P.L.NT + kDW * P.tg - 1) / (kDW * P.tg);
    const int gx = (int)(tiles < 8192 ? tiles : 8192);
    if (P.tg == 1) hipLaunchKernelGGL(dense_kernel<1>, dim3(gx, ysplit), dim3(kDThreads), lds, stream, P);
    else if (P.tg == 2) hipLaunchKernelGGL(dense_kernel<2>, dim3(gx, ysplit), dim3(kDThreads), lds, stream, P);
    else hipLaunchKernelGGL(dense_kernel<4>, dim3(gx, ysplit), dim3(kDThreads), lds, stream, P);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// vote_layer tail: hidden = relu(x W1 + b1) [rows,H], offsets = hidden W2 + b2 [rows,3], out = xyz + clip(offsets, lo, -lo)
// in one launch (vote_tail_kernel).  H <= 128 (one workgroup holds a 32-row hidden tile): SA_ERR_UNSUPPORTED otherwise,
// the caller then uses sa_dense twice and sa_vote_translate.  Same bits as that three-launch form.
extern "C" int sa_vote_tail(long rows, int K, int H, const float *x, const void *w1pack, const float *bias1,
                            const void *w2pack, const float *bias2, float *hidden, float *offsets, const float *xyz,
                            float lo_x, float lo_y, float lo_z, float *out, hipStream_t stream) {
    if (rows <= 0 || K <= 0 || H <= 0 || !x || !w1pack || !bias1 || !w2pack || !bias2 || !hidden || !offsets || !xyz || !out)
        return SA_ERR_INVALID;
    if (H > kDW * 32) return SA_ERR_UNSUPPORTED;
    VoteTailParams V{};
    DenseParams &P = V.D;
    P.x = x; P.y = hidden; P.rows = rows; P.relu = 1;
    P.L.w = (const uint4 *)w1pack; P.L.bias = bias1; P.L.K = K; P.L.N = H;
    P.L.KS = roundup(K, 16) / 16;
    P.L.NT = roundup(H, 32) / 32;
    P.KC = P.L.KS * 16 < 256 ? P.L.KS * 16 : 256;
    P.stride = P.KC * 4 + 16;
    P.tg = 1;
    V.L2.w = (const uint4 *)w2pack; V.L2.bias = bias2; V.L2.K = H; V.L2.N = 3;
    V.L2.KS = roundup(H, 16) / 16;
    V.L2.NT = 1;
    V.offsets = offsets; V.xyz = xyz; V.out = out;
    V.lo[0] = lo_x; V.lo[1] = lo_y; V.lo[2] = lo_z;
    const size_t lds1 = (size_t)kRows * P.stride, lds2 = (size_t)kRows * (P.L.NT * 32 * 4 + 16);
    const long tiles = (rows + kRows - 1) / kRows;
    hipLaunchKernelGGL(vote_tail_kernel, dim3((unsigned)(tiles < 8192 ? tiles : 8192)), dim3(kDThreads), lds1 > lds2 ? lds1 : lds2,
                       stream, V);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

extern "C" int sa_vote_translate(long npoints, const float *xyz, const float *off, float lo_x, float lo_y,
                                 float lo_z, float *out, hipStream_t stream) {
    if (npoints <= 0 || !xyz || !off || !out) return SA_ERR_INVALID;
    const long total = npoints * 3;
    const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(vote_translate_kernel, dim3(grid), dim3(256), 0, stream, total, xyz, off, lo_x, lo_y,
                       lo_z, out);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

#ifdef SA_MLP_TIMING
extern "C" int sa_debug_mlp_prof(unsigned long long *host8, int reset) {
    static unsigned long long *h = (unsigned long long *)calloc(65536 * 8, sizeof(unsigned long long));
    if (host8) {
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mlp_prof), 65536 * 8 * sizeof(unsigned long long)) != hipSuccess) return SA_ERR_LAUNCH;
        for (int i = 0; i < 8; ++i) host8[i] = 0;
        for (int r = 0; r < 65536; ++r) for (int i = 0; i < 8; ++i) host8[i] += h[r * 8 + i];
    }
    if (reset) {
        void *d = nullptr;
        if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_mlp_prof)) != hipSuccess) return SA_ERR_LAUNCH;
        if (hipMemset(d, 0, 65536 * 8 * sizeof(unsigned long long)) != hipSuccess) return SA_ERR_LAUNCH;
    }
    return SA_OK;
}
#endif
