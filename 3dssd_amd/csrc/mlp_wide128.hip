// Fused grouped MLP of the widest scales (layer4 of 3dssd.yaml: 259 -> 256 -> 256|512 -> 512|1024) on 96-ROW items.
//
// group_mlp_wide_kernel (mlp.hip) keeps a 64-row item in LDS from gather to pooled output and every wave streams the
// weight fragments of its own output tiles from L2: 1.42 MB per item at 512 -> 1024, 64 flop per weight byte.  It is
// bound by bytes in flight: the L2s answer a loaded request after ~1.7 us, a wave has registers for 8 fragments of
// lookahead, so a CU draws ~40 GB/s of weights whatever the matrix pipe could take (SQ_VALU_MFMA_BUSY 0.26).  Here an
// item is 96 rows -- the most whose activations fit the CU's 160 KiB of LDS in fp16 (second hidden layer 96 x 512 =
// 100 KB, which first holds the gathered 96 x 272 input, + first hidden layer 96 x 256 = 51 KB) -- so a weight fragment
// feeds THREE 32-row MFMA tiles instead of two: a third fewer weight bytes per row for the same bytes in flight.  The
// last layer's accumulators (4 column tiles x 3 row tiles per wave = 192 registers) do not fit beside the fragments, so
// it runs in passes of 16 column tiles (2 per wave x 3 row tiles = 96 registers) over the LDS-resident hidden layer:
// nothing is recomputed, only the LDS operand reads repeat.  (A 128-row form was tried first: its second hidden layer
// does not fit LDS and had to be produced in 256-column chunks inside every pass while the pass's 128 accumulator
// registers were live -- 31 registers spilled.)  Same arithmetic as the other fp16 kernels: one v_mfma_f32_32x32x16_f16
// pass per k-step, fp32 accumulate, k ascending, bias in the accumulator of the hidden layers (D^T form) and added at
// the pooled write of the last (D form) -- bit-identical to group_mlp_wide_kernel / mlp_rs_kernel, and range-guarded
// (mlp_act.h).
#include "sa_common.h"
#include "mlp_plan.h"
#include "mlp_act.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector for the weight ring (plain loads / moves)

constexpr int kNW = 8, kThreads = kNW * 64;
constexpr int kRows = 96, kRT = kRows / 32;         // row tiles of an item
constexpr int kGB = 16;                              // LDS bytes of an 8-channel group (one fp16 plane)
// weight fragments (k-steps) in flight per wave.  The L2s answer after ~1.7 us under this load whatever is asked, so a
// wave's weight rate is (fragments in flight) x 1 KiB / 1.7 us: depth is what the spare registers are spent on.
#ifndef SA_W96_DH
#define SA_W96_DH 8
#endif
#ifndef SA_W96_D1
#define SA_W96_D1 6      // first layer: 17 k-steps (259 + padding) = 3 blocks of 6, one step idle
#endif
#ifndef SA_W96_DL
#define SA_W96_DL 8
#endif
constexpr int kDepthFirst = SA_W96_D1, kDepthHidden = SA_W96_DH, kDepthLast = SA_W96_DL;

struct W128Layer {
    const uint4 *w;      // packed fp16 fragments [NT][KS][64]
    const float *bias;   // [NT * 32]
    int KS, NT, N;
};
struct W128Params {
    const float *xyz, *feat, *new_xyz;
    const int *idx, *cnt;
    float *out;
    int n, m, ns, C;
    int out_stride, out_off;
    const int *gran, *hdr;
    W128Layer L[3];
    int strideA, strideB;   // LDS row strides in bytes: buffer A (gathered input, then hidden 2), buffer B (hidden 1)
    int *ovf;
};

#ifdef SA_W96_PROF
// debug build only (tools/w96_prof.py): cycles of wave 0 per item: [0] gather + plan entries, [1] hidden 1, [2] hidden 2,
// [3] last layer MFMA loops, [4] pooling / write-out, [5] items
__device__ unsigned long long g_w96_prof[8];
#define WP_T0() unsigned long long wp_t = __builtin_readcyclecounter(), wp_acc[6] = {0, 0, 0, 0, 0, 0}
#define WP_TICK(i) { const unsigned long long n__ = __builtin_readcyclecounter(); wp_acc[i] += n__ - wp_t; wp_t = n__; }
#define WP_FLUSH() if (tid == 0) { for (int i__ = 0; i__ < 6; ++i__) atomicAdd(&g_w96_prof[i__], wp_acc[i__]); }
#else
#define WP_T0()
#define WP_TICK(i)
#define WP_FLUSH()
#endif

__device__ __forceinline__ f32x16 mfma_f16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// (ball, flat source point) of the rows lane and lane + 64 of an item: plan entry -> index / count -> point, two
// dependent global round trips.  Resolved one item AHEAD (during the previous item's matrix work).
struct RowRefs { int pt[2], ball[2]; };
__device__ __forceinline__ RowRefs resolve_rows(const W128Params &P, int item, int ngran, int lane) {
    RowRefs R;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = lane + 64 * h < kRows ? lane + 64 * h : kRows - 1;
        const int ent = sa::plan_entry(P.gran, ngran, item * (kRows / 8) + (row >> 3));
        const int ball = ent >= 0 ? sa::plan_ball(ent) : 0;
        const int s = sa::plan_sample(ent, row & 7, P.ns);
        const int a_raw = P.idx[(size_t)ball * P.ns + s];
        const int a = P.cnt[ball] > 0 ? a_raw : 0;          // layers_util.py:157-159
        R.ball[h] = ball;
        R.pt[h] = (ball / P.m) * P.n + a;
    }
    return R;
}

// gather the item's 96 rows x [features, xyz - centre, 0 padding] into `buf` as fp16 (row-major, 16 bytes per
// 8-channel group) from the rows resolved by resolve_rows (every wave holds all of them: lane -> rows lane, lane + 64).
__device__ __forceinline__ void gather128(const W128Params &P, unsigned char *buf, const RowRefs &R, int tid,
                                          sa::f16_guard_t &det) {
    const int r_pt[2] = {R.pt[0], R.pt[1]}, r_ball[2] = {R.ball[0], R.ball[1]};
    const int GF = P.C >> 3;                                // full feature groups per row (C % 8 == 0)
    const int G0 = P.L[0].KS * 2;                           // groups of the padded input row
    const int totF = kRows * GF;
#pragma unroll 8
    for (int it0 = 0; it0 < totF; it0 += kThreads) {
        const bool live = it0 + tid < totF;
        const int it = live ? it0 + tid : totF - 1;
        const int row = it / GF, g = it - row * GF;
        const int p0 = __shfl(r_pt[0], row & 63), p1 = __shfl(r_pt[1], row & 63);
        const long pt = row < 64 ? p0 : p1;
        const float4 f0 = *(const float4 *)(P.feat + pt * P.C + g * 8);
        const float4 f1 = *(const float4 *)(P.feat + pt * P.C + g * 8 + 4);
        const uint4 r = make_uint4(sa::cvt2_f16(f0.x, f0.y), sa::cvt2_f16(f0.z, f0.w), sa::cvt2_f16(f1.x, f1.y), sa::cvt2_f16(f1.z, f1.w));
        sa::f16_guard_signed(r, det);
        if (live) *(uint4 *)(buf + row * P.strideA + g * kGB) = r;
    }
    const int GT = G0 - GF;                                 // tail groups: [dx, dy, dz, 0 ...], then zero groups
    const int totT = kRows * GT;
    for (int it0 = 0; it0 < totT; it0 += kThreads) {
        const bool live = it0 + tid < totT;
        const int it = live ? it0 + tid : totT - 1;
        const int row = it / GT, g = GF + (it - row * GT);
        const int p0 = __shfl(r_pt[0], row & 63), p1 = __shfl(r_pt[1], row & 63);
        const int b0 = __shfl(r_ball[0], row & 63), b1 = __shfl(r_ball[1], row & 63);
        const long pt = row < 64 ? p0 : p1, ball = row < 64 ? b0 : b1;
        const float px = P.xyz[pt * 3 + 0] - P.new_xyz[ball * 3 + 0];
        const float py = P.xyz[pt * 3 + 1] - P.new_xyz[ball * 3 + 1];
        const float pz = P.xyz[pt * 3 + 2] - P.new_xyz[ball * 3 + 2];
        const bool first = g == GF;
        const uint4 r = make_uint4(sa::cvt2_f16(first ? px : 0.0f, first ? py : 0.0f), sa::cvt2_f16(first ? pz : 0.0f, 0.0f), 0u, 0u);
        sa::f16_guard_signed(r, det);
        if (live) *(uint4 *)(buf + row * P.strideA + g * kGB) = r;
    }
}

// One 32-column tile `ct` of a hidden layer for the item's four row tiles (D^T form: weights are the MFMA A operand):
// out[row][ocol0*32 .. +32) = relu(bias + in[row][:] W[:, ct]).  The weight fragments of DEPTH k-steps are in flight.
template <int DEPTH>
__device__ __forceinline__ void hidden_tile(const unsigned char *in, int strideIn, unsigned char *outb, int strideOut,
                                            const W128Layer &L, int ct, int ocol0, int lane, sa::f16_guard_t &det) {
    const int half = lane >> 5, col = lane & 31;
    const unsigned char *arow = in + col * strideIn + half * kGB;
    f32x16 acc[kRT];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 bv = *(const float4 *)(L.bias + ct * 32 + 8 * q + 4 * half);
#pragma unroll
        for (int r = 0; r < kRT; ++r) {
            acc[r][4 * q + 0] = bv.x; acc[r][4 * q + 1] = bv.y; acc[r][4 * q + 2] = bv.z; acc[r][4 * q + 3] = bv.w;
        }
    }
    // The k loop runs in straight-line blocks of DEPTH k-steps with NO branch inside: every step consumes ring slot d and
    // refills it with the fragment DEPTH steps ahead, so DEPTH loads stay in flight across the whole loop.  (With an
    // `if (ks < KS)` around each step the blocks end after every step and hipcc waits for the refill -- `s_waitcnt
    // vmcnt(0)` -- before leaving the block: one L2 round trip per k-step.)  Steps past the end use a zero fragment.
    const u32x4 *wb = (const u32x4 *)L.w + (size_t)ct * L.KS * 64 + lane;
    const int KS = L.KS;
    u32x4 wq[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) wq[d] = wb[(d < KS ? d : KS - 1) * 64];
    // The ring is loop-carried: slot d enters the loop from the prologue load above and from the refill below.  InstCombine
    // folds such a PHI of two loads into ONE load of a PHI of the addresses, placed at the top of the iteration that uses
    // it -- the software pipeline silently becomes "load, wait, use" (seen in the ISA).  Passing the prologue values
    // through an empty asm makes the PHI's inputs differ in kind and keeps the refills where they are written.
#ifndef SA_W96_NOLAUNDER
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) asm volatile("" : "+v"(wq[d]));
#endif
    for (int ks0 = 0; ks0 < KS; ks0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int ks = ks0 + d;
            const bool live = ks < KS;
            const int ksc = live ? ks : KS - 1;
            u32x4 wv = wq[d];
            if (!live) wv = u32x4{0u, 0u, 0u, 0u};
            const uint4 wf = __builtin_bit_cast(uint4, wv);
            const int kw = ks + DEPTH < KS ? ks + DEPTH : KS - 1;
            wq[d] = wb[kw * 64];
            uint4 af[kRT];
#pragma unroll
            for (int r = 0; r < kRT; ++r) af[r] = *(const uint4 *)(arow + r * 32 * strideIn + ksc * 2 * kGB);
#pragma unroll
            for (int r = 0; r < kRT; ++r) acc[r] = mfma_f16(wf, af[r], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < kRT; ++r) {
        unsigned char *orow = outb + (r * 32 + col) * strideOut + ocol0 * 4 * kGB + 8 * half;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint2 pk = make_uint2(sa::cvt2_f16_relu(acc[r][4 * q], acc[r][4 * q + 1]), sa::cvt2_f16_relu(acc[r][4 * q + 2], acc[r][4 * q + 3]));
            sa::f16_guard(pk, det);
            *(uint2 *)(orow + q * kGB) = pk;
        }
    }
}

// Last layer (D form), column tiles ct0, ct0 + 1 of this wave, k-steps [ks_lo, ks_hi) of the contraction whose operand
// columns start at k-step ks_lo in `in`; accumulators persist across the calls of a pass.
template <int DEPTH>
__device__ __forceinline__ void last_partial(f32x16 (&acc)[2][kRT], const unsigned char *in, int strideIn,
                                             const W128Layer &L, int ct0, int ks_lo, int ks_hi, int lane) {
    const int half = lane >> 5, col = lane & 31;
    const unsigned char *arow = in + col * strideIn + half * kGB;
    const u32x4 *wb0 = (const u32x4 *)L.w + (size_t)(ct0 < L.NT ? ct0 : L.NT - 1) * L.KS * 64 + lane;
    const u32x4 *wb1 = (const u32x4 *)L.w + (size_t)(ct0 + 1 < L.NT ? ct0 + 1 : L.NT - 1) * L.KS * 64 + lane;
    u32x4 wq[DEPTH][2];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const int kd = ks_lo + d < ks_hi ? ks_lo + d : ks_hi - 1;
        wq[d][0] = wb0[kd * 64];
        wq[d][1] = wb1[kd * 64];
    }
#ifndef SA_W96_NOLAUNDER
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) asm volatile("" : "+v"(wq[d][0]), "+v"(wq[d][1]));   // see hidden_tile
#endif
    for (int ks0 = ks_lo; ks0 < ks_hi; ks0 += DEPTH) {       // branch-free blocks of DEPTH k-steps (see hidden_tile)
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int ks = ks0 + d;
            const bool live = ks < ks_hi;
            const int ksc = live ? ks : ks_hi - 1;
            u32x4 v0 = wq[d][0], v1 = wq[d][1];
            if (!live) { v0 = u32x4{0u, 0u, 0u, 0u}; v1 = v0; }
            const uint4 w0 = __builtin_bit_cast(uint4, v0), w1 = __builtin_bit_cast(uint4, v1);
            const int kw = ks + DEPTH < ks_hi ? ks + DEPTH : ks_hi - 1;
            wq[d][0] = wb0[kw * 64];
            wq[d][1] = wb1[kw * 64];
            uint4 af[kRT];
#pragma unroll
            for (int r = 0; r < kRT; ++r) af[r] = *(const uint4 *)(arow + r * 32 * strideIn + (ksc - ks_lo) * 2 * kGB);
#pragma unroll
            for (int r = 0; r < kRT; ++r) {
                acc[0][r] = mfma_f16(af[r], w0, acc[0][r]);
                acc[1][r] = mfma_f16(af[r], w1, acc[1][r]);
            }
        }
    }
}

__global__ __launch_bounds__(kThreads, 2) void group_mlp_wide128_kernel(W128Params P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *bufA = smem;                               // gathered input, later the hidden-2 chunk
    unsigned char *bufB = smem + kRows * P.strideA;           // hidden 1
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6) & (kNW - 1);
    const int ngran = __builtin_amdgcn_readfirstlane(P.hdr[0]);
    const int nitems = (ngran + kRows / 8 - 1) / (kRows / 8);      // 96-row items = 12 granules of the plan
    const W128Layer &L1 = P.L[0], &L2 = P.L[1], &L3 = P.L[2];
    const int npass = (L3.NT + 2 * kNW - 1) / (2 * kNW);      // passes of 16 column tiles over the last layer
    sa::f16_guard_t det = 0;

    int istride;
    const int item0 = sa::xcd_block(blockIdx.x, gridDim.x, nitems, istride);
    if (item0 < 0) return;
    WP_T0();
    RowRefs refs = resolve_rows(P, item0, ngran, lane);
    int gr_ent = sa::plan_entry(P.gran, ngran, item0 * (kRows / 8) + (lane < kRT * 4 ? lane : 0));
    int gr_cnt = gr_ent >= 0 ? P.cnt[sa::plan_ball(gr_ent)] : 0;
    for (int item = item0; item < nitems; item += istride) {
        gather128(P, bufA, refs, tid, det);
        // the plan entries and ball counts of the item's 12 granules (lane g holds granule g; resolved one item ahead)
        int ent[kRT][4], cn[kRT][4];
#pragma unroll
        for (int g = 0; g < kRT * 4; ++g) {
            ent[g >> 2][g & 3] = __builtin_amdgcn_readlane(gr_ent, g);
            cn[g >> 2][g & 3] = __builtin_amdgcn_readlane(gr_cnt, g);
        }
        // the next item's rows and granules: requested now, needed at the top of the next iteration
        {
            const int nxt = item + istride < nitems ? item + istride : item;
            refs = resolve_rows(P, nxt, ngran, lane);
            gr_ent = sa::plan_entry(P.gran, ngran, nxt * (kRows / 8) + (lane < kRT * 4 ? lane : 0));
            gr_cnt = gr_ent >= 0 ? P.cnt[sa::plan_ball(gr_ent)] : 0;
        }
        __syncthreads();
        WP_TICK(0)
        // ---- hidden 1: 8 column tiles, one per wave: bufA -> bufB
        hidden_tile<kDepthFirst>(bufA, P.strideA, bufB, P.strideB, L1, w, w, lane, det);
        __syncthreads();
        WP_TICK(1)
        // ---- hidden 2: column tiles w, w + 8, ...: bufB -> bufA (the gathered input is dead)
        for (int ct = w; ct < L2.NT; ct += kNW) hidden_tile<kDepthHidden>(bufB, P.strideB, bufA, P.strideA, L2, ct, ct, lane, det);
        __syncthreads();
        WP_TICK(2)
        // ---- last layer in passes of 16 column tiles over the LDS-resident hidden 2; pooled write per pass
        for (int p = 0; p < npass; ++p) {
            const int ct0 = p * 2 * kNW + 2 * w;              // this wave's two column tiles of the pass
            // their bias: requested HERE, used after the matrix loop (a load in front of the pooled write was waited
            // for -- a full memory round trip per tile)
            float bias3[2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) bias3[t2] = L3.bias[(ct0 + t2 < L3.NT ? ct0 + t2 : L3.NT - 1) * 32 + (lane & 31)];
            f32x16 acc[2][kRT];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < kRT; ++r)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t2][r][e] = 0.0f;
            last_partial<kDepthLast>(acc, bufA, P.strideA, L3, ct0, 0, L3.KS, lane);
            WP_TICK(3)
            const int col = lane & 31;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int ct = ct0 + t2;
                if (ct < L3.NT) {
                    const int ch = ct * 32 + col;
                    const float bc = bias3[t2];
#pragma unroll
                    for (int r = 0; r < kRT; ++r) {
                        float qm[4];
                        sa::granule_max(acc[t2][r], qm);
                        sa::pool_write_tile(qm, ent[r], cn[r], bc, ch, L3.N, P.out, P.out_stride, P.out_off, lane);
                    }
                }
            }
        }
        __syncthreads();                                      // the next item's gather overwrites bufA
#ifdef SA_W96_PROF
        WP_TICK(4)
        wp_acc[5] += 1;
#endif
    }
    WP_FLUSH()
    sa::f16_overflow_report(det, P.ovf, lane);
}

int r128_roundup(int x, int q) { return (x + q - 1) / q * q; }

}  // namespace

// Returns 1 when the scale has the shape this kernel takes and the launch was issued (status in *st): three fp16
// layers, c a multiple of 8, first hidden width 256 (one column tile per wave), second hidden width a multiple of 32 up
// to 512 (it must fit LDS beside the first), last width of at least 16 column tiles.
int sa_wide128_try(int b, int n, int m, int ns, int c, const float *xyz, const float *feat, const float *new_xyz,
                   const int *idx, const int *cnt, int nl, const int *dims, const void *const *wpack,
                   const float *const *bias, float *out, int out_stride, int out_off, const int *plan_hdr,
                   const int *plan_gran, long max_tiles, int fp16, int force, int dry, int *overflow, hipStream_t stream, int *st) {
    if (!fp16 || nl != 3 || c < 8 || (c & 7) || !feat) return 0;
    if (dims[1] != 32 * kNW || (dims[2] & 31) || dims[2] > 512 || dims[2] < 128 || dims[3] < 32 * 2 * kNW || dims[3] > 2048) return 0;
    // measured on layer4 of 3dssd.yaml: 259-256-512-1024 96 -> 88 us against group_mlp_wide_kernel's 105; 259-256-256-512
    // 48 us against mlp_rs_kernel's 35 (LDS-streamed weights win while the whole scale's weights are small): the narrower
    // scale stays where it was unless the caller forces this kernel (flags bit 5, tests)
    if (!force && (long)dims[2] * dims[3] < 512l * 1024) return 0;
    W128Params P{};
    P.xyz = xyz; P.feat = feat; P.new_xyz = new_xyz; P.idx = idx; P.cnt = cnt; P.out = out;
    P.n = n; P.m = m; P.ns = ns; P.C = c; P.out_stride = out_stride; P.out_off = out_off;
    P.gran = plan_gran; P.hdr = plan_hdr; P.ovf = overflow;
    for (int l = 0; l < 3; ++l) {
        P.L[l].w = (const uint4 *)wpack[l];
        P.L[l].bias = bias[l];
        P.L[l].KS = r128_roundup(dims[l], 16) / 16;
        P.L[l].NT = r128_roundup(dims[l + 1], 32) / 32;
        P.L[l].N = dims[l + 1];
    }
    const int wA = P.L[0].KS * 16 > dims[2] ? P.L[0].KS * 16 : dims[2];                     // gathered input row / hidden 2
    P.strideA = wA * 2 + 16;
    P.strideB = dims[1] * 2 + 16;
    const size_t lds = (size_t)kRows * (P.strideA + P.strideB);
    if (lds > 160 * 1024) return 0;
    if (dry) { *st = SA_OK; return 1; }          // the shape would be taken (nothing launched)
    (void)hipFuncSetAttribute((const void *)group_mlp_wide128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    const long nitems = (max_tiles + kRT - 1) / kRT;
    // a PERSISTENT grid, one workgroup per CU (150 KB of LDS: no second one fits): until round 5 the grid was one
    // workgroup per item of the densest plan (16 384 at 128 frames for 6 100 real items), so every item started cold --
    // its row references (two dependent global round trips) and biases with nothing to overlap them -- and the
    // "resolved one item ahead" prefetch in the kernel never had a next item.  SA_W96_GRID (tuning build) = workgroups.
    static const int grid_knob = SA_KNOB("SA_W96_GRID", 0);
    int cus = 256;
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        (void)hipGetLastError();
    }
    const long cap = grid_knob > 0 ? grid_knob : cus;
    const int grid = (int)(nitems < cap ? (nitems > 0 ? nitems : 1) : cap);
    hipLaunchKernelGGL(group_mlp_wide128_kernel, dim3(grid), dim3(kThreads), lds, stream, P);
    *st = hipGetLastError() == hipSuccess ? SA_OK : SA_ERR_LAUNCH;
    return 1;
}

#ifdef SA_W96_PROF
extern "C" int sa_debug_w96_prof(unsigned long long *host8, int reset) {
    if (host8 && hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_w96_prof), sizeof(g_w96_prof)) != hipSuccess) return SA_ERR_LAUNCH;
    if (reset) {
        void *d = nullptr;
        if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_w96_prof)) != hipSuccess) return SA_ERR_LAUNCH;
        if (hipMemset(d, 0, sizeof(g_w96_prof)) != hipSuccess) return SA_ERR_LAUNCH;
    }
    return SA_OK;
}
#endif
