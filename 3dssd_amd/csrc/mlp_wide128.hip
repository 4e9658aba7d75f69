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
//
// Round 6 -- the item's phases overlap through LDS.  Until round 5 the gather of an item (12 100 of its 64 800 cycles:
// 96 rows x 1 KiB of features, load latency + conversion) ran with the matrix pipe idle, and hiding it through
// registers (54 per lane) made hipcc spill.  LDS is now THREE regions of 96 rows x 272 fp16 channels (157.5 KiB); each
// holds the gathered input, the first hidden layer, or one 256-column half of the second:
//     item q   input in I, hidden 1 -> H, hidden 2 -> (R2 | I), last layer reads (R2 | I) WHILE item q+1 is gathered
//              into H (free since hidden 2 finished): one 16-byte group per thread and block of k-steps, requested at
//              the top of a block and converted + stored at the top of the next (8 registers in flight, not 54);
//     item q+1 input in H: the roles of I and H swap, R2 never moves.
// The xyz tail of the next input (192 groups) is requested in front of hidden 2 and stored behind it.  The second
// hidden layer takes its two column tiles per wave TOGETHER (2 x 3 accumulator tiles like the last layer), so an
// activation fragment read from LDS feeds two MFMAs instead of one -- the hidden layers sat on LDS operand reads
// (128 bytes per clock and CU = one 1 KiB fragment per MFMA issue slot of the CU).
#include "sa_common.h"
#include "mlp_plan.h"
#include "mlp_act.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector for the weight ring (plain loads / moves)

#ifndef SA_W96_NW
#define SA_W96_NW 8      // waves per workgroup: 8 (two per SIMD, two column tiles per wave and pass) or 16 (four per SIMD, one)
#endif
constexpr int kNW = SA_W96_NW, kThreads = kNW * 64;
constexpr int kTPW = 16 / kNW;                       // column tiles of a wave in a pass of 16 over the last layer / in hidden 2
constexpr int kH1W = 8;                              // column tiles of hidden 1 (width 256): the first 8 waves run it
static_assert(kNW == 8 || kNW == 16, "8 or 16 waves");
constexpr int kRows = 96, kRT = kRows / 32;         // row tiles of an item
constexpr int kGB = 16;                              // LDS bytes of an 8-channel group (one fp16 plane)
// weight fragments (k-steps) in flight per wave and column tile.  (A ring kept ALIVE across the phases of an item -- the
// last block of a phase fetching the first fragments of the next -- was built in round 6: with 64 more registers live
// through the epilogues and the pooling hipcc spilled 92 of them and the item took 98 000 cycles instead of 67 000.)
#ifndef SA_W96_DEPTH
#define SA_W96_DEPTH 8
#endif
#ifndef SA_W96_H2PAIR
#define SA_W96_H2PAIR 1  // hidden 2: the two column tiles of a wave together (an LDS activation fragment feeds two MFMAs)
#endif
#ifndef SA_W96_SBMASK
#define SA_W96_SBMASK 0x78F  // sched_barrier: everything but memory reads / writes (VMEM) may cross
#endif
#ifndef SA_W96_SBMASK_DS
#define SA_W96_SBMASK_DS 0      // sched_barrier between a fragment's MFMAs and its re-read: nothing crosses (0x47F = all but LDS: hipcc bunches the reads again)
#endif
#ifndef SA_W96_GDEPTH
#define SA_W96_GDEPTH 1  // gather steps in flight under the last layer (8 registers each)
#endif
constexpr int kD = SA_W96_DEPTH;

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
    int rstride;            // LDS row stride in bytes of a region (96 rows x max(padded input, 256) fp16 channels + 16)
    int ksplit;             // k-steps of hidden 2 held by region 2 (a run-time value: as a constant hipcc peels the blocks
                            // of the last layer in front of it and loses the weight pipeline across the peeled copies)
    float inv_gf;           // 1 / (C / 8): feature groups per row
    int *ovf;
};

#ifdef SA_W96_PROF
// debug build only (tools/w96_prof.py): cycles of wave 0 per item: [0] plan entries (+ the first item's gather),
// [1] hidden 1, [2] hidden 2, [3] last layer MFMA loops (the next item's gather rides in them), [4] pooling / write-out,
// [5] items
__device__ unsigned long long g_w96_prof[16];   // [8 + w]: cycles wave w spent in phase 0 (the wait at the top-of-item barrier)
#define WP_T0() unsigned long long wp_t = __builtin_readcyclecounter(), wp_acc[6] = {0, 0, 0, 0, 0, 0}
#define WP_TICK(i) { const unsigned long long n__ = __builtin_readcyclecounter(); wp_acc[i] += n__ - wp_t; wp_t = n__; }
#define WP_FLUSH() { if (tid == 0) { for (int i__ = 0; i__ < 6; ++i__) atomicAdd(&g_w96_prof[i__], wp_acc[i__]); } if (lane == 0) atomicAdd(&g_w96_prof[8 + w], wp_acc[0]); }
#else
#define WP_T0()
#define WP_TICK(i)
#define WP_FLUSH()
#endif

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is fence + s_barrier, and on gfx9 (one vmcnt for loads
// and stores) the fence is `s_waitcnt vmcnt(0)`: it drains every global load in flight -- here the weight ring's
// fragments for the next phase, the next item's row references and gather -- a full L2 round trip at each of the three
// barriers of an item.  The phases only exchange data through LDS; the pooled global stores need no ordering.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ f32x16 mfma_f16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// (ball, flat source point) of the rows lane and lane + 64 of an item: plan entry -> index / count -> point, two
// dependent global round trips.  Resolved one item AHEAD (during the previous item's hidden layers).
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

// ---- gather of an item's 96 rows x [features, xyz - centre, 0 padding] into a region as fp16 (row-major, 16 bytes per
// 8-channel group), from the rows resolved by resolve_rows (every wave holds all of them: lane -> rows lane, lane + 64).
// The feature part runs as a PIPE of steps: step j covers groups j * 512 + tid of the item (row-major); gp_tick() stores
// the step requested by the previous tick and requests the next one -- no branch, steps past the end re-read the last
// group and store nothing -- so the last layer's blocks of k-steps carry one tick each.
constexpr int kGD = SA_W96_GDEPTH;
struct GatherPipe {
    float4 f0[kGD], f1[kGD];   // the groups in flight, oldest first
    int off[kGD];              // their LDS byte offsets from the region's base (an idle step: the dummy slot)
    int j;                     // next step to request
};
__device__ __forceinline__ void gp_reset(GatherPipe &G, int dummy_off) {
#pragma unroll
    for (int d = 0; d < kGD; ++d) { G.f0[d] = make_float4(0.f, 0.f, 0.f, 0.f); G.f1[d] = G.f0[d]; G.off[d] = dummy_off; }
    G.j = 0;
}
__device__ __forceinline__ void gp_tick(const W128Params &P, unsigned char *buf, int dummy_off, const RowRefs &R, int tid,
                                        GatherPipe &G, sa::f16_guard_t &det) {
    {   // store the oldest group in flight (requested kGD ticks ago)
        const float4 a = G.f0[0], b = G.f1[0];
        const uint4 r = make_uint4(sa::cvt2_f16(a.x, a.y), sa::cvt2_f16(a.z, a.w), sa::cvt2_f16(b.x, b.y), sa::cvt2_f16(b.z, b.w));
        sa::f16_guard_signed(r, det);
        *(uint4 *)(buf + G.off[0]) = r;                     // an idle step stores into the dummy slot: no branch in the block
#pragma unroll
        for (int d = 0; d + 1 < kGD; ++d) { G.f0[d] = G.f0[d + 1]; G.f1[d] = G.f1[d + 1]; G.off[d] = G.off[d + 1]; }
    }
    const int GF = P.C >> 3, totF = kRows * GF;
    const int it_raw = G.j * kThreads + tid;
    const bool live = it_raw < totF;
    const int it = live ? it_raw : totF - 1;
    int row = (int)((float)it * P.inv_gf);                   // it < 2^16: one float estimate, exact after +-1
    int g = it - row * GF;
    { const bool hi = g >= GF, lo = g < 0; row += hi ? 1 : (lo ? -1 : 0); g += hi ? -GF : (lo ? GF : 0); }
    const int p0 = __shfl(R.pt[0], row & 63), p1 = __shfl(R.pt[1], row & 63);
    const long pt = row < 64 ? p0 : p1;
    G.f0[kGD - 1] = *(const float4 *)(P.feat + pt * P.C + g * 8);
    G.f1[kGD - 1] = *(const float4 *)(P.feat + pt * P.C + g * 8 + 4);
    G.off[kGD - 1] = live ? row * P.rstride + g * kGB : dummy_off;
    G.j += 1;
}
__device__ __forceinline__ int gp_steps(const W128Params &P) { return (kRows * (P.C >> 3) + kThreads - 1) / kThreads + kGD; }   // + the last stores

// The tail groups of a row: [dx, dy, dz, 0 ...], then zero groups up to the padded width; at most 2 per row (c % 8 == 0),
// i.e. one step of the 512 threads.  Requested (gt_issue) well before they are stored (gt_store).
struct GatherTail { float p[3], c[3]; };
__device__ __forceinline__ GatherTail gt_issue(const W128Params &P, const RowRefs &R, int tid) {
    const int GF = P.C >> 3, GT = P.L[0].KS * 2 - GF, totT = kRows * GT;
    const int it = tid < totT ? tid : totT - 1;
    const int row = GT == 2 ? it >> 1 : it;
    const int p0 = __shfl(R.pt[0], row & 63), p1 = __shfl(R.pt[1], row & 63);
    const int b0 = __shfl(R.ball[0], row & 63), b1 = __shfl(R.ball[1], row & 63);
    const long pt = row < 64 ? p0 : p1, ball = row < 64 ? b0 : b1;
    GatherTail T;
#pragma unroll
    for (int e = 0; e < 3; ++e) { T.p[e] = P.xyz[pt * 3 + e]; T.c[e] = P.new_xyz[ball * 3 + e]; }
    return T;
}
__device__ __forceinline__ void gt_store(const W128Params &P, unsigned char *buf, const GatherTail &T, int tid, sa::f16_guard_t &det) {
    const int GF = P.C >> 3, GT = P.L[0].KS * 2 - GF, totT = kRows * GT;
    const int it = tid < totT ? tid : totT - 1;
    const int row = GT == 2 ? it >> 1 : it, g = GF + (GT == 2 ? it & 1 : 0);
    const bool first = g == GF;
    const float px = T.p[0] - T.c[0], py = T.p[1] - T.c[1], pz = T.p[2] - T.c[2];
    const uint4 r = make_uint4(sa::cvt2_f16(first ? px : 0.0f, first ? py : 0.0f), sa::cvt2_f16(first ? pz : 0.0f, 0.0f), 0u, 0u);
    sa::f16_guard_signed(r, det);
    if (tid < totT) *(uint4 *)(buf + row * P.rstride + g * kGB) = r;
}

// NCT 32-column tiles of a hidden layer for the item's three row tiles (D^T form: weights are the MFMA A operand):
// out[t][row][ocol[t]*32 .. +32) = relu(bias + in[row][:] W[:, ct[t]]).  The weight fragments of DEPTH k-steps are in
// flight per column tile; an activation fragment read from LDS feeds NCT MFMAs.  TAIL: the layer's k-steps need not be
// a multiple of DEPTH (the first layer: 17 = 2 x 8 + 1) -- the remainder runs behind the ring, its first fragment
// requested in front of it.
template <int NCT, bool TAIL>
__device__ __forceinline__ void hidden_tiles(const unsigned char *in, int strideIn, unsigned char *const (&outb)[NCT], int strideOut,
                                             const W128Layer &L, const int (&ct)[NCT], const int (&ocol)[NCT], int lane,
                                             sa::f16_guard_t &det) {
    constexpr int DEPTH = kD;
    const int half = lane >> 5, col = lane & 31;
    const unsigned char *arow = in + col * strideIn + half * kGB;
    // The k loop runs in straight-line blocks of DEPTH k-steps with NO branch inside: every step consumes ring slot d and
    // refills it with the fragment DEPTH steps ahead, so DEPTH loads stay in flight across the whole loop.  The refill of
    // slot d goes BEHIND the MFMAs that read it, and no memory read may rise above them (sched_barrier): the new fragment
    // then lands in the registers the old one just left.  Until round 5 it was requested in front of them: it needed
    // registers of its own, the ring rotated through register copies at the end of the block, and the copy of the
    // youngest load was waited for there -- `s_waitcnt vmcnt(1)`, the whole pipeline drained once per block (seen in
    // the ISA); a zero fragment selected for steps past the end (v_cndmask on every loaded register) had the same effect.
    const int KS = L.KS, KSm = TAIL ? KS / DEPTH * DEPTH : KS;   // !TAIL: KS % DEPTH == 0 (host check)
    const u32x4 *wb[NCT];
    u32x4 wq[DEPTH][NCT], wt[NCT];
#pragma unroll
    for (int t = 0; t < NCT; ++t) {
        wb[t] = (const u32x4 *)L.w + (size_t)ct[t] * KS * 64 + lane;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) wq[d][t] = wb[t][d * 64];   // KS >= DEPTH (host check)
        if (TAIL) wt[t] = wb[t][(KSm < KS ? KSm : KS - 1) * 64];
    }
    // The ring is loop-carried: slot d enters the loop from the prologue load above and from the refill below.  InstCombine
    // folds such a PHI of two loads into ONE load of a PHI of the addresses, placed at the top of the iteration that uses
    // it -- the software pipeline silently becomes "load, wait, use" (seen in the ISA).  Passing the prologue values
    // through an empty asm makes the PHI's inputs differ in kind and keeps the refills where they are written.
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int t = 0; t < NCT; ++t) asm volatile("" : "+v"(wq[d][t]));
    f32x16 acc[NCT][kRT];
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bv = *(const float4 *)(L.bias + ct[t] * 32 + 8 * q + 4 * half);
#pragma unroll
            for (int r = 0; r < kRT; ++r) {
                acc[t][r][4 * q + 0] = bv.x; acc[t][r][4 * q + 1] = bv.y; acc[t][r][4 * q + 2] = bv.z; acc[t][r][4 * q + 3] = bv.w;
            }
        }
    // The activation fragments rotate IN PLACE as well: fragment r of the next k-step is requested right behind the MFMAs
    // that read fragment r of this one (no LDS read may rise above them), so it has the other fragments' MFMAs to arrive.
    // Left to itself under this register pressure hipcc read one fragment, waited for it (`lgkmcnt(0)`), issued its
    // MFMAs, read the next into the same registers, waited ...: the LDS latency sat in front of every pair of MFMAs
    // (ISA of rounds 3-5; the matrix pipe of a wave was busy a third of the time).
    uint4 af[kRT];
#pragma unroll
    for (int r = 0; r < kRT; ++r) af[r] = *(const uint4 *)(arow + r * 32 * strideIn);
    for (int ks0 = 0; ks0 < KSm; ks0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int ks = ks0 + d;
            const int kn = ks + 1 < KS ? ks + 1 : KS - 1;
            uint4 wf[NCT];
#pragma unroll
            for (int t = 0; t < NCT; ++t) wf[t] = __builtin_bit_cast(uint4, wq[d][t]);
#pragma unroll
            for (int r = 0; r < kRT; ++r) {
#pragma unroll
                for (int t = 0; t < NCT; ++t) acc[t][r] = mfma_f16(wf[t], af[r], acc[t][r]);
                __builtin_amdgcn_sched_barrier(SA_W96_SBMASK_DS);
                af[r] = *(const uint4 *)(arow + r * 32 * strideIn + kn * 2 * kGB);
            }
            __builtin_amdgcn_sched_barrier(SA_W96_SBMASK);
            const int kw = ks + DEPTH < KSm ? ks + DEPTH : KSm - 1;
#pragma unroll
            for (int t = 0; t < NCT; ++t) wq[d][t] = wb[t][kw * 64];
        }
    }
    if (TAIL) {
        for (int ks = KSm; ks < KS; ++ks) {
            const int kn = ks + 1 < KS ? ks + 1 : KS - 1;
            uint4 wf[NCT];
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                wf[t] = __builtin_bit_cast(uint4, wt[t]);
                wt[t] = wb[t][kn * 64];
            }
#pragma unroll
            for (int r = 0; r < kRT; ++r) {
#pragma unroll
                for (int t = 0; t < NCT; ++t) acc[t][r] = mfma_f16(wf[t], af[r], acc[t][r]);
                __builtin_amdgcn_sched_barrier(SA_W96_SBMASK_DS);
                af[r] = *(const uint4 *)(arow + r * 32 * strideIn + kn * 2 * kGB);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int r = 0; r < kRT; ++r) {
            unsigned char *orow = outb[t] + (r * 32 + col) * strideOut + ocol[t] * 4 * kGB + 8 * half;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint2 pk = make_uint2(sa::cvt2_f16_relu(acc[t][r][4 * q], acc[t][r][4 * q + 1]), sa::cvt2_f16_relu(acc[t][r][4 * q + 2], acc[t][r][4 * q + 3]));
                sa::f16_guard(pk, det);
                *(uint2 *)(orow + q * kGB) = pk;
            }
        }
}

// Last layer (D form), column tiles ct0, ct0 + 1 of this wave over the whole contraction (KS a multiple of kD: host
// check): k-steps below `ksplit` come from region `lo`, the others from region `hi` (the two 256-column halves of hidden
// 2; ksplit is a multiple of DEPTH or >= KS).  Every block of DEPTH k-steps carries one tick of the next item's gather into `gbuf`.
__device__ __forceinline__ void last_pass(f32x16 (&acc)[kTPW][kRT], const unsigned char *lo, const unsigned char *hi, int ksplit,
                                          int stride, const W128Layer &L, int ct0, int lane, const W128Params &P,
                                          unsigned char *gbuf, int gdummy, const RowRefs &R, int tid, GatherPipe &G,
                                          sa::f16_guard_t &det) {
    constexpr int DEPTH = kD;
    const int half = lane >> 5, col = lane & 31;
    const int lane_off = col * stride + half * kGB;
    const int KS = L.KS;
    const u32x4 *wb[kTPW];
    u32x4 wq[DEPTH][kTPW];
#pragma unroll
    for (int t = 0; t < kTPW; ++t) {
        wb[t] = (const u32x4 *)L.w + (size_t)(ct0 + t < L.NT ? ct0 + t : L.NT - 1) * KS * 64 + lane;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) wq[d][t] = wb[t][d * 64];
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int t = 0; t < kTPW; ++t) asm volatile("" : "+v"(wq[d][t]));   // see hidden_tiles
    // activation fragment r of k-step ks: region `lo` below ksplit, `hi` from there on (wave-uniform select)
    auto aptr = [&](int ks, int r) -> const uint4 * {
        const unsigned char *base = ks < ksplit ? lo + ks * 2 * kGB : hi + (ks - ksplit) * 2 * kGB;
        return (const uint4 *)(base + lane_off + r * 32 * stride);
    };
    uint4 af[kRT];
#pragma unroll
    for (int r = 0; r < kRT; ++r) af[r] = *aptr(0, r);
    for (int ks0 = 0; ks0 < KS; ks0 += DEPTH) {              // branch-free blocks of DEPTH k-steps (see hidden_tiles)
        gp_tick(P, gbuf, gdummy, R, tid, G, det);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int ks = ks0 + d;
            const int kn = ks + 1 < KS ? ks + 1 : KS - 1;
            uint4 wf[kTPW];
#pragma unroll
            for (int t = 0; t < kTPW; ++t) wf[t] = __builtin_bit_cast(uint4, wq[d][t]);   // KS % DEPTH == 0 (host check)
#pragma unroll
            for (int r = 0; r < kRT; ++r) {
#pragma unroll
                for (int t = 0; t < kTPW; ++t) acc[t][r] = mfma_f16(af[r], wf[t], acc[t][r]);
                __builtin_amdgcn_sched_barrier(SA_W96_SBMASK_DS);     // in-place rotation of the activation fragments (see hidden_tiles)
                af[r] = *aptr(kn, r);
            }
            // in-place refill behind the MFMAs that read the slot (see hidden_tiles)
            __builtin_amdgcn_sched_barrier(SA_W96_SBMASK);
            const int kw = ks + DEPTH < KS ? ks + DEPTH : KS - 1;
#pragma unroll
            for (int t = 0; t < kTPW; ++t) wq[d][t] = wb[t][kw * 64];
        }
    }
}

constexpr int kHalfKS = 16;                                   // k-steps (256 columns) of hidden 2 that live in region 2
static_assert(kHalfKS % kD == 0, "a block of k-steps of the last layer must not straddle the two halves of hidden 2");

__global__ __launch_bounds__(kThreads, kNW / 4) void group_mlp_wide128_kernel(W128Params P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int rbytes = kRows * P.rstride;
    unsigned char *regI = smem, *regH = smem + rbytes;        // input of this item / hidden 1 (and the next item's input)
    unsigned char *const reg2 = smem + 2 * rbytes;            // columns 0 .. 255 of hidden 2
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6) & (kNW - 1);
    const int ngran = __builtin_amdgcn_readfirstlane(P.hdr[0]);
    const int nitems = (ngran + kRows / 8 - 1) / (kRows / 8);      // 96-row items = 12 granules of the plan
    const W128Layer &L1 = P.L[0], &L2 = P.L[1], &L3 = P.L[2];
    const int npass = (L3.NT + 15) / 16;                      // passes of 16 column tiles over the last layer
    const int nsteps = gp_steps(P);
    sa::f16_guard_t det = 0;

    int istride;
    const int item0 = sa::xcd_block(blockIdx.x, gridDim.x, nitems, istride);
    if (item0 < 0) return;
    WP_T0();
    GatherPipe gp;
    // the dummy slot sits behind the three regions; as an offset from region I / region H
    int dumI = 3 * rbytes, dumH = 2 * rbytes;
    RowRefs refs = resolve_rows(P, item0, ngran, lane);
    int gr_ent = sa::plan_entry(P.gran, ngran, item0 * (kRows / 8) + (lane < kRT * 4 ? lane : 0));
    int gr_cnt = gr_ent >= 0 ? P.cnt[sa::plan_ball(gr_ent)] : 0;
    {   // the first item of this workgroup: gathered with nothing to hide behind
        const GatherTail T = gt_issue(P, refs, tid);
        gp_reset(gp, dumI);
        for (int s = 0; s < nsteps; ++s) gp_tick(P, regI, dumI, refs, tid, gp, det);
        gt_store(P, regI, T, tid, det);
    }
    for (int item = item0; item < nitems; item += istride) {
        // the plan entries and ball counts of the item's 12 granules (lane g holds granule g; resolved one item ahead)
        int ent[kRT][4], cn[kRT][4];
#pragma unroll
        for (int g = 0; g < kRT * 4; ++g) {
            ent[g >> 2][g & 3] = __builtin_amdgcn_readlane(gr_ent, g);
            cn[g >> 2][g & 3] = __builtin_amdgcn_readlane(gr_cnt, g);
        }
        // the next item's rows and granules: requested now, needed behind hidden 1 (the last item gathers itself again:
        // no branch, region H is free either way)
        {
            const int nxt = item + istride < nitems ? item + istride : item;
            refs = resolve_rows(P, nxt, ngran, lane);
            gr_ent = sa::plan_entry(P.gran, ngran, nxt * (kRows / 8) + (lane < kRT * 4 ? lane : 0));
            gr_cnt = gr_ent >= 0 ? P.cnt[sa::plan_ball(gr_ent)] : 0;
        }
        lds_barrier();                                        // this item's input is complete (gathered during the last item)
        WP_TICK(0)
        // ---- hidden 1: 8 column tiles, one per wave: I -> H
        if (w < kH1W) {
            unsigned char *const ob[1] = {regH};
            const int ct[1] = {w}, oc[1] = {w};
            hidden_tiles<1, true>(regI, P.rstride, ob, P.rstride, L1, ct, oc, lane, det);
        }
        lds_barrier();
        WP_TICK(1)
        // ---- hidden 2: column tiles w, w + 8 together: H -> (region 2 | I) (the gathered input is dead)
        const GatherTail tail = gt_issue(P, refs, tid);       // the next item's xyz tail, stored behind hidden 2
        for (int ct = w; ct < L2.NT; ct += (SA_W96_H2PAIR && kTPW == 2 ? 2 : 1) * kNW) {
            if (SA_W96_H2PAIR && kTPW == 2 && ct + kNW < L2.NT) {
                const int c2[2] = {ct, ct + kNW};
                unsigned char *ob[2];
                int oc[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) { ob[t] = c2[t] < kHalfKS / 2 ? reg2 : regI; oc[t] = c2[t] < kHalfKS / 2 ? c2[t] : c2[t] - kHalfKS / 2; }
                unsigned char *const obc[2] = {ob[0], ob[1]};
                hidden_tiles<2, false>(regH, P.rstride, obc, P.rstride, L2, c2, oc, lane, det);
            } else {
                unsigned char *const ob[1] = {ct < kHalfKS / 2 ? reg2 : regI};
                const int c1[1] = {ct}, oc[1] = {ct < kHalfKS / 2 ? ct : ct - kHalfKS / 2};
                hidden_tiles<1, false>(regH, P.rstride, ob, P.rstride, L2, c1, oc, lane, det);
            }
        }
        lds_barrier();
        WP_TICK(2)
        // ---- last layer in passes of 16 column tiles over the LDS-resident hidden 2; pooled write per pass.  Region H
        //      is free: the next item's input goes there, one gather step per block of k-steps
        gt_store(P, regH, tail, tid, det);
        gp_reset(gp, dumH);
        for (int p = 0; p < npass; ++p) {
            const int ct0 = p * 16 + kTPW * w;                // this wave's column tile(s) of the pass
            // their bias: requested HERE, used after the matrix loop (a load in front of the pooled write was waited
            // for -- a full memory round trip per tile)
            float bias3[kTPW];
#pragma unroll
            for (int t2 = 0; t2 < kTPW; ++t2) bias3[t2] = L3.bias[(ct0 + t2 < L3.NT ? ct0 + t2 : L3.NT - 1) * 32 + (lane & 31)];
            f32x16 acc[kTPW][kRT];
#pragma unroll
            for (int t2 = 0; t2 < kTPW; ++t2)
#pragma unroll
                for (int r = 0; r < kRT; ++r)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t2][r][e] = 0.0f;
            last_pass(acc, reg2, regI, P.ksplit, P.rstride, L3, ct0, lane, P, regH, dumH, refs, tid, gp, det);
            WP_TICK(3)
            const int col = lane & 31;
#pragma unroll
            for (int t2 = 0; t2 < kTPW; ++t2) {
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
        while (gp.j < nsteps) gp_tick(P, regH, dumH, refs, tid, gp, det);   // fewer blocks than gather steps (narrow last layers)
        { unsigned char *t = regI; regI = regH; regH = t; }
        { const int t = dumI; dumI = dumH; dumH = t; }
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
    // shape / eligibility first -- the `dry` question (sa_group_mlp_granule_rows: which kernel WOULD take the scale) is
    // answered from the scalars alone, before any pointer is looked at
    if (!fp16 || nl != 3 || c < 8 || (c & 7)) return 0;
    if (dims[1] != 32 * kH1W || (dims[2] & 31) || dims[2] > 512 || dims[2] < 128 || dims[3] < 32 * 16 || dims[3] > 2048) return 0;
    // measured on layer4 of 3dssd.yaml: 259-256-512-1024 96 -> 88 us against group_mlp_wide_kernel's 105; 259-256-256-512
    // 48 us against mlp_rs_kernel's 35 (LDS-streamed weights win while the whole scale's weights are small): the narrower
    // scale stays where it was unless the caller forces this kernel (flags bit 5, tests)
    if (!force && (long)dims[2] * dims[3] < 512l * 1024) return 0;
    // the weight rings of hidden 2 and of the last layer refill in place, without a tail: whole blocks of k-steps only
    if ((dims[1] / 16) % kD || (dims[2] / 16) % kD || r128_roundup(dims[0], 16) / 16 < kD) return 0;
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
    // a region holds the padded input row, hidden 1 (256 columns) or one half of hidden 2 (<= 256 columns)
    const int wR = P.L[0].KS * 16 > 256 ? P.L[0].KS * 16 : 256;
    P.rstride = wR * 2 + 16;
    P.inv_gf = 1.0f / (float)(c >> 3);
    P.ksplit = kHalfKS;
    const size_t lds = (size_t)3 * kRows * P.rstride + 16;
    if (lds > 160 * 1024) return 0;
    if (dry) { *st = SA_OK; return 1; }          // the shape would be taken (nothing launched, no pointer read)
    if (!feat) return 0;
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
extern "C" int sa_debug_w96_prof(unsigned long long *host8, int reset) {   // host8: 16 words
    if (host8 && hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_w96_prof), sizeof(g_w96_prof)) != hipSuccess) return SA_ERR_LAUNCH;
    if (reset) {
        void *d = nullptr;
        if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_w96_prof)) != hipSuccess) return SA_ERR_LAUNCH;
        if (hipMemset(d, 0, sizeof(g_w96_prof)) != hipSuccess) return SA_ERR_LAUNCH;
    }
    return SA_OK;
}
#endif
