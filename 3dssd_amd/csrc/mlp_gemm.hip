// Grouped MLP of the WIDE scales as a chain of large-tile GEMMs (layer4 of 3dssd.yaml: 259 -> 256 -> 256|512 -> 512|1024).
//
// Reference op sequence: group_point x2, concat, tf_util.conv2d x3, reduce_max, mask (lib/utils/layers_util.py:157-181,
// lib/utils/tf_util.py:127-201).  The fused kernels of mlp.hip / mlp_rowwave.hip keep a 32..64-row tile on one
// workgroup from gather to pooled output; for the wide scales that costs (a) a weight stream per 64 rows (1.42 MB per
// item at 512 -> 1024: 64 flop per weight byte), (b) a front -- plan entry -> index -> point -> feature loads, two
// hidden layers with a barrier each -- that one workgroup per CU cannot overlap with anything (0.063 of the 0.107 ms
// of group_mlp_wide_kernel for 28 % of its arithmetic), (c) a 2.25 -> 3 rounds tail of 35 us items.
// Here a scale is three launches over the rows of the row plan (mlp_plan.h: only the distinct rows of every ball):
//     phase 1   gather [features, xyz - centre] -> fp16 -> GEMM with W1 -> ReLU -> H1   (fp16, packed, global)
//     phase 2   H1 x W2 -> ReLU -> H2                                                      (fp16, packed, global)
//     phase 3   H2 x W3 -> max over the rows of each ball -> relu(max + bias), mask -> out (fp32, the concat slice)
// H1 / H2 are written in MFMA OPERAND ORDER (a 1 KiB block per (32-row tile, 16-channel k-step), lane l = row l&31,
// channels 8(l>>5)..+7: the layout of the packed weights), so every operand fetch of the next phase is a contiguous
// 1 KiB wave access and needs no transposition; they stay in L2 / Infinity Cache (18 + 37 MB for layer4 scale 1).
// A workgroup (8 waves) owns a tile of BRT x BCT 32x32 MFMA tiles (256 x 256 outputs in phase 3): both operands of a
// k-step are staged ONCE per workgroup through LDS (register-staged global loads PD stages ahead, two LDS buffers, one
// barrier per stage) and every wave reads the RTw + CTw fragments of its RTw x CTw accumulator tiles from there: a
// weight byte fetched from L2 feeds 256 rows (64 before).  The hidden phases compute D^T = W^T X^T (weights as the A
// operand) so that an accumulator becomes two operand fragments of the next layer with the packed ReLU / permlane32
// swap of mlp_rowwave.hip; phase 3 computes D = X W so that a lane holds 16 rows of one channel and pooling is the
// granule maximum of mlp_plan.h.  fp16 operands (one MFMA pass, fp32 accumulate) with the range guard of mlp_act.h.
#include "sa_common.h"
#include "mlp_plan.h"
#include "mlp_act.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWaves = 4;         // per workgroup: several independent workgroups share a CU (their fill / epilogue
constexpr int kThreads = kWaves * 64;   // phases overlap each other's matrix work; with 8 waves and one workgroup per CU
                                        // fill + epilogue were 36 % of a work item's cycles, measured)
#ifndef SA_GEMM_KPS
#define SA_GEMM_KPS 2
#endif
#ifndef SA_GEMM_PD
#define SA_GEMM_PD 2
#endif
constexpr int kKPS = SA_GEMM_KPS;   // k-steps per LDS stage (one barrier per stage)
constexpr int kPD = SA_GEMM_PD;     // stages of global loads in flight ahead of the stage being computed (1..4)

#ifdef SA_GEMM_PROF
// debug build only (tools/gemm_prof.py): cycles of wave 0 per work item, summed per phase kernel:
// [0] set-up (row refs, bias), [1] pipeline fill until the first barrier, [2] stage loop, [3] epilogue, [4] work items,
// [5] stages, [6] whole kernel of workgroups that had work, [7] those workgroups
__device__ unsigned long long g_gemm_prof[4][8];
#define GP_T0() unsigned long long gp_t = __builtin_readcyclecounter(), gp_k0 = gp_t, gp_acc[6] = {0, 0, 0, 0, 0, 0}
#define GP_TICK(i) { const unsigned long long n__ = __builtin_readcyclecounter(); gp_acc[i] += n__ - gp_t; gp_t = n__; }
#define GP_ADD(i, v) gp_acc[i] += (v)
#define GP_FLUSH(ph) if (tid == 0) { for (int i__ = 0; i__ < 6; ++i__) atomicAdd(&g_gemm_prof[ph][i__], gp_acc[i__]); \
    atomicAdd(&g_gemm_prof[ph][6], __builtin_readcyclecounter() - gp_k0); atomicAdd(&g_gemm_prof[ph][7], 1ull); }
#else
#define GP_T0()
#define GP_TICK(i)
#define GP_ADD(i, v)
#define GP_FLUSH(ph)
#endif

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is fence + s_barrier, and on gfx9 (one vmcnt for loads
// and stores) the fence is `s_waitcnt vmcnt(0)`: it would drain the prefetched global loads of the next stages at every
// barrier and serialise the pipeline on the full memory latency.  Here: wait for this wave's LDS operations, barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct GemmScale {
    const uint4 *w;              // packed fp16 weights of THIS layer: [NT][KS][64] uint4
    const float *bias;           // [NT * 32]
    const uint4 *x;              // phases 2, 3: packed input activations [row tiles][KS][64]
    uint4 *y;                    // phases 1, 2: packed output activations [row tiles][2 * NT][64]
    int KS, NT;                  // k-steps (16 channels) of the input, 32-channel tiles of the output
    // phase 1: the gather (layers_util.py:157-165)
    const float *xyz, *feat, *new_xyz;
    const int *idx, *cnt;
    int n, m, ns, C;
    // row plan
    const int *hdr, *gran;
    // phase 3: pooled output (layers_util.py:178-183)
    float *out;
    int out_stride, out_off, N;
    int *ovf;
};
struct GemmArgs { GemmScale s[2]; };

__device__ __forceinline__ f32x16 mfma_f16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// D^T accumulator of a hidden layer (reg r of lane (row, h) = channel (r&3) + 8*(r>>2) + 4h of the 32-channel tile)
// -> bias already inside -> ReLU -> the two operand fragments (k-steps 2ct, 2ct+1 of the next layer) of this lane:
// convert + packed ReLU first, then swap the packed pairs between the lane halves (as acc_to_frags of mlp_rowwave.hip)
__device__ __forceinline__ void acc_to_frag_pair(const f32x16 &acc, uint4 &f0, uint4 &f1) {
#pragma unroll
    for (int hk = 0; hk < 2; ++hk) {
        unsigned pa[2], pb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            pa[j] = sa::cvt2_f16_relu(acc[8 * hk + 2 * j], acc[8 * hk + 2 * j + 1]);
            pb[j] = sa::cvt2_f16_relu(acc[8 * hk + 4 + 2 * j], acc[8 * hk + 4 + 2 * j + 1]);
        }
        const auto s0 = __builtin_amdgcn_permlane32_swap(pa[0], pb[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pa[1], pb[1], false, false);
        const uint4 f = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        if (hk == 0) f0 = f; else f1 = f;
    }
}

struct StageCtx { int rb, cb, ntiles, KS, w, lane, half, g_pt; float g_rel[3]; };
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector: staging arrays of HIP's uint4 struct (and
typedef float f32x4 __attribute__((ext_vector_type(4)));      // loads under branches) were kept in scratch memory by hipcc

// global loads of stage `st` (kKPS k-steps) into this wave's staging registers: fragments w, w + 8, ... of
// [BCT weight fragments | BRT activation fragments]; phase 1 gathers its activation fragment as 8 fp32 channels per
// lane.  Every load is UNCONDITIONAL (addresses are selected / clamped, never skipped): straight-line code whose
// staging arrays stay in registers; what a wave does not own is simply not written to LDS by commit_stage.
template <int PHASE, int BRT, int BCT, int NLD>
__device__ __forceinline__ void issue_stage(const GemmScale &S, const StageCtx &X, int st, u32x4 (&lw)[kKPS][NLD],
                                            f32x4 (&lg)[kKPS][2]) {
#pragma unroll
    for (int kk = 0; kk < kKPS; ++kk) {
        int ks = st * kKPS + kk;
        ks = ks < X.KS ? ks : X.KS - 1;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int f = X.w + kWaves * j;
            int ct = X.cb * BCT + (f < BCT ? f : 0);
            ct = ct < S.NT ? ct : S.NT - 1;
            const uint4 *src = S.w + ((size_t)ct * X.KS + ks) * 64;
            if (PHASE != 1) {
                int rt = X.rb * BRT + (f >= BCT ? f - BCT : 0);
                rt = rt < X.ntiles ? rt : X.ntiles - 1;
                const uint4 *srcx = S.x + ((size_t)rt * X.KS + ks) * 64;
                src = f < BCT ? src : srcx;
            }
            lw[kk][j] = *(const u32x4 *)(src + X.lane);
        }
        if (PHASE == 1) {
            const int g8 = 2 * ks + X.half, GF = S.C >> 3;                    // 8-channel group of the grouped row
            const int gc = g8 < GF ? g8 : GF - 1;
            const float *src = S.feat + (size_t)X.g_pt * S.C + 8 * gc;
            lg[kk][0] = *(const f32x4 *)src;
            lg[kk][1] = *(const f32x4 *)(src + 4);
        }
    }
}
// staging registers -> LDS buffer `buf` ([kKPS][NF][64]); phase 1 converts its gathered channels here
template <int PHASE, int BRT, int BCT, int NLD>
__device__ __forceinline__ void commit_stage(const GemmScale &S, const StageCtx &X, int st, const u32x4 (&lw)[kKPS][NLD],
                                             const f32x4 (&lg)[kKPS][2], u32x4 (*buf)[BCT + BRT][64], sa::f16_guard_t &det) {
    constexpr int NF = BCT + BRT;
#pragma unroll
    for (int kk = 0; kk < kKPS; ++kk) {
        const int ks = st * kKPS + kk;
        u32x4 gath = {0u, 0u, 0u, 0u};
        if (PHASE == 1) {
            const int g8 = 2 * ks + X.half, GF = S.C >> 3;
            float v[8] = {lg[kk][0][0], lg[kk][0][1], lg[kk][0][2], lg[kk][0][3], lg[kk][1][0], lg[kk][1][1], lg[kk][1][2], lg[kk][1][3]};
            const bool feat = g8 < GF, tail = g8 == GF;             // past the features: [dx, dy, dz, 0 ...], then zeros
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = feat ? v[e] : ((tail && e < 3) ? X.g_rel[e] : 0.0f);
            gath = u32x4{sa::cvt2_f16(v[0], v[1]), sa::cvt2_f16(v[2], v[3]), sa::cvt2_f16(v[4], v[5]), sa::cvt2_f16(v[6], v[7])};
            sa::f16_guard_signed(make_uint4(gath[0], gath[1], gath[2], gath[3]), det);
        }
        // a k-step past the end of the contraction (odd KS: the second half of the last stage) is staged with ZERO
        // weights, so that the matrix work of a stage needs no branch
        const bool live = ks < X.KS;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int f = X.w + kWaves * j;
            u32x4 v = (PHASE == 1 && f >= BCT) ? gath : lw[kk][j];
            if (!live && f < BCT) v = u32x4{0u, 0u, 0u, 0u};
            if (NF % kWaves == 0 || f < NF) buf[kk][f][X.lane] = v;
        }
    }
}

// PHASE 1: gather + first layer; 2: hidden layer on packed input; 3: last layer + pooling.
// BRT x BCT: MFMA tiles (32 rows x 32 channels) of the workgroup tile; WR: wave rows (8 / WR wave columns).
// MINW: waves per SIMD the kernel is compiled for (= workgroups per CU: 2 -> 256 registers, 3 -> 168, 4 -> 128).
template <int PHASE, int BRT, int BCT, int WR, int MINW>
__global__ __launch_bounds__(kThreads, MINW) void mlp_gemm_kernel(GemmArgs A) {
    constexpr int WC = kWaves / WR, RTw = BRT / WR, CTw = BCT / WC;
    constexpr int NF = BCT + BRT;                       // fragments of one k-step: weights first, then activations
    constexpr int NLD = (NF + kWaves - 1) / kWaves;     // fragments a wave moves per k-step
    static_assert(BRT % WR == 0 && BCT % WC == 0, "wave grid must divide the tile");
    __shared__ u32x4 s_stage[2][kKPS][NF][64];          // 2 buffers x 2 k-steps x NF KiB
    // this scale's parameters, selected FIELD BY FIELD: a struct copy (or a dynamic index into the kernel argument) is
    // kept in scratch memory by hipcc (144 bytes per lane), scalar selects land in SGPRs
    GemmScale S;
    {
        const bool s1 = blockIdx.y != 0;
#define SA_SF(f) S.f = s1 ? A.s[1].f : A.s[0].f;
        SA_SF(w) SA_SF(bias) SA_SF(x) SA_SF(y) SA_SF(KS) SA_SF(NT) SA_SF(xyz) SA_SF(feat) SA_SF(new_xyz) SA_SF(idx) SA_SF(cnt)
        SA_SF(n) SA_SF(m) SA_SF(ns) SA_SF(C) SA_SF(hdr) SA_SF(gran) SA_SF(out) SA_SF(out_stride) SA_SF(out_off) SA_SF(N) SA_SF(ovf)
#undef SA_SF
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6) & (kWaves - 1);   // (the mask tells the compiler the range)
    const int wr = w / WC, wc = w - wr * WC;
    const int half = lane >> 5, row = lane & 31;

    const int ngran = __builtin_amdgcn_readfirstlane(S.hdr[0]);
    const int ntiles = (ngran + 3) >> 2;                // 32-row tiles of the plan
    const int RB = (ntiles + BRT - 1) / BRT, CB = (S.NT + BCT - 1) / BCT;
    const int nwork = RB * CB;
    const int KS = S.KS;
    const int nstage = (KS + kKPS - 1) / kKPS;
    sa::f16_guard_t det = 0;

    int wstride;
    const int w0 = sa::xcd_block(blockIdx.x, gridDim.x, nwork, wstride);
    if (w0 < 0) return;
    GP_T0();
    for (int work = w0; work < nwork; work += wstride) {
        const int rb = work / CB, cb = work - rb * CB;  // column-fastest: neighbouring workgroups share the rows
        GP_ADD(4, 1); GP_ADD(5, nstage);
        // ---- this wave's share of the loads of a k-step: fragments w, w + 8, ... of [BCT weights | BRT activations]
        // phase 1: activation fragment g belongs to row tile rb*BRT + g and is GATHERED by the wave that moves it
        int g_pt = 0;                                   // phase 1: flat source point of this lane's row
        float g_rel[3] = {0.f, 0.f, 0.f};               // and its coordinates relative to the ball centre
        int g_frag = -1;                                // which activation fragment this wave gathers (phase 1)
        if (PHASE == 1) {
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int f = w + kWaves * j;
                if (f >= BCT && f < NF) g_frag = f - BCT;
            }
            if (g_frag >= 0) {
                int rt = rb * BRT + g_frag;
                rt = rt < ntiles ? rt : ntiles - 1;
                const int ent = sa::plan_entry(S.gran, ngran, rt * 4 + (row >> 3));
                const int ball = ent >= 0 ? sa::plan_ball(ent) : 0;
                const int smp = sa::plan_sample(ent, row & 7, S.ns);
                const int a_raw = S.idx[(size_t)ball * S.ns + smp];
                const int c = S.cnt[ball];
                const int a = c > 0 ? a_raw : 0;        // layers_util.py:157-159
                g_pt = (ball / S.m) * S.n + a;
                g_rel[0] = S.xyz[(size_t)g_pt * 3 + 0] - S.new_xyz[(size_t)ball * 3 + 0];
                g_rel[1] = S.xyz[(size_t)g_pt * 3 + 1] - S.new_xyz[(size_t)ball * 3 + 1];
                g_rel[2] = S.xyz[(size_t)g_pt * 3 + 2] - S.new_xyz[(size_t)ball * 3 + 2];
            }
        }
        static_assert(PHASE != 1 || BRT <= kWaves, "phase 1: one gathered fragment (row tile) per wave at most");

        // staging registers: stage s lives in ring slot s % kPD (issue_stage / commit_stage below)
        u32x4 ldw0[kKPS][NLD], ldw1[kKPS][NLD], ldw2[kKPS][NLD], ldw3[kKPS][NLD];   // packed fragments (weights; activations in phases 2, 3)
        f32x4 ldg0[kKPS][2], ldg1[kKPS][2], ldg2[kKPS][2], ldg3[kKPS][2];           // phase 1: the 8 gathered fp32 channels of this lane
        const StageCtx X{rb, cb, ntiles, KS, w, lane, half, g_pt, {g_rel[0], g_rel[1], g_rel[2]}};
        // phase 3: the plan entries and ball counts of this wave's row tiles (wave-uniform; fetched NOW, two dependent
        // loads that cost ~6 000 cycles when they were issued in the epilogue)
        int p_ent[RTw][4], p_cn[RTw][4];
        if (PHASE == 3) {
#pragma unroll
            for (int ri = 0; ri < RTw; ++ri) {
                int rt = rb * BRT + wr * RTw + ri;
                rt = rt < ntiles ? rt : ntiles - 1;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int e = sa::plan_entry(S.gran, ngran, rt * 4 + g);
                    p_ent[ri][g] = __builtin_amdgcn_readfirstlane(e);
                    p_cn[ri][g] = __builtin_amdgcn_readfirstlane(e >= 0 ? S.cnt[sa::plan_ball(e)] : 0);
                }
            }
        }
        f32x16 acc[CTw][RTw];
#pragma unroll
        for (int ci = 0; ci < CTw; ++ci) {
            int ct = cb * BCT + wc * CTw + ci;
            ct = ct < S.NT ? ct : S.NT - 1;
#pragma unroll
            for (int ri = 0; ri < RTw; ++ri) {
                if (PHASE == 3) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ci][ri][r] = 0.0f;
                } else {                                // D^T form: reg r = channel (r&3) + 8*(r>>2) + 4*half
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 bv = *(const float4 *)(S.bias + ct * 32 + 8 * q + 4 * half);
                        acc[ci][ri][4 * q + 0] = bv.x; acc[ci][ri][4 * q + 1] = bv.y;
                        acc[ci][ri][4 * q + 2] = bv.z; acc[ci][ri][4 * q + 3] = bv.w;
                    }
                }
            }
        }

        // ---- pipeline: loads of stages 0 .. kPD-1 in flight; stage st: commit -> issue st + kPD -> barrier -> MFMAs.
        //      The two ring slots are two NAMED register sets (an array indexed by the slot ended up in scratch).
        static_assert(kPD >= 1 && kPD <= 4, "the stage loop below names up to four staging slots");
        auto do_stage = [&](int st, u32x4 (&lw)[kKPS][NLD], f32x4 (&lg)[kKPS][2]) __attribute__((always_inline)) {
            const int buf = st & 1;
            commit_stage<PHASE, BRT, BCT>(S, X, st, lw, lg, s_stage[buf], det);
            issue_stage<PHASE, BRT, BCT>(S, X, st + kPD < nstage ? st + kPD : nstage - 1, lw, lg);
            lds_barrier();
            // all fragments of the stage first, then its MFMAs: one straight block the scheduler can interleave
            uint4 fw[kKPS][CTw], fx[kKPS][RTw];
#pragma unroll
            for (int kk = 0; kk < kKPS; ++kk) {
#pragma unroll
                for (int ci = 0; ci < CTw; ++ci) fw[kk][ci] = __builtin_bit_cast(uint4, s_stage[buf][kk][wc * CTw + ci][lane]);
#pragma unroll
                for (int ri = 0; ri < RTw; ++ri) fx[kk][ri] = __builtin_bit_cast(uint4, s_stage[buf][kk][BCT + wr * RTw + ri][lane]);
            }
#pragma unroll
            for (int kk = 0; kk < kKPS; ++kk)
#pragma unroll
                for (int ci = 0; ci < CTw; ++ci)
#pragma unroll
                    for (int ri = 0; ri < RTw; ++ri)
                        acc[ci][ri] = PHASE == 3 ? mfma_f16(fx[kk][ri], fw[kk][ci], acc[ci][ri])
                                                 : mfma_f16(fw[kk][ci], fx[kk][ri], acc[ci][ri]);
        };
        const int lst = nstage - 1;
        GP_TICK(0)
        issue_stage<PHASE, BRT, BCT>(S, X, 0, ldw0, ldg0);
        if (kPD > 1) issue_stage<PHASE, BRT, BCT>(S, X, 1 < lst ? 1 : lst, ldw1, ldg1);
        if (kPD > 2) issue_stage<PHASE, BRT, BCT>(S, X, 2 < lst ? 2 : lst, ldw2, ldg2);
        if (kPD > 3) issue_stage<PHASE, BRT, BCT>(S, X, 3 < lst ? 3 : lst, ldw3, ldg3);
        for (int st0 = 0; st0 < nstage; st0 += kPD) {
            do_stage(st0, ldw0, ldg0);
#ifdef SA_GEMM_PROF
            if (st0 == 0) GP_TICK(1)
#endif
            if (kPD > 1 && st0 + 1 < nstage) do_stage(st0 + 1, ldw1, ldg1);
            if (kPD > 2 && st0 + 2 < nstage) do_stage(st0 + 2, ldw2, ldg2);
            if (kPD > 3 && st0 + 3 < nstage) do_stage(st0 + 3, ldw3, ldg3);
        }
        lds_barrier();                                  // the next work item's first commit reuses buffer 0
        GP_TICK(2)

        // ---- epilogue
#pragma unroll
        for (int ri = 0; ri < RTw; ++ri) {
            const int rt = rb * BRT + wr * RTw + ri;
            if (rt >= ntiles) continue;                 // wave-uniform
            if (PHASE == 3) {
                int ent[4], cn[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) { ent[g] = p_ent[ri][g]; cn[g] = p_cn[ri][g]; }
#pragma unroll
                for (int ci = 0; ci < CTw; ++ci) {
                    const int ct = cb * BCT + wc * CTw + ci;
                    if (ct >= S.NT) continue;
                    float qm[4];
                    sa::granule_max(acc[ci][ri], qm);
                    const int c = ct * 32 + row;
                    sa::pool_write_tile(qm, ent, cn, S.bias[c], c, S.N, S.out, S.out_stride, S.out_off, lane);
                }
            } else {
                const int KSo = 2 * S.NT;               // k-steps of the next layer's input
#pragma unroll
                for (int ci = 0; ci < CTw; ++ci) {
                    const int ct = cb * BCT + wc * CTw + ci;
                    if (ct >= S.NT) continue;
                    uint4 f0, f1;
                    acc_to_frag_pair(acc[ci][ri], f0, f1);
                    sa::f16_guard(f0, det);
                    sa::f16_guard(f1, det);
                    uint4 *dst = S.y + ((size_t)rt * KSo + 2 * ct) * 64 + lane;
                    dst[0] = f0;
                    dst[64] = f1;
                }
            }
        }
        GP_TICK(3)
    }
    if (PHASE != 3) sa::f16_overflow_report(det, S.ovf, lane);
    GP_FLUSH(PHASE)
}

int roundup_i(int x, int q) { return (x + q - 1) / q * q; }

int gemm_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

}  // namespace

// Scratch of one scale for the GEMM chain: packed H1 and H2 for the densest row plan (max_tiles 32-row tiles).
size_t sa_mlp_gemm_scratch_bytes(long max_tiles, const int *dims) {
    const size_t nt1 = (size_t)roundup_i(dims[1], 32) / 32, nt2 = (size_t)roundup_i(dims[2], 32) / 32;
    return (size_t)max_tiles * (2 * nt1 + 2 * nt2) * 1024;
}

// Can the chain take this scale?  Three layers, fp16 planes packed contiguously or not (each layer has its own pointer),
// a feature count the in-kernel gather handles, hidden widths that are whole 32-channel tiles (a padded tile of a hidden
// layer would feed garbage channels into the next layer only if its weights rows were non-zero: the packer zero-pads,
// but the packed H layout assumes KS_next == 2 * NT), and enough work to fill the chip.
bool sa_mlp_gemm_eligible(int c, int nl, const int *dims, int fp16) {
    if (!fp16 || nl != 3 || c < 8 || (c & 7)) return false;
    if ((dims[1] & 31) || (dims[2] & 31)) return false;
    return dims[1] >= 128 && dims[2] >= 128 && dims[3] >= 256;
}

// nscale (1 or 2) scales of one SA layer that share b, n, m, c and the input tensors; per scale: nsample, idx / cnt,
// widths dims[4 i ..], packed fp16 weights / biases, output slice, plan (hdr, gran) and scratch (H1 | H2).
int sa_mlp_gemm_launch(int nscale, int b, int n, int m, const int *ns, int c, const float *xyz, const float *feat,
                       const float *new_xyz, const int *const *idx, const int *const *cnt, const int *dims,
                       const void *const *wpack, const float *const *bias, float *out, int out_stride, const int *out_off,
                       const int *const *plan_hdr, const int *const *plan_gran, const long *max_tiles,
                       void *const *scratch, int *overflow, hipStream_t stream) {
    if (nscale < 1 || nscale > 2) return SA_ERR_UNSUPPORTED;
    GemmArgs P1{}, P2{}, P3{};
    long mt = 0;
    int nt1m = 0, nt2m = 0, nt3m = 0;
    for (int i = 0; i < nscale; ++i) {
        const int *d = dims + 4 * i;
        const int KS0 = roundup_i(d[0], 16) / 16, NT1 = d[1] / 32, NT2 = d[2] / 32, NT3 = roundup_i(d[3], 32) / 32;
        uint4 *H1 = (uint4 *)scratch[i];
        uint4 *H2 = H1 + (size_t)max_tiles[i] * 2 * NT1 * 64;
        GemmScale base{};
        base.xyz = xyz; base.feat = feat; base.new_xyz = new_xyz; base.idx = idx[i]; base.cnt = cnt[i];
        base.n = n; base.m = m; base.ns = ns[i]; base.C = c;
        base.hdr = plan_hdr[i]; base.gran = plan_gran[i];
        base.out = out; base.out_stride = out_stride; base.out_off = out_off[i]; base.N = d[3];
        base.ovf = overflow;
        P1.s[i] = base; P1.s[i].w = (const uint4 *)wpack[3 * i + 0]; P1.s[i].bias = bias[3 * i + 0];
        P1.s[i].KS = KS0; P1.s[i].NT = NT1; P1.s[i].y = H1;
        P2.s[i] = base; P2.s[i].w = (const uint4 *)wpack[3 * i + 1]; P2.s[i].bias = bias[3 * i + 1];
        P2.s[i].KS = 2 * NT1; P2.s[i].NT = NT2; P2.s[i].x = H1; P2.s[i].y = H2;
        P3.s[i] = base; P3.s[i].w = (const uint4 *)wpack[3 * i + 2]; P3.s[i].bias = bias[3 * i + 2];
        P3.s[i].KS = 2 * NT2; P3.s[i].NT = NT3; P3.s[i].x = H2;
        if (max_tiles[i] > mt) mt = max_tiles[i];
        if (NT1 > nt1m) nt1m = NT1;
        if (NT2 > nt2m) nt2m = NT2;
        if (NT3 > nt3m) nt3m = NT3;
    }
    // persistent grids: as many workgroups as fit the chip at once (wgs per CU = the kernel's waves per SIMD), at most
    // the work items of the densest plan; workgroups without work leave at once
    const long cus = gemm_num_cus();
    auto grid_for = [&](int brt, int bct, int ntm, int per_cu) {
        long g = ((mt + brt - 1) / brt) * ((ntm + bct - 1) / bct);
        const long cap = cus * per_cu;
        return (unsigned)(g < cap ? (g > 0 ? g : 1) : cap);
    };
    hipLaunchKernelGGL((mlp_gemm_kernel<1, 4, 4, 2, 3>), dim3(grid_for(4, 4, nt1m, 3), nscale), dim3(kThreads), 0, stream, P1);
    SA_CHECK_LAUNCH();
    hipLaunchKernelGGL((mlp_gemm_kernel<2, 4, 4, 2, 4>), dim3(grid_for(4, 4, nt2m, 4), nscale), dim3(kThreads), 0, stream, P2);
    SA_CHECK_LAUNCH();
    hipLaunchKernelGGL((mlp_gemm_kernel<3, 4, 8, 2, 2>), dim3(grid_for(4, 8, nt3m, 2), nscale), dim3(kThreads), 0, stream, P3);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

#ifdef SA_GEMM_PROF
extern "C" int sa_debug_gemm_prof(unsigned long long *host32, int reset) {
    if (host32 && hipMemcpyFromSymbol(host32, HIP_SYMBOL(g_gemm_prof), sizeof(g_gemm_prof)) != hipSuccess) return SA_ERR_LAUNCH;
    if (reset) {
        void *d = nullptr;
        if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_gemm_prof)) != hipSuccess) return SA_ERR_LAUNCH;
        if (hipMemset(d, 0, sizeof(g_gemm_prof)) != hipSuccess) return SA_ERR_LAUNCH;
    }
    return SA_OK;
}
#endif
