// Farthest-point sampling for gfx950: D-FPS on coordinates, generic c-channel FPS, and FPS on
// a precomputed distance matrix (F-FPS).  One 1024-thread workgroup (16 wave64) per frame, like
// the reference launch (lib/utils/tf_ops/sampling/tf_sampling_g.cu:392-398), but:
//   * the frame's coordinates and the running min-distance live in VGPRs for the whole kernel
//     (16 points x 4 floats per thread at n = 16384) -- the reference re-reads both from global
//     memory in every one of the m-1 dependent iterations;
//   * the per-iteration arg-max is a 6-step wave64 all-reduce (DPP + permlane swaps) plus one
//     16-entry cross-wave stage in LDS with ONE __syncthreads, instead of a 10-level
//     shared-memory tree with a barrier per level (tf_sampling_g.cu:161-171).
// Semantics follow tf_sampling_g.cu:123-230 exactly, including the (k mod 1024, k) tie-break:
// thread t owns k = t, t+1024, ... and keeps its first strict maximum; among equal wave maxima
// the lowest lane wins, among equal workgroup maxima the lowest wave wins.
// Distance arithmetic: d = fmaf(diff, diff, d) over channels ascending (nvcc -fmad=true form of
// tf_sampling_g.cu:146-150); built with -ffp-contract=off so only the explicit fmaf fuses.
#include <stdlib.h>

#include "sa_common.h"

namespace {

constexpr int kBlock = 1024;
constexpr int kWaves = kBlock / 64;
constexpr float kInit = 1e38f;      // tf_sampling_g.cu:136
constexpr float kAbsent = -3.0e38f; // slot of a thread that owns no point: never beats best = -1

// Cross-wave stage shared by all three kernels.  Wave w has published (value, payload) in slot w of
// buffer `par`; returns the payload of the winning wave to every thread.
// BCAST: value and payload of slot lane&15 are requested together and the winner's payload is taken with
// v_readlane (scalar result) -- one LDS round trip instead of two dependent ones.  Used where only the index is
// consumed (distance-matrix / generic kernels, -4 % measured); the coordinate kernel keeps the second LDS read:
// its payload feeds every VALU instruction of the next iteration, and a VALU instruction with an SGPR source
// issues at half the rate of the all-VGPR form on gfx950 (tools/microbench/valu_rates.hip).
template <bool BCAST>
__device__ __forceinline__ float4 cross_wave_pick(const float (*s_val)[kWaves],
                                                  const float4 (*s_pt)[kWaves], int par, int lane) {
    const float v = s_val[par][lane & (kWaves - 1)];
    float4 pt;
    if (BCAST) pt = s_pt[par][lane & (kWaves - 1)];
    const float M = sa::row16_allmax(v);
    const unsigned long long eq = __ballot(v == M);
    const int ws = __builtin_ctzll(eq) & (kWaves - 1);   // lowest wave holding the maximum
    if (!BCAST) return s_pt[par][ws];
    float4 r;
    r.x = r.y = r.z = 0.0f;
    r.w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pt.w), ws));
    return r;
}

// The same for the distance-matrix sampler, which also wants the RUNNER-UP of the arg-max (the row to request early; it
// never enters the result).  Every wave publishes its maximum AND its second maximum (over its other lanes; a lane's own
// second-best point is not looked at): the runner-up is the largest of the other waves' maxima and the winner wave's second.
// It must be exact across lanes: on FPS-ordered input (layer 2 samples the layer-1 picks) the next pick is nearly always
// an index neighbour of the current one, i.e. in the SAME wave -- the best of the other waves predicts 5 % of the picks,
// the true runner-up 90-96 % (tools/ffps_spec_hitrate.py).
__device__ __forceinline__ int cross_wave_pick2(const float (*s_val)[kWaves], const float4 (*s_pt)[kWaves], int par, int lane,
                                                int &c1) {
    const int slot = lane & (kWaves - 1);
    const float v = s_val[par][slot];
    const float4 q = s_pt[par][slot];               // .x: the wave's second maximum, .y: its index, .w: index of the maximum
    const int idx = __float_as_int(q.w), idx2 = __float_as_int(q.y);
    const float M = sa::row16_allmax(v);
    const int ws = __builtin_ctzll(__ballot(v == M)) & (kWaves - 1);       // lowest wave holding the maximum
    const float r = slot == ws ? q.x : v;
    const float R = sa::row16_allmax(r);
    const int ws2 = __builtin_ctzll(__ballot(r == R)) & (kWaves - 1);
    const int a = __builtin_amdgcn_readlane(idx, ws2), b2 = __builtin_amdgcn_readlane(idx2, ws);
    c1 = ws2 == ws ? b2 : a;
    return __builtin_amdgcn_readlane(idx, ws);
}

// Optional inputs/outputs of the *_ex2 entry points: a frame stride for `inp` (so that a range slice xyz[:, s:e] of a
// larger tensor is sampled in place, no copy launch) and the picked points themselves (the gather_point that
// layers_util.py:116-119 runs right after the sampler, fused into the sampler's epilogue: one launch less per layer).
struct FpsSide {
    long in_bstride;     // elements between consecutive frames of inp (0: dense, n*c / n*n)
    float *ctr;          // [b, ., 3] centres out, or null
    long ctr_bstride;    // floats between frames of ctr
    const float *xyz;    // coordinates the centres are read from (distance-matrix kernel); null: inp itself
    long xyz_bstride;
};
// after the pick loop: o[0..m) hold idx_off + local index; every thread copies some of the picked rows
__device__ __forceinline__ void write_centres(const FpsSide &S, const float *src, const int *o, int m, int idx_off,
                                              int b, int t, int nthreads) {
    if (!S.ctr) return;
    __syncthreads();                                 // thread 0's index stores are visible to the workgroup
    float *c = S.ctr + (size_t)b * S.ctr_bstride;
    for (int i = t; i < m; i += nthreads) {
        const int k = o[i] - idx_off;
        c[i * 3 + 0] = src[k * 3 + 0]; c[i * 3 + 1] = src[k * 3 + 1]; c[i * 3 + 2] = src[k * 3 + 2];
    }
}

// ---- D-FPS, c == 3, n <= 1024*PPT, everything register resident ------------------------------
template <int PPT>
__device__ __forceinline__ void fps3_reg_body(int n, int m, const float *__restrict__ inp, int *__restrict__ out,
                                              int out_stride, int idx_off, const FpsSide &S) {
    __shared__ float s_val[2][kWaves];
    __shared__ float4 s_pt[2][kWaves];
    const int b = blockIdx.x;
    const float *p = inp + (size_t)b * (S.in_bstride ? S.in_bstride : (long)n * 3);
    int *o = out + (size_t)b * out_stride;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;

    // ext_vector_type keeps each array in consecutive VGPRs, so the winner's coordinates can be
    // fetched with one indexed register read (s_set_gpr_idx) instead of a PPT-long select chain.
    typedef float vecf __attribute__((ext_vector_type(PPT)));
    vecf px, py, pz, td;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        int k = t + kBlock * j;
        bool ok = k < n;
        int kk = ok ? k : 0;
        px[j] = p[kk * 3 + 0];
        py[j] = p[kk * 3 + 1];
        pz[j] = p[kk * 3 + 2];
        td[j] = ok ? kInit : kAbsent;
    }
    float ox = p[0], oy = p[1], oz = p[2];          // old = 0, tf_sampling_g.cu:130-133
    if (t == 0) o[0] = idx_off;

    for (int it = 1; it < m; ++it) {
        float best = -1.0f;                         // :141
        int bj = 0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            float dx = px[j] - ox, dy = py[j] - oy, dz = pz[j] - oz;
            float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            float t2 = sa::fmin_nn(d, td[j]);       // :151-153
            td[j] = t2;
            bool g = t2 > best;                     // strict: first maximum wins, :154
            best = g ? t2 : best;
            bj = g ? j : bj;
        }
        const float wmax = sa::wave_allmax(best);
        const unsigned long long cand = __ballot(best == wmax);
        const int first = __builtin_ctzll(cand);
        const int par = it & 1;
        if (lane == first) {
            const int ju = __builtin_amdgcn_readfirstlane(bj);   // one active lane: its own bj
            s_val[par][w] = wmax;
            s_pt[par][w] = make_float4(px[ju], py[ju], pz[ju], __int_as_float(t + kBlock * ju));
        }
        __syncthreads();
        const float4 wp = cross_wave_pick<false>(s_val, s_pt, par, lane);
        ox = wp.x; oy = wp.y; oz = wp.z;
        if (t == 0) o[it] = __float_as_int(wp.w) + idx_off;
    }
    write_centres(S, p, o, m, idx_off, b, t, kBlock);
}
template <int PPT>
__global__ __launch_bounds__(kBlock) void fps3_reg_kernel(int n, int m, const float *__restrict__ inp,
                                                          int *__restrict__ out, int out_stride,
                                                          int idx_off, FpsSide S) {
    fps3_reg_body<PPT>(n, m, inp, out, out_stride, idx_off, S);
}

// ---- FPS on a precomputed [n,n] distance matrix (F-FPS), n <= 1024*PPT ------------------------
template <int PPT>
__device__ __forceinline__ void fpsdist_reg_body(int n, int m, const float *__restrict__ dist, int *__restrict__ out,
                                                 int out_stride, int idx_off, const FpsSide &S) {
    __shared__ float s_val[2][kWaves];
    __shared__ float4 s_pt[2][kWaves];
    const int b = blockIdx.x;
    const float *D = dist + (size_t)b * n * n;
    int *o = out + (size_t)b * out_stride;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;

    float td[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) td[j] = (t + kBlock * j) < n ? kInit : kAbsent;
    int old = 0;
    if (t == 0) o[0] = idx_off;

    for (int it = 1; it < m; ++it) {
        const float *row = D + (size_t)old * n;     // tf_sampling_g.cu:202
        float dv[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            int k = t + kBlock * j;
            dv[j] = k < n ? row[k] : 0.0f;
        }
        float best = -1.0f;
        int bk = 0;                                 // besti starts at 0, :190-191
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            float t2 = sa::fmin_nn(dv[j], td[j]);
            td[j] = t2;
            bool g = t2 > best;
            best = g ? t2 : best;
            bk = g ? (t + kBlock * j) : bk;
        }
        const float wmax = sa::wave_allmax(best);
        const unsigned long long cand = __ballot(best == wmax);
        const int first = __builtin_ctzll(cand);
        const int par = it & 1;
        if (lane == first) {
            s_val[par][w] = wmax;
            s_pt[par][w] = make_float4(0.f, 0.f, 0.f, __int_as_float(bk));
        }
        __syncthreads();
        const float4 wp = cross_wave_pick<true>(s_val, s_pt, par, lane);
        old = __builtin_amdgcn_readfirstlane(__float_as_int(wp.w));
        if (t == 0) o[it] = old + idx_off;
    }
    if (S.xyz) write_centres(S, S.xyz + (size_t)b * S.xyz_bstride, o, m, idx_off, b, t, kBlock);
}
#ifdef SA_FFPS_SPEC_STATS
__device__ unsigned long long g_ffps_spec[4];   // debug build: [0] picks, [1] picks whose row was not the held one, [2] loop clocks, [3] workgroups
#endif
// ---- the same with the next pick's row requested AHEAD (matrices read from HBM, rows of <= 16 KB) ----
// A pick waits one memory round trip for the row of the point just picked (0.8 of its 1.2 us at the layer-2 shape: the
// 64 MB matrix of a frame lives in HBM).  Round 5: the NEXT pick is nearly always known one pick in advance -- it is the
// runner-up of the current arg-max in 90-96 % of the picks (tools/ffps_spec_hitrate.py on backbone features of three data
// variants; two picks ahead the prediction is worthless: 2-20 %).  So whenever a pick is made, the row of its runner-up
// candidate is requested into a second set of registers; a pick whose row is held there has had one pick's time of its
// latency hidden.  The VALUES are the matrix rows either way: results are bit-identical, only the waiting changes.
// Extra HBM reads: the mispredicted rows only (7 % at the layer-2 shape: counter traffic 1.16 GB against 1.08 algorithmic).
// Used where it pays (host rule below): a matrix that fits the on-chip caches (layer 3: 1 MB per frame, rows answer in
// 0.3 us) gains nothing and pays for the runner-up's second reduction (+6 %); rows of 32-64 KB (n > 4096) are bound by
// the CU's load rate, not by latency (configs[2]: no change).
template <int PPT>
__device__ __forceinline__ void fpsdist_ahead_body(int n, int m, const float *__restrict__ dist, int *__restrict__ out,
                                                 int out_stride, int idx_off, const FpsSide &S) {
    __shared__ float s_val[2][kWaves];
    __shared__ float4 s_pt[2][kWaves];
    // the picks are collected in LDS and written out once: a store per pick from thread 0 sits, in order, between the row
    // requests of its wave, and the wait for a held row then also waits for that store's acknowledgement
    __shared__ int s_picks[kBlock * PPT];
    const int b = blockIdx.x;
    const float *D = dist + (size_t)b * n * n;
    int *o = out + (size_t)b * out_stride;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;

    float td[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) td[j] = (t + kBlock * j) < n ? kInit : kAbsent;
    int old = 0;
    if (t == 0) s_picks[0] = 0;
    // Two register sets used in turn: pick `it` reads set (it & 1), whose row was requested at the START of the previous pick,
    // and first of all requests the row of its own runner-up candidate into the other set -- so a held row has had a whole
    // pick (its wait included) to arrive, and per pick the chain is (latency + arg-max) / 2 instead of their sum.  Order of
    // the requests inside a pick: the row of this pick if it is not the held one (its set is idle: the request it replaces
    // is waited for), THEN the candidate's row; the update then waits for all but the newest PPT loads -- on either path.
    int heldA = -1, heldB = -1, c1 = 0;             // rows held in the two sets | runner-up of the arg-max that produced `old`
    float svA[PPT], svB[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) { svA[j] = 0.0f; svB[j] = 0.0f; }
    auto fetch = [&](float (&dst)[PPT], int r) {
        const float *row = D + (size_t)r * n;       // tf_sampling_g.cu:202
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int k = t + kBlock * j;
            dst[j] = row[k < n ? k : 0];            // (a thread without point k reads element 0: its slot stays kAbsent under
                                                    //  the min below whatever it reads -- no branch around the load, and no select
                                                    //  behind it, which would make the request wait for its own data)
        }
    };
#ifdef SA_FFPS_SPEC_STATS
    unsigned long long misses__ = 0, clk0__ = __builtin_readcyclecounter();
#endif
    auto pick = [&](float (&cur)[PPT], int &held_cur, float (&nxt)[PPT], int &held_nxt, int it) {
#ifdef SA_FFPS_SPEC_STATS
        misses__ += old != held_cur;
#endif
        if (old != held_cur) fetch(cur, old);
        held_nxt = c1;
        fetch(nxt, c1);
        float best = -1.0f;
        int bk = 0;                                 // besti starts at 0, :190-191
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            float t2 = sa::fmin_nn(cur[j], td[j]);
            td[j] = t2;
            bool g = t2 > best;
            best = g ? t2 : best;
            bk = g ? (t + kBlock * j) : bk;
        }
        const float wmax = sa::wave_allmax(best);
        const unsigned long long cand = __ballot(best == wmax);
        const int first = __builtin_ctzll(cand);
        const int par = it & 1;
        // the wave's second maximum: over the lanes other than the winning one
        const float best2 = lane == first ? -__builtin_inff() : best;
        const float w2max = sa::wave_allmax(best2);
        const int sec = __builtin_ctzll(__ballot(best2 == w2max));
        const int bk2 = __builtin_amdgcn_readlane(bk, sec);
        if (lane == first) {
            s_val[par][w] = wmax;
            s_pt[par][w] = make_float4(w2max, __int_as_float(bk2), 0.f, __int_as_float(bk));
        }
        __syncthreads();
        old = cross_wave_pick2(s_val, s_pt, par, lane, c1);
        if (t == 0) s_picks[it] = old;
    };
    for (int it = 1; it < m; it += 2) {
        pick(svB, heldB, svA, heldA, it);
        if (it + 1 < m) pick(svA, heldA, svB, heldB, it + 1);
    }
#ifdef SA_FFPS_SPEC_STATS
    if (t == 0) {
        atomicAdd(&g_ffps_spec[0], (unsigned long long)(m - 1)); atomicAdd(&g_ffps_spec[1], misses__);
        atomicAdd(&g_ffps_spec[2], __builtin_readcyclecounter() - clk0__); atomicAdd(&g_ffps_spec[3], 1ull);
    }
#endif
    __syncthreads();
    for (int i = t; i < m; i += kBlock) o[i] = s_picks[i] + idx_off;
    if (S.ctr && S.xyz) {                           // the picked points themselves (write_centres, from the LDS copy)
        const float *src = S.xyz + (size_t)b * S.xyz_bstride;
        float *c = S.ctr + (size_t)b * S.ctr_bstride;
        for (int i = t; i < m; i += kBlock) {
            const int k = s_picks[i];
            c[i * 3 + 0] = src[k * 3 + 0]; c[i * 3 + 1] = src[k * 3 + 1]; c[i * 3 + 2] = src[k * 3 + 2];
        }
    }
}
template <int PPT, bool AHEAD>
__global__ __launch_bounds__(kBlock) void fpsdist_reg_kernel(int n, int m,
                                                             const float *__restrict__ dist,
                                                             int *__restrict__ out, int out_stride,
                                                             int idx_off, FpsSide S) {
    if (AHEAD) fpsdist_ahead_body<PPT>(n, m, dist, out, out_stride, idx_off, S);
    else fpsdist_reg_body<PPT>(n, m, dist, out, out_stride, idx_off, S);
}
// rows are requested ahead when the call's matrices exceed what the on-chip caches hold (the rule sqdist.hip uses for its
// non-temporal stores: such a matrix is read from HBM), a row is at most 16 KB, and the picks fit the kernel's LDS list
// (m > n is legal -- the surplus picks repeat -- and stays on the old loop)
static inline bool ffps_rows_ahead(int b, int n, int m) { return n <= 4 * kBlock && m <= n && (size_t)b * n * n * sizeof(float) > ((size_t)192 << 20); }

// The two samplers of an 'FS' layer (layers_util.py:93-98) -- or of a layer whose two ranges use F-FPS and D-FPS
// (:101-106) -- in ONE launch: blockIdx.y == 0 runs the matrix sampler, blockIdx.y == 1 the coordinate sampler.  They are
// independent serial chains on b workgroups each; side by side the launch lasts as long as the longer one (0.54 ms at
// layer2 instead of 0.54 + 0.29), without a second stream.
struct FpsDualArgs {
    int n, m, out_stride, idx_off;
    const float *src;       // distance matrix / coordinates
    int *out;
    FpsSide S;
};
template <int PPTF, int PPTD, bool AHEAD>
__global__ __launch_bounds__(kBlock) void fps_dual_kernel(FpsDualArgs F, FpsDualArgs D) {
    if (blockIdx.y == 0) {
        if (AHEAD) fpsdist_ahead_body<PPTF>(F.n, F.m, F.src, F.out, F.out_stride, F.idx_off, F.S);
        else fpsdist_reg_body<PPTF>(F.n, F.m, F.src, F.out, F.out_stride, F.idx_off, F.S);
    }
    else fps3_reg_body<PPTD>(D.n, D.m, D.src, D.out, D.out_stride, D.idx_off, D.S);
}

// ---- generic fallback: any c, any n; running min-distance in global `temp` like the reference --
// MODE 0: c-channel points inp[b,n,c].  MODE 1: distance matrix inp[b,n,n].
template <int MODE>
__global__ __launch_bounds__(kBlock) void fps_generic_kernel(int n, int c, int m,
                                                             const float *__restrict__ inp,
                                                             float *__restrict__ temp,
                                                             int *__restrict__ out, int out_stride,
                                                             int idx_off) {
    __shared__ float s_val[2][kWaves];
    __shared__ float4 s_pt[2][kWaves];
    const int b = blockIdx.x;
    const float *p = inp + (size_t)b * n * (MODE == 0 ? (size_t)c : (size_t)n);
    float *td = temp + (size_t)b * n;
    int *o = out + (size_t)b * out_stride;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;

    for (int k = t; k < n; k += kBlock) td[k] = kInit;
    int old = 0;
    if (t == 0) o[0] = idx_off;

    for (int it = 1; it < m; ++it) {
        float best = -1.0f;
        int bk = 0;
        const float *po = p + (size_t)old * (MODE == 0 ? c : n);
        for (int k = t; k < n; k += kBlock) {
            float d;
            if (MODE == 0) {
                const float *pk = p + (size_t)k * c;
                d = 0.0f;
                for (int l = 0; l < c; ++l) {
                    float diff = pk[l] - po[l];
                    d = __builtin_fmaf(diff, diff, d);
                }
            } else {
                d = po[k];
            }
            float t2 = sa::fmin_nn(d, td[k]);
            td[k] = t2;
            if (t2 > best) { best = t2; bk = k; }
        }
        const float wmax = sa::wave_allmax(best);
        const unsigned long long cand = __ballot(best == wmax);
        const int first = __builtin_ctzll(cand);
        const int par = it & 1;
        if (lane == first) {
            s_val[par][w] = wmax;
            s_pt[par][w] = make_float4(0.f, 0.f, 0.f, __int_as_float(bk));
        }
        __syncthreads();
        const float4 wp = cross_wave_pick<true>(s_val, s_pt, par, lane);
        old = __builtin_amdgcn_readfirstlane(__float_as_int(wp.w));
        if (t == 0) o[it] = old + idx_off;
    }
}

// ---- farthest_point_sample_with_preidx (tf_sampling_g.cu:232-318): FPS continued from an already chosen set ------
// The running minimum starts as the distance to the nearest of the m1 points preidx[b, :] (not at 1e38), the first
// output is the arg-max of that field -- found by a SERIAL ascending scan in the reference (:262-269), i.e. the lowest
// index among equal maxima, unlike the (k mod 1024, k) order of the iterations that follow -- and the remaining m-1
// picks are ordinary FPS iterations on the same field.  Same fmaf chains as the other kernels (decision A).
__global__ __launch_bounds__(kBlock) void fps_preidx_kernel(int n, int c, int m, int m1,
                                                            const float *__restrict__ inp,
                                                            const int *__restrict__ preidx,
                                                            float *__restrict__ temp, int *__restrict__ out) {
    __shared__ float s_val[2][kWaves];
    __shared__ float4 s_pt[2][kWaves];
    __shared__ unsigned s_first[kWaves];
    const int b = blockIdx.x;
    const float *p = inp + (size_t)b * n * c;
    const int *pre = preidx + (size_t)b * m1;
    float *td = temp + (size_t)b * n;
    int *o = out + (size_t)b * m;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;

    float fbest = -1.0f;                               // :261
    unsigned fj = 0xFFFFFFFFu;
    for (int j = t; j < n; j += kBlock) {
        float best = kInit;                            // :247
        for (int k = 0; k < m1; ++k) {
            const float *pp = p + (size_t)pre[k] * c;
            float d = 0.0f;
            for (int l = 0; l < c; ++l) {
                const float diff = p[(size_t)j * c + l] - pp[l];
                d = __builtin_fmaf(diff, diff, d);
            }
            best = sa::fmin_nn(best, d);
        }
        td[j] = best;
        if (best > fbest) { fbest = best; fj = (unsigned)j; }      // ascending j per thread: first maximum
    }
    {
        const float wmax = sa::wave_allmax(fbest);
        const unsigned wj = sa::wave_allmin_u32(fbest == wmax ? fj : 0xFFFFFFFFu);
        if (lane == 0) { s_val[0][w] = wmax; s_first[w] = wj; }
    }
    __syncthreads();
    int old = 0;                                       // :260 (kept when no entry exceeds -1, e.g. NaN-free empty case)
    {
        float mv = -1.0f;
        unsigned mj = 0xFFFFFFFFu;
        for (int i = 0; i < kWaves; ++i) {
            const float v = s_val[0][i];
            const unsigned j = s_first[i];
            if (j != 0xFFFFFFFFu && (v > mv || (v == mv && j < mj))) { mv = v; mj = j; }
        }
        if (mj != 0xFFFFFFFFu) old = (int)mj;
    }
    if (t == 0) o[0] = old;
    __syncthreads();                                   // s_val[0] is reused by iteration 2

    for (int it = 1; it < m; ++it) {
        float best = -1.0f;
        int bk = 0;
        const float *po = p + (size_t)old * c;
        for (int k = t; k < n; k += kBlock) {
            const float *pk = p + (size_t)k * c;
            float d = 0.0f;
            for (int l = 0; l < c; ++l) {
                const float diff = pk[l] - po[l];
                d = __builtin_fmaf(diff, diff, d);
            }
            const float t2 = sa::fmin_nn(d, td[k]);
            td[k] = t2;
            if (t2 > best) { best = t2; bk = k; }
        }
        const float wmax = sa::wave_allmax(best);
        const unsigned long long cand = __ballot(best == wmax);
        const int first = __builtin_ctzll(cand);
        const int par = it & 1;
        if (lane == first) {
            s_val[par][w] = wmax;
            s_pt[par][w] = make_float4(0.f, 0.f, 0.f, __int_as_float(bk));
        }
        __syncthreads();
        const float4 wp = cross_wave_pick<true>(s_val, s_pt, par, lane);
        old = __builtin_amdgcn_readfirstlane(__float_as_int(wp.w));
        if (t == 0) o[it] = old;
    }
}

// ---- c-channel points, any n: the frame is walked in TILE-point tiles staged through LDS ------------------
// The generic kernel above reads channel l of point k as p[k*c + l] from every lane: a 4-byte access with a
// stride of c floats, i.e. one cache line per lane and instruction (0.8 ms per iteration at n = 16384, c = 67).
// Here a tile of TILE consecutive points (TILE*c consecutive floats) is copied flat and fully coalesced into
// LDS, then point k is evaluated by thread k mod 1024 -- exactly the reference's thread <-> point assignment, so
// the (k mod 1024, k) tie-break is unchanged -- reading its row from LDS (row stride c floats: conflict-free
// for odd c).  Same fmaf chain over the channels, same running minimum in `temp`.
__global__ __launch_bounds__(kBlock) void fps_points_tiled_kernel(int n, int c, int m, int tile_pts,
                                                                  const float *__restrict__ inp,
                                                                  float *__restrict__ temp,
                                                                  int *__restrict__ out, int out_stride,
                                                                  int idx_off) {
    extern __shared__ float s_dyn[];             // tile [tile_pts * c] | old point [c]
    __shared__ float s_val[2][kWaves];
    __shared__ float4 s_pt[2][kWaves];
    float *s_tile = s_dyn, *s_po = s_dyn + (size_t)tile_pts * c;
    const int b = blockIdx.x;
    const float *p = inp + (size_t)b * n * c;
    float *td = temp + (size_t)b * n;
    int *o = out + (size_t)b * out_stride;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;

    for (int k = t; k < n; k += kBlock) td[k] = kInit;
    int old = 0;
    if (t == 0) o[0] = idx_off;

    for (int it = 1; it < m; ++it) {
        float best = -1.0f;
        int bk = 0;
        for (int l = t; l < c; l += kBlock) s_po[l] = p[(size_t)old * c + l];
        for (int k0 = 0; k0 < n; k0 += tile_pts) {
            const int np = min(tile_pts, n - k0);
            const float *src = p + (size_t)k0 * c;
            for (int e = t; e < np * c; e += kBlock) s_tile[e] = src[e];
            __syncthreads();
            const int j = (t - k0) & (kBlock - 1);            // point k0 + j belongs to thread (k0 + j) mod 1024
            if (j < np) {
                const float *row = s_tile + j * c;
                float d = 0.0f;
                for (int l = 0; l < c; ++l) {
                    const float diff = row[l] - s_po[l];
                    d = __builtin_fmaf(diff, diff, d);
                }
                const int k = k0 + j;
                const float t2 = sa::fmin_nn(d, td[k]);
                td[k] = t2;
                if (t2 > best) { best = t2; bk = k; }
            }
            __syncthreads();
        }
        const float wmax = sa::wave_allmax(best);
        const unsigned long long cand = __ballot(best == wmax);
        const int first = __builtin_ctzll(cand);
        const int par = it & 1;
        if (lane == first) {
            s_val[par][w] = wmax;
            s_pt[par][w] = make_float4(0.f, 0.f, 0.f, __int_as_float(bk));
        }
        __syncthreads();
        const float4 wp = cross_wave_pick<true>(s_val, s_pt, par, lane);
        old = __builtin_amdgcn_readfirstlane(__float_as_int(wp.w));
        if (t == 0) o[it] = old + idx_off;
    }
}

// launches the tiled kernel when a tile of at least 64 points fits LDS; returns false otherwise
bool launch_points_tiled(int b, int n, int c, int m, const float *inp, float *temp, int *out, int out_stride,
                         int idx_off, hipStream_t stream) {
    if (c < 8) return false;                     // a few channels per point: the strided form is already near-coalesced
    int tile = kBlock;
    while (tile > 64 && ((size_t)tile * c + c) * sizeof(float) > 150 * 1024) tile >>= 1;
    const size_t lds = ((size_t)tile * c + c) * sizeof(float);
    if (lds > 150 * 1024) return false;
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)fps_points_tiled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(fps_points_tiled_kernel, dim3(b), dim3(kBlock), lds, stream, n, c, m, tile, inp, temp, out,
                       out_stride, idx_off);
    return true;
}

int ppt_for(int n) {
    int ppt = (n + kBlock - 1) / kBlock;
    int r = 1;
    while (r < ppt) r <<= 1;
    return r;
}

}  // namespace

// Internal launchers with an output row stride / index offset so that an SA layer can write the
// F-FPS and D-FPS halves straight into one [b, npoint_total] index tensor (layers_util.py:96-108).
extern "C" int sa_fps_bucket_ex2(int b, int n, int m, const float *inp, long in_bstride, int *out, int out_stride,
                                 int idx_off, float *ctr, long ctr_bstride, hipStream_t stream);

extern "C" int sa_fps_coop_ex(int b, int n, int c, int m, const float *inp, float *temp, int *out, int out_stride,
                              int idx_off, int capture_ok, hipStream_t stream);

// in_bstride: elements between frames of inp (0 = dense n*c); ctr / ctr_bstride: when ctr is non-null the picked rows
// (c == 3) are also written to ctr + frame * ctr_bstride + 3 * i -- the gather_point of layers_util.py:116-119.
// Both extras need the register-resident c == 3 kernels (n <= 16384): SA_ERR_UNSUPPORTED otherwise, the caller then
// slices / gathers with separate launches.
// flags bit 0: the caller keeps every multi-workgroup sampler launch of the process on ONE stream, so on a capturing
// stream that kernel may be launched plainly (fps_coop.hip); without it a capture takes the single-workgroup kernels.
extern "C" int sa_fps_ex3(int b, int n, int c, int m, const float *inp, long in_bstride, float *temp, int *out,
                          int out_stride, int idx_off, float *ctr, long ctr_bstride, int flags, hipStream_t stream) {
    if (b <= 0 || n <= 0 || c <= 0 || m <= 0 || !inp || !out || out_stride < m) return SA_ERR_INVALID;
    const bool extras = (in_bstride != 0 && in_bstride != (long)n * c) || ctr != nullptr;
    // Layer-1 shape: the wave-bucket culled kernel (fps_bucket.hip), bit-identical output, ~1.4x faster than the
    // plain kernel at 16384 -> 4096.  SA_FPS_BUCKET_MIN_N = smallest n it is used for (0 = never).
    static const int bucket_min_n = SA_KNOB("SA_FPS_BUCKET_MIN_N", 8192);
    if (c == 3 && bucket_min_n > 0 && n >= bucket_min_n && n <= 16384 && m >= 64)
        return sa_fps_bucket_ex2(b, n, m, inp, in_bstride, out, out_stride, idx_off, ctr, ctr_bstride, stream);
    const int ppt = ppt_for(n);
    if (c == 3 && ppt <= 16) {
        FpsSide S{};
        S.in_bstride = in_bstride; S.ctr = ctr; S.ctr_bstride = ctr_bstride;
        switch (ppt) {
#define SA_FPS3(P) case P: hipLaunchKernelGGL(fps3_reg_kernel<P>, dim3(b), dim3(kBlock), 0, stream, n, m, inp, out, out_stride, idx_off, S); break;
            SA_FPS3(1) SA_FPS3(2) SA_FPS3(4) SA_FPS3(8) SA_FPS3(16)
#undef SA_FPS3
        }
    } else {
        if (extras) return SA_ERR_UNSUPPORTED;
        if (!temp) return SA_ERR_INVALID;
        // frames too large for one CU's registers/LDS: several cooperating workgroups per frame (fps_coop.hip)
        const int rc = sa_fps_coop_ex(b, n, c, m, inp, temp, out, out_stride, idx_off, flags & 1, stream);
        if (rc != SA_ERR_UNSUPPORTED) return rc;
        if (!launch_points_tiled(b, n, c, m, inp, temp, out, out_stride, idx_off, stream))
            hipLaunchKernelGGL(fps_generic_kernel<0>, dim3(b), dim3(kBlock), 0, stream, n, c, m, inp, temp,
                               out, out_stride, idx_off);
    }
    SA_CHECK_LAUNCH();
    return SA_OK;
}

extern "C" int sa_fps_ex2(int b, int n, int c, int m, const float *inp, long in_bstride, float *temp, int *out,
                          int out_stride, int idx_off, float *ctr, long ctr_bstride, hipStream_t stream) {
    return sa_fps_ex3(b, n, c, m, inp, in_bstride, temp, out, out_stride, idx_off, ctr, ctr_bstride, 0, stream);
}

extern "C" int sa_fps_ex(int b, int n, int c, int m, const float *inp, float *temp, int *out,
                         int out_stride, int idx_off, hipStream_t stream) {
    return sa_fps_ex3(b, n, c, m, inp, 0, temp, out, out_stride, idx_off, nullptr, 0, 0, stream);
}

extern "C" int sa_fps_bucket_ex(int b, int n, int m, const float *inp, int *out, int out_stride, int idx_off,
                                hipStream_t stream) {
    return sa_fps_bucket_ex2(b, n, m, inp, 0, out, out_stride, idx_off, nullptr, 0, stream);
}

// xyz / xyz_bstride / ctr / ctr_bstride: when ctr is non-null, rows xyz[frame][pick] (3 floats, frames xyz_bstride
// floats apart) are written to ctr + frame * ctr_bstride + 3 * i (register-resident kernel only, n <= 16384).
extern "C" int sa_fps_with_distance_ex2(int b, int n, int m, const float *dist, float *temp, int *out,
                                        int out_stride, int idx_off, const float *xyz, long xyz_bstride, float *ctr,
                                        long ctr_bstride, hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || !dist || !out || out_stride < m || (ctr && !xyz)) return SA_ERR_INVALID;
    const int ppt = ppt_for(n);
    if (ppt <= 16) {
        FpsSide S{};
        S.ctr = ctr; S.ctr_bstride = ctr_bstride; S.xyz = ctr ? xyz : nullptr; S.xyz_bstride = xyz_bstride;
        const bool ahead = ffps_rows_ahead(b, n, m);
        switch (ppt) {
#define SA_FPSD(P) case P: if (ahead && P <= 4) hipLaunchKernelGGL((fpsdist_reg_kernel<P, P <= 4>), dim3(b), dim3(kBlock), 0, stream, n, m, dist, out, out_stride, idx_off, S); \
                           else hipLaunchKernelGGL((fpsdist_reg_kernel<P, false>), dim3(b), dim3(kBlock), 0, stream, n, m, dist, out, out_stride, idx_off, S); break;
            SA_FPSD(1) SA_FPSD(2) SA_FPSD(4) SA_FPSD(8) SA_FPSD(16)
#undef SA_FPSD
        }
    } else {
        if (ctr) return SA_ERR_UNSUPPORTED;
        if (!temp) return SA_ERR_INVALID;
        hipLaunchKernelGGL(fps_generic_kernel<1>, dim3(b), dim3(kBlock), 0, stream, n, 0, m, dist, temp,
                           out, out_stride, idx_off);
    }
    SA_CHECK_LAUNCH();
    return SA_OK;
}

extern "C" int sa_fps_with_distance_ex(int b, int n, int m, const float *dist, float *temp, int *out,
                                       int out_stride, int idx_off, hipStream_t stream) {
    return sa_fps_with_distance_ex2(b, n, m, dist, temp, out, out_stride, idx_off, nullptr, 0, nullptr, 0, stream);
}

// F-FPS on dist [b, nf, nf] (mf picks, centres from xyz) and D-FPS on inp [b, nd, 3] (md picks) in one launch
// (fps_dual_kernel).  Register-resident kernels with the same points-per-thread class only: SA_ERR_UNSUPPORTED
// otherwise (the caller launches the two samplers one after the other).
extern "C" int sa_fps_dual_ex(int b, int nf, int mf, const float *dist, int *out_f, int out_stride_f, int idx_off_f,
                              const float *xyz_f, long xyz_bstride_f, float *ctr_f, long ctr_bstride_f, int nd, int md,
                              const float *inp, long in_bstride, int *out_d, int out_stride_d, int idx_off_d,
                              float *ctr_d, long ctr_bstride_d, hipStream_t stream) {
    if (b <= 0 || nf <= 0 || mf <= 0 || nd <= 0 || md <= 0 || !dist || !inp || !out_f || !out_d || out_stride_f < mf ||
        out_stride_d < md || (ctr_f && !xyz_f))
        return SA_ERR_INVALID;
    const int pf = ppt_for(nf), pd = ppt_for(nd);
    if (pf != pd || pf > 4 || b > 65535) return SA_ERR_UNSUPPORTED;
    FpsDualArgs F{}, D{};
    F.n = nf; F.m = mf; F.out_stride = out_stride_f; F.idx_off = idx_off_f; F.src = dist; F.out = out_f;
    F.S.ctr = ctr_f; F.S.ctr_bstride = ctr_bstride_f; F.S.xyz = ctr_f ? xyz_f : nullptr; F.S.xyz_bstride = xyz_bstride_f;
    D.n = nd; D.m = md; D.out_stride = out_stride_d; D.idx_off = idx_off_d; D.src = inp; D.out = out_d;
    D.S.in_bstride = in_bstride; D.S.ctr = ctr_d; D.S.ctr_bstride = ctr_bstride_d;
    const bool ahead = ffps_rows_ahead(b, nf, mf);
#define SA_FPSDUAL(P) if (ahead) hipLaunchKernelGGL((fps_dual_kernel<P, P, true>), dim3(b, 2), dim3(kBlock), 0, stream, F, D); \
                      else hipLaunchKernelGGL((fps_dual_kernel<P, P, false>), dim3(b, 2), dim3(kBlock), 0, stream, F, D)
    switch (pf) {
        case 1: SA_FPSDUAL(1); break;
        case 2: SA_FPSDUAL(2); break;
        default: SA_FPSDUAL(4); break;
    }
#undef SA_FPSDUAL
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// Force the generic (global-temp) kernels: used by tests to cover the n > 16384 path at small n.
extern "C" int sa_fps_generic(int b, int n, int c, int m, const float *inp, float *temp, int *out,
                              int mode, hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || !inp || !temp || !out) return SA_ERR_INVALID;
    if (mode == 0)
        hipLaunchKernelGGL(fps_generic_kernel<0>, dim3(b), dim3(kBlock), 0, stream, n, c, m, inp, temp, out, m, 0);
    else
        hipLaunchKernelGGL(fps_generic_kernel<1>, dim3(b), dim3(kBlock), 0, stream, n, 0, m, inp, temp, out, m, 0);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// Reference launcher signatures (lib/utils/tf_ops/sampling/tf_sampling.cpp:131,164) + stream.
extern "C" int sa_farthest_point_sample(int b, int n, int c, int m, const float *inp, float *temp,
                                        int *out, hipStream_t stream) {
    return sa_fps_ex(b, n, c, m, inp, temp, out, m, 0, stream);
}
extern "C" int sa_farthest_point_sample_with_distance(int b, int n, int m, const float *dist,
                                                      float *temp, int *out, hipStream_t stream) {
    return sa_fps_with_distance_ex(b, n, m, dist, temp, out, m, 0, stream);
}

// farthestpointsamplingwithpreidxLauncher(b,n,c,m,m1,inp,preidx,temp,out) -- tf_sampling.cpp:195
extern "C" int sa_farthest_point_sample_with_preidx(int b, int n, int c, int m, int m1, const float *inp,
                                                    const int *preidx, float *temp, int *out, hipStream_t stream) {
    if (b <= 0 || n <= 0 || c <= 0 || m <= 0 || m1 < 0 || !inp || (m1 > 0 && !preidx) || !temp || !out)
        return SA_ERR_INVALID;
    hipLaunchKernelGGL(fps_preidx_kernel, dim3(b), dim3(kBlock), 0, stream, n, c, m, m1, inp, preidx, temp, out);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

#ifdef SA_FFPS_SPEC_STATS
extern "C" int sa_debug_ffps_spec(unsigned long long *host4, int reset) {
    if (host4 && hipMemcpyFromSymbol(host4, HIP_SYMBOL(g_ffps_spec), sizeof(g_ffps_spec)) != hipSuccess) return SA_ERR_LAUNCH;
    if (reset) {
        void *d = nullptr;
        if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_ffps_spec)) != hipSuccess || hipMemset(d, 0, sizeof(g_ffps_spec)) != hipSuccess) return SA_ERR_LAUNCH;
    }
    return SA_OK;
}
#endif
