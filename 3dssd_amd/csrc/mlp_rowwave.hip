// "Row-wave" fused grouped MLP for the narrow / mid SA scales (layer1 and layer2 of 3dssd.yaml, the
// configs[0] layer): gather -> concat[features, rel-xyz] -> 3 x (1x1 conv + folded BN + ReLU) -> max over
// nsample -> empty-ball mask, like group_mlp_max_kernel in mlp.hip (same arithmetic: split bf16, 3 MFMA
// passes, fp32 accumulate), but organised around what bounds those scales on MI355X:
//   * ALL weights of the scale live in LDS for the lifetime of a persistent workgroup (6 .. 92 KiB, fragment
//     order, copied once) -- mlp.hip streams every weight fragment from L2 for every 32-row tile;
//   * a wave owns a 32-row tile from gather to pooled output and the activations never leave its registers:
//     the gather loads land directly in MFMA B-operand order (lane = (row, k-half) reads 8 consecutive
//     channels of its row), and the D^T accumulator of a hidden layer becomes the next layer's B operand with
//     four v_permlane32_swap per 16 channels (lane halves hold channels {0-3,8-11} / {4-7,12-15} of a
//     16-channel group; swapping the second quad of the lower half with the first quad of the upper half
//     leaves {0-7} / {8-15}) -- no LDS round trip, no workgroup barrier inside the tile loop;
//   * fp32 -> bf16 hi/lo splitting uses the gfx950 converter (v_cvt_pk_bf16_f32, 6 VALU instructions per
//     pair instead of 26 for the bit-twiddled form): these scales are VALU-issue bound, not MFMA bound;
//   * the index -> point -> feature load chain of tile q+1 (and the index load of tile q+2) is issued before
//     the matrix work of tile q.
// Instantiated for the padded shapes of the reference configuration; any other shape takes mlp.hip.
#include <stdlib.h>

#include "sa_common.h"
#include "mlp_plan.h"
#include "mlp_act.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct RwParams {
    const float *xyz, *feat, *new_xyz;
    const int *idx, *cnt;
    float *out;
    const uint4 *w[3];
    const float *bias[3];
    int n, m, ns, C;
    long nballs;
    int out_stride, out_off;
    int m_shift;   // log2(m) when m is a power of two, else -1
    float inv_m;
    int N3;        // true output channels of the last layer
    const int *gran;   // row plan (mlp_plan.h): granule entries, 4 per 32-row tile
    const int *hdr;    // hdr[0] = number of granules (written by mlp_plan_kernel earlier on the stream)
    int *ovf;          // fp16 form: raised when a converted activation left the fp16 range (mlp_act.h); may be null
};

__device__ __forceinline__ f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 mfma_f16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// x ~= hi + lo, two values per v_cvt_pk_bf16_f32 (round to nearest even); residuals stay scalar so that no
// aligned register pairs are forced on the allocator
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float a, float b, unsigned &hi_pk, unsigned &lo_pk) {
    const f32x2 v = {a, b};
    hi_pk = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const float ra = a - __uint_as_float(hi_pk << 16);
    const float rb = b - __uint_as_float(hi_pk & 0xFFFF0000u);
    const f32x2 r = {ra, rb};
    lo_pk = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}
__device__ __forceinline__ void split8(const float (&v)[8], uint4 &hi, uint4 &lo) {
    split2(v[0], v[1], hi.x, lo.x);
    split2(v[2], v[3], hi.y, lo.y);
    split2(v[4], v[5], hi.z, lo.z);
    split2(v[6], v[7], hi.w, lo.w);
}
// 8 fp32 -> the operand planes of precision PR (lo is untouched for PR == 1).  The fp16 operand form (PR == 1, see
// mlp.hip "Operand precision") is one plane, v_cvt_pk_f16_f32 (nearest even); `det` is the range guard of mlp_act.h.
template <int PR>
__device__ __forceinline__ void to_planes(const float (&v)[8], uint4 &hi, uint4 &lo, sa::f16_guard_t &det) {
    if (PR == 3) split8(v, hi, lo);
    else {
        hi = make_uint4(sa::cvt2_f16(v[0], v[1]), sa::cvt2_f16(v[2], v[3]), sa::cvt2_f16(v[4], v[5]), sa::cvt2_f16(v[6], v[7]));
        sa::f16_guard_signed(hi, det);
    }
}

#ifdef SA_RW_TIMING
// debug build only (tools/rw_phase_prof.py): per-wave phase clocks, one row of 8 per wave, summed on the host
__device__ unsigned long long g_rw_prof[65536 * 8];
#define RW_T0() unsigned long long t__ = __builtin_readcyclecounter(), acc__[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define RW_TICK(i) { const unsigned long long n__ = __builtin_readcyclecounter(); acc__[i] += n__ - t__; t__ = n__; }
#define RW_COUNT() acc__[7]++
#define RW_FLUSH(wave) if (lane == 0) { unsigned long long *r__ = g_rw_prof + (size_t)((wave) & 65535) * 8; for (int i__ = 0; i__ < 8; ++i__) r__[i__] += acc__[i__]; }
#else
#define RW_T0()
#define RW_TICK(i)
#define RW_COUNT()
#define RW_FLUSH(wave)
#endif

struct RowRef { int pt, ball, cnt, ent; };   // source point (flat index), ball, its count and the plan entry of this lane's row

// All element offsets on this path fit 32 bits (checked on the host), so every access is base pointer (SGPR
// pair) + 32-bit lane offset: no 64-bit address arithmetic on the VALU.
// tile, row of the tile -> plan entry (granule row>>3 of the tile) -> ball, sample -> source point; the index and
// count loads issue together.  Granules past the end of the plan (ent < 0) read ball 0 and are never written.
// The chain is TWO dependent round trips (plan entry -> index / count).  In the tile loops the plan entry is therefore
// requested one tile earlier than the index (load_plan_ent, then row_ref_of on the next iteration): requested together,
// the index load waited a full memory round trip for the entry at every tile -- 20-40 % of a wave's time in the narrow
// scales (tools/rw_phase_prof.py).
// GR: rows per granule of the plan (8, or 4 since round 5: mlp_plan.h); a tile holds 32 / GR entries.
template <int GR>
__device__ __forceinline__ int load_plan_ent(const RwParams &P, int ngran, int tile, int row) {
    return sa::plan_entry(P.gran, ngran, tile * (32 / GR) + row / GR);
}
template <int GR>
__device__ __forceinline__ RowRef row_ref_of(const RwParams &P, int ent, int row) {
    const int ball = ent >= 0 ? sa::plan_ball(ent) : 0;
    const int s = sa::plan_sample<GR>(ent, row & (GR - 1), P.ns);
    const int a_raw = P.idx[(unsigned)(ball * P.ns + s)];
    const int c = P.cnt[(unsigned)ball];
    int frame;
    if (P.m_shift >= 0) frame = ball >> P.m_shift;
    else {                                                  // ball < 2^24: one float estimate, exact after +-1
        frame = (int)((float)ball * P.inv_m);
        const int r = ball - frame * P.m;
        frame += r >= P.m ? 1 : (r < 0 ? -1 : 0);
    }
    RowRef r;
    r.cnt = c;
    r.ent = ent;
    r.ball = ball;
    r.pt = frame * P.n + (c > 0 ? a_raw : 0);               // layers_util.py:157-159
    return r;
}
template <int GR>
__device__ __forceinline__ RowRef load_row_ref(const RwParams &P, int ngran, int tile, int row) {
    return row_ref_of<GR>(P, load_plan_ent<GR>(P, ngran, tile, row), row);
}
// the tile's plan entries and ball counts, wave-uniform (rows 0, GR, 2 GR, ...), and the pooled write of one column tile
template <int GR>
__device__ __forceinline__ void tile_entries(const RowRef &cur, int (&ent)[32 / GR], int (&cn)[32 / GR]) {
#pragma unroll
    for (int g = 0; g < 32 / GR; ++g) {
        ent[g] = __builtin_amdgcn_readlane(cur.ent, GR * g);
        cn[g] = __builtin_amdgcn_readlane(cur.cnt, GR * g);
    }
}
template <int GR>
__device__ __forceinline__ void pool_tile(const f32x16 &acc, const int (&ent)[32 / GR], const int (&cn)[32 / GR], float bias_c,
                                          int c, const RwParams &P, int lane) {
    float qm[32 / GR];
    if constexpr (GR == 8) sa::granule_max(acc, qm);
    else if constexpr (GR == 4) sa::granule_max4(acc, qm);
    else sa::granule_max2(acc, qm);
    sa::pool_write_tile<32 / GR>(qm, ent, cn, bias_c, c, P.N3, P.out, P.out_stride, P.out_off, lane);
}

// Relative coordinates (and, for C == 1, the single feature channel) of this lane's row.
struct RowTail { float t[4]; };
template <int TAILF>
__device__ __forceinline__ RowTail load_row_tail(const RwParams &P, const RowRef &rr) {
    RowTail r;
    const unsigned po = (unsigned)rr.pt * 3u, bo = (unsigned)rr.ball * 3u;
    const float px = P.xyz[po + 0] - P.new_xyz[bo + 0];
    const float py = P.xyz[po + 1] - P.new_xyz[bo + 1];
    const float pz = P.xyz[po + 2] - P.new_xyz[bo + 2];
    if (TAILF) { r.t[0] = P.feat[(unsigned)(rr.pt * P.C + (P.C - 1))]; r.t[1] = px; r.t[2] = py; r.t[3] = pz; }
    else { r.t[0] = px; r.t[1] = py; r.t[2] = pz; r.t[3] = 0.0f; }
    return r;
}

// channels [8g, 8g+8) of the grouped row: features first, then xyz - centre, then zero padding
// (layers_util.py:160-165).  The row-wave path takes C = 8*GF + TAILF with TAILF = 0, or C == 1: group g is a
// full feature group (g < GF), the tail group (g == GF: [last feature,] dx, dy, dz, 0...), or zeros.
__device__ __forceinline__ void load_group(const RwParams &P, const RowRef &rr, const RowTail &tl, int g,
                                           float (&v)[8]) {
    const int GF = P.C >> 3;
    if (g < GF) {
        const unsigned fo = (unsigned)(rr.pt * P.C + 8 * g);
        const float4 f0 = *(const float4 *)(P.feat + fo);
        const float4 f1 = *(const float4 *)(P.feat + fo + 4);
        v[0] = f0.x; v[1] = f0.y; v[2] = f0.z; v[3] = f0.w;
        v[4] = f1.x; v[5] = f1.y; v[6] = f1.z; v[7] = f1.w;
    } else {
        const bool is = g == GF;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = is ? tl.t[e] : 0.0f;
#pragma unroll
        for (int e = 4; e < 8; ++e) v[e] = 0.0f;
    }
}

// acc (D^T form: reg r of lane (row, h) = channel (r&3) + 8*(r>>2) + 4h of the tile) -> ReLU -> the two
// B-operand fragments (k-steps 2*ct and 2*ct+1 of the next layer) of this lane
template <int KSN, int PR = 3>
__device__ __forceinline__ void acc_to_frags(const f32x16 &acc, int ct, uint4 (&fh)[KSN], uint4 (&fl)[KSN], sa::f16_guard_t &det) {
#pragma unroll
    for (int hk = 0; hk < 2; ++hk) {
        if (2 * ct + hk < KSN) {
            if (PR == 3) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = sa::fmax_nn(acc[8 * hk + r], 0.0f);
                    const float b = sa::fmax_nn(acc[8 * hk + 4 + r], 0.0f);
                    // upper-half lanes of `a` <-> lower-half lanes of `b`
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
                    v[r] = __uint_as_float(sw[0]);
                    v[4 + r] = __uint_as_float(sw[1]);
                }
                split8(v, fh[2 * ct + hk], fl[2 * ct + hk]);
            } else {
                // fp16: convert (+ packed ReLU + range guard, mlp_act.h) first, then swap the PACKED pairs -- half the
                // cross-half moves
                unsigned pa[2], pb[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    pa[j] = sa::cvt2_f16_relu(acc[8 * hk + 2 * j], acc[8 * hk + 2 * j + 1]);
                    pb[j] = sa::cvt2_f16_relu(acc[8 * hk + 4 + 2 * j], acc[8 * hk + 4 + 2 * j + 1]);
                }
                uint4 f;
                {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pa[0], pb[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pa[1], pb[1], false, false);
                    f = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
                fh[2 * ct + hk] = f;
                sa::f16_guard(f, det);
            }
        }
    }
}

// hidden layer: KS k-steps of input fragments (ih, il) -> NT output tiles -> next layer's fragments (oh, ol)
template <int KS, int NT, int KSN>
__device__ __forceinline__ void hidden_layer(const uint4 *W, const float *bias, const uint4 (&ih)[KS],
                                             const uint4 (&il)[KS], uint4 (&oh)[KSN], uint4 (&ol)[KSN], int lane) {
    const int half = lane >> 5;
    constexpr int TG = NT >= 2 ? 2 : 1;
#pragma unroll
    for (int ct0 = 0; ct0 < NT; ct0 += TG) {
        f32x16 acc[TG];
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) {
            const int ct = ct0 + tt < NT ? ct0 + tt : NT - 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *(const float4 *)(bias + ct * 32 + 8 * q + 4 * half);
                acc[tt][4 * q + 0] = bv.x; acc[tt][4 * q + 1] = bv.y;
                acc[tt][4 * q + 2] = bv.z; acc[tt][4 * q + 3] = bv.w;
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) {
                if (ct0 + tt < NT) {
                    const uint4 wh = W[((ct0 + tt) * KS + ks) * 128 + lane];
                    const uint4 wl = W[((ct0 + tt) * KS + ks) * 128 + 64 + lane];
                    acc[tt] = mfma_bf16(wh, ih[ks], acc[tt]);
                    acc[tt] = mfma_bf16(wl, ih[ks], acc[tt]);
                    acc[tt] = mfma_bf16(wh, il[ks], acc[tt]);
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < TG; ++tt)
            if (ct0 + tt < NT) { sa::f16_guard_t nodet = 0; acc_to_frags<KSN>(acc[tt], ct0 + tt, oh, ol, nodet); }
    }
}

// KS0: k-steps of the gathered input; (NT1, KS1), (NT2, KS2): output tiles of hidden layer 1 / 2 and the
// k-steps the next layer reads of them; NT3: output tiles of the last layer.  NW waves per workgroup.
template <int KS0, int NT1, int KS1, int NT2, int KS2, int NT3, int NW, int WPE, int TAILF, int GR>
__device__ __forceinline__ void rw_body(const RwParams &P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int nW0 = NT1 * KS0 * 128, nW1 = NT2 * KS1 * 128, nW2 = NT3 * KS2 * 128;   // uint4 counts
    uint4 *W0 = (uint4 *)smem, *W1 = W0 + nW0, *W2 = W1 + nW1;
    float *b0 = (float *)(W2 + nW2), *b1 = b0 + NT1 * 32, *b2 = b1 + NT2 * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ngran = __builtin_amdgcn_readfirstlane(P.hdr[0]);
    const int ntiles = (ngran + 32 / GR - 1) / (32 / GR);
    int gstride;                                         // XCD x takes a contiguous eighth of every pass over the tiles
    const int bx = sa::xcd_block(blockIdx.x, gridDim.x, (ntiles + NW - 1) / NW, gstride);
    if (bx < 0 || bx * NW >= ntiles) return;             // persistent grid sized for the densest plan: no work, no copy
    RW_T0();
    // one-time copy of the packed weights and biases: all loads of a layer are issued before its stores
    {
        constexpr int NT_ = NW * 64;
        static_assert(nW0 <= 8 * NT_ && nW1 <= 8 * NT_ && nW2 <= 8 * NT_, "weight copy batch too small");
#define SA_RW_COPY(SRC, DST, CNT)                                                                         \
        {                                                                                                 \
            uint4 tmp[8];                                                                                 \
            _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                 \
                if (j * NT_ < (CNT)) { const int i = j * NT_ + tid; tmp[j] = (SRC)[i < (CNT) ? i : (CNT) - 1]; } \
            _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                 \
                if (j * NT_ < (CNT)) { const int i = j * NT_ + tid; if (i < (CNT)) (DST)[i] = tmp[j]; }   \
        }
        SA_RW_COPY(P.w[0], W0, nW0)
        SA_RW_COPY(P.w[1], W1, nW1)
        SA_RW_COPY(P.w[2], W2, nW2)
#undef SA_RW_COPY
        if (tid < NT1 * 32) b0[tid] = P.bias[0][tid];
        if (tid < NT2 * 32) b1[tid] = P.bias[1][tid];
        if (tid < NT3 * 32) b2[tid] = P.bias[2][tid];
        static_assert(NT1 * 32 <= NT_ && NT2 * 32 <= NT_ && NT3 * 32 <= NT_, "bias copy needs one pass");
    }
    __syncthreads();
    RW_TICK(0)

    // (A wave's own MFMA and VALU instructions do not overlap on gfx950, those of two waves on one SIMD can --
    //  tools/microbench/mfma_valu_overlap.hip.  Neither unequal s_setprio priorities nor staggered wave starts
    //  changed the measured time of these kernels: it tracks MFMA-busy + VALU-issue cycles.)
    const int row = lane & 31, half = lane >> 5;
    const int gw = bx * NW + w, nwaves = gstride * NW;
    if (gw >= ntiles) return;
    // this wave's tiles: gw, gw + nwaves, ...  Cursors of the three pipeline stages: compute (tc), feature loads (one
    // tile ahead), index loads (two ahead); past the end they stay on the last tile.
    auto advance = [&](int &t) { if (t + nwaves < ntiles) t += nwaves; };
    int tc = gw, tf = gw;
    advance(tf);
    int ti = tf;
    advance(ti);
    int te = ti;                                         // plan entries: three ahead
    advance(te);

    float raw[KS0][8];
    RowRef cur = load_row_ref<GR>(P, ngran, tc, row);
    {
        const RowTail tl = load_row_tail<TAILF>(P, cur);
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) load_group(P, cur, tl, 2 * ks + half, raw[ks]);
    }
    RowRef nxt = load_row_ref<GR>(P, ngran, tf, row);
    int ent_i = load_plan_ent<GR>(P, ngran, ti, row);

    for (bool more = true; more;) {
        more = tc + nwaves < ntiles;
        // ---- this tile's input as B-operand fragments
        uint4 h0[KS0], l0[KS0];
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) split8(raw[ks], h0[ks], l0[ks]);
        // the tile's plan entries and ball counts, wave-uniform (rows 0, GR, 2 GR, ...)
        int ent[32 / GR], cn[32 / GR];
        tile_entries<GR>(cur, ent, cn);
        RW_COUNT();
        RW_TICK(1)
        // ---- loads of the next tile (features) and of the one after (indices), then pin them above the math
        cur = nxt;
        {
            const RowTail tl = load_row_tail<TAILF>(P, cur);
#pragma unroll
            for (int ks = 0; ks < KS0; ++ks) load_group(P, cur, tl, 2 * ks + half, raw[ks]);
        }
        nxt = row_ref_of<GR>(P, ent_i, row);             // tile ti: its entry arrived during the previous tile
        ent_i = load_plan_ent<GR>(P, ngran, te, row);
        advance(tc);
        advance(tf);
        advance(ti);
        advance(te);
        __builtin_amdgcn_sched_barrier(0);
        RW_TICK(2)

        uint4 h1[KS1], l1[KS1];
        hidden_layer<KS0, NT1, KS1>(W0, b0, h0, l0, h1, l1, lane);
        RW_TICK(3)
        uint4 h2[KS2], l2[KS2];
        hidden_layer<KS1, NT2, KS2>(W1, b1, h1, l1, h2, l2, lane);
        RW_TICK(4)
        // ---- last layer (D form), max over the rows of each granule, relu(max + bias) written per ball run
        //      (layers_util.py:178-181)
        constexpr int TG = NT3 >= 2 ? 2 : 1;
#pragma unroll
        for (int ct0 = 0; ct0 < NT3; ct0 += TG) {
            f32x16 acc[TG];
#pragma unroll
            for (int tt = 0; tt < TG; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tt][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
#pragma unroll
                for (int tt = 0; tt < TG; ++tt) {
                    if (ct0 + tt < NT3) {
                        const uint4 wh = W2[((ct0 + tt) * KS2 + ks) * 128 + lane];
                        const uint4 wl = W2[((ct0 + tt) * KS2 + ks) * 128 + 64 + lane];
                        acc[tt] = mfma_bf16(h2[ks], wh, acc[tt]);
                        acc[tt] = mfma_bf16(h2[ks], wl, acc[tt]);
                        acc[tt] = mfma_bf16(l2[ks], wh, acc[tt]);
                    }
                }
            }
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) {
                if (ct0 + tt < NT3) {
                    const int c = (ct0 + tt) * 32 + (lane & 31);
                    pool_tile<GR>(acc[tt], ent, cn, b2[c], c, P, lane);
                }
            }
        }
        RW_TICK(5)
    }
    RW_FLUSH(gw)
}
template <int KS0, int NT1, int KS1, int NT2, int KS2, int NT3, int NW, int WPE, int TAILF, int GR>
__global__ __launch_bounds__(NW * 64, WPE) void mlp_rw_kernel(RwParams P) {
    rw_body<KS0, NT1, KS1, NT2, KS2, NT3, NW, WPE, TAILF, GR>(P);
}

// =====================================================================================================
// Streamed-weight variant for the mid-width scales (layer3 of 3dssd.yaml: 131 -> 128 -> 128..256 -> 256) whose
// packed weights (264 .. 456 KiB) do not fit LDS.  Same wave-owns-a-tile / activations-in-registers structure,
// but the NW waves of a workgroup walk the weight stream together: it is cut into 12 chunks of G k-step tiles
// (2 KiB each, in exactly the order the unrolled MFMA loops consume them == the packed global order), two chunk
// slots live in LDS, and at every chunk boundary (a compile-time position in the unrolled code)
//     barrier -> store the staged registers of chunk c+1 into the slot chunk c-1 just vacated
//             -> issue the global loads of chunk c+1+DEPTH into that staging set -> compute on chunk c.
// One barrier per chunk, every weight byte crosses L2 -> LDS once per NW*32 rows (mlp.hip: once per 32 rows),
// and the loads of a chunk have DEPTH chunks of matrix work (~G*96 cycles per wave each) to land.
// Eight waves per workgroup (two per SIMD): a wave's own MFMA and VALU work serialise on gfx950, the other
// wave of the SIMD fills the gaps; that halves the register budget (256), hence one accumulator chain and the
// next tile's loads issued after the tile instead of under its last layer.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector: plain loads / stores, no memcpy
#ifndef SA_RS_LA
#define SA_RS_LA 4
#endif
constexpr int kRsLA16 = SA_RS_LA;     // fp16 form: weight fragments requested from LDS ahead of the MFMA that uses them
struct RsCtx {
    uint4 *ring;          // 2 slots x PC x 64 uint4
    int w, lane;
    int abase;            // uint4 index of this lane's fragment slot in the current chunk
    u32x4 qh[kRsLA16 > 4 ? kRsLA16 : 4], ql[kRsLA16 > 4 ? kRsLA16 : 4];   // weight fragments (hi, lo plane) of the next LA k-step tiles, already requested from LDS
};

// The stream of a pass is cut into CPP chunks of G = TOT/CPP k-step tiles (PC = NP*G pieces of 1 KiB, NP = planes per
// tile: 2 for split bf16, 1 for fp16; wave w moves pieces w, w+NW, ...).  DEPTH staging sets: chunk k travels in set
// k % DEPTH, its loads are issued DEPTH boundaries before the boundary that stores it into LDS (CPP % DEPTH == 0 keeps
// every index a compile-time constant across passes), i.e. they have DEPTH chunks of matrix work to land.
constexpr int kRsCPP = 12;

// global loads of stream chunk sc (0 .. CPP-1) into staging set `st`.  The three layers of the scale are packed
// back to back in ONE device buffer (checked on the host): the stream is a plain linear array of pieces.
template <int PC, int NW, int PPW>
__device__ __forceinline__ void rs_issue_chunk(const RwParams &P, RsCtx &X, u32x4 (&st)[PPW], int sc) {
    int wl = X.w;
    asm volatile("" : "+s"(wl));        // opaque: keeps the per-chunk addresses from being hoisted out of the pass loop
    const unsigned off = (unsigned)((sc * PC + wl) * 64 + X.lane);
#pragma unroll
    for (int i = 0; i < PPW; ++i)
        if (NW * i + NW <= PC || wl + NW * i < PC)          // ragged last round: wave-uniform
            st[i] = *(const u32x4 *)(P.w[0] + (off + (unsigned)(NW * i * 64)));
}
template <int PC, int NW, int PPW>
__device__ __forceinline__ void rs_store_stage(RsCtx &X, const u32x4 (&st)[PPW], int slot) {
#pragma unroll
    for (int i = 0; i < PPW; ++i)
        if (NW * i + NW <= PC || X.w + NW * i < PC)
            *(u32x4 *)(X.ring + ((slot * PC + X.w + NW * i) * 64 + X.lane)) = st[i];
}
// chunk boundary in front of stream position p (p % G == 0)
template <int G, int NP, int NW, int PPW, int DEPTH, int CPP>
__device__ __forceinline__ void rs_boundary(const RwParams &P, RsCtx &X, u32x4 (&st)[DEPTH][PPW], int p) {
    constexpr int PC = NP * G;
    const int cidx = p / G;
    __builtin_amdgcn_sched_barrier(0);      // the chunk's loads must not drift up across earlier boundaries
    __syncthreads();
    rs_store_stage<PC, NW, PPW>(X, st[(cidx + 1) % DEPTH], (cidx + 1) & 1);
    rs_issue_chunk<PC, NW, PPW>(P, X, st[(cidx + 1) % DEPTH], (cidx + 1 + DEPTH) % CPP);
    __builtin_amdgcn_sched_barrier(0);
    X.abase = (cidx & 1) * PC * 64 + X.lane;
}
// one output tile of a layer: k-step tiles base .. base+KS-1 of the stream.  One accumulator chain: two waves
// share a SIMD in this kernel, their chains interleave on the matrix pipe, and the register budget (256) has no
// room for a second accumulator.  The weight fragments of the next LA k-step tiles are requested from LDS before
// the current tile's MFMAs are issued: a fragment feeds three passes (96 cycles) in the split-bf16 form, so one
// tile of lookahead covers the LDS round trip there, but only one 32-cycle pass in the fp16 form -- LA = 4.
template <int KS, int G, int PR, int NW, int PPW, int DEPTH, int CPP, bool WFIRST>
__device__ __forceinline__ void rs_tile_mma(const RwParams &P, RsCtx &X, u32x4 (&st)[DEPTH][PPW],
                                            const uint4 (&ih)[KS], const uint4 (&il)[KS], int base,
                                            f32x16 &acc) {
    constexpr int NP = PR == 3 ? 2 : 1;
    constexpr int LA = PR == 3 ? 1 : kRsLA16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int p = base + ks, pc = p % G;
        if (pc == 0) {
            rs_boundary<G, NP, NW, PPW, DEPTH, CPP>(P, X, st, p);
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                if (j < G) {
                    X.qh[j] = *(const u32x4 *)(X.ring + (X.abase + (j * NP) * 64));
                    if (PR == 3) X.ql[j] = *(const u32x4 *)(X.ring + (X.abase + (j * NP + 1) * 64));
                }
            }
        }
        const u32x4 ch = X.qh[pc % LA], cl = X.ql[pc % LA];
        // (the first LA tiles of a chunk cannot be requested before its barrier)
        if (pc + LA < G) {
            X.qh[pc % LA] = *(const u32x4 *)(X.ring + (X.abase + ((pc + LA) * NP) * 64));
            if (PR == 3) X.ql[pc % LA] = *(const u32x4 *)(X.ring + (X.abase + ((pc + LA) * NP + 1) * 64));
            __builtin_amdgcn_sched_barrier(0);
        }
        const uint4 wh = __builtin_bit_cast(uint4, ch), wl = __builtin_bit_cast(uint4, cl);
        if (PR == 3) {
            if (WFIRST) {
                acc = mfma_bf16(wh, ih[ks], acc);
                acc = mfma_bf16(wl, ih[ks], acc);
                acc = mfma_bf16(wh, il[ks], acc);
            } else {
                acc = mfma_bf16(ih[ks], wh, acc);
                acc = mfma_bf16(ih[ks], wl, acc);
                acc = mfma_bf16(il[ks], wh, acc);
            }
        } else {
            acc = WFIRST ? mfma_f16(wh, ih[ks], acc) : mfma_f16(ih[ks], wh, acc);
        }
    }
}
__device__ __forceinline__ void load_bias_tile(const float *bias, int ct, int half, f32x16 &a) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        const float4 bv = *(const float4 *)(bias + ct * 32 + 8 * qd + 4 * half);
        a[4 * qd + 0] = bv.x; a[4 * qd + 1] = bv.y; a[4 * qd + 2] = bv.z; a[4 * qd + 3] = bv.w;
    }
}

// PR: operand precision (3 = split bf16, 1 = fp16, see mlp.hip); CPP: chunks per pass (a divisor of the k-step tiles
// of the scale); PF: 1 = the next tile's rows are requested right after this tile's last layer (two or more tiles per
// wave), 0 = at the top of the pass (one tile per wave: nothing to prefetch, 17 x 8 registers less).
template <int KS0, int NT1, int KS1, int NT2, int KS2, int NT3, int NW, int WPE, int TAILF, int DEPTH, int PR, int CPP, int PF, int GR>
__device__ __forceinline__ void rs_body(const RwParams &P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KT0 = NT1 * KS0, KT1 = NT2 * KS1, KT2 = NT3 * KS2, TOT = KT0 + KT1 + KT2;
    static_assert(TOT % CPP == 0 && CPP % DEPTH == 0, "CPP chunks per pass, staging depth a divisor of CPP");
    constexpr int NP = PR == 3 ? 2 : 1;      // planes (1 KiB pieces) per k-step tile
    constexpr int G = TOT / CPP;             // k-step tiles per chunk
    constexpr int PC = NP * G;               // 1 KiB pieces per chunk
    constexpr int PPW = (PC + NW - 1) / NW;  // pieces a wave moves per chunk (last round may be ragged)
    RsCtx X;
    u32x4 stage[DEPTH][PPW];                 // chunk k waits in stage[k % DEPTH]
    X.ring = (uint4 *)smem;
    float *b0 = (float *)(X.ring + 2 * PC * 64), *b1 = b0 + NT1 * 32, *b2 = b1 + NT2 * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 31, half = lane >> 5;
    X.w = w; X.lane = lane; X.abase = 0;

    const int ngran = __builtin_amdgcn_readfirstlane(P.hdr[0]);
    const int ntiles = (ngran + 32 / GR - 1) / (32 / GR);
    int gstride;                                         // XCD x takes a contiguous eighth of every pass over the tiles
    const int bx = sa::xcd_block(blockIdx.x, gridDim.x, (ntiles + NW - 1) / NW, gstride);
    if (bx < 0 || bx * NW >= ntiles) return;             // persistent grid sized for the densest plan

    for (int i = tid; i < NT1 * 32; i += NW * 64) b0[i] = P.bias[0][i];
    for (int i = tid; i < NT2 * 32; i += NW * 64) b1[i] = P.bias[1][i];
    for (int i = tid; i < NT3 * 32; i += NW * 64) b2[i] = P.bias[2][i];
    RW_T0();

    // every wave of a workgroup runs the same number of tiles (barriers inside); tiles past the end resolve to
    // invalid plan entries (ball 0 is read, nothing is written)
    const int nwaves = gstride * NW, gw = bx * NW + w;
    const int npass = (ntiles - bx * NW + nwaves - 1) / nwaves;
    int tc = gw, tf = gw + nwaves;

    // ---- prologue: chunk 0 into slot 0, chunks 1 .. DEPTH staged; first tile's rows and features
    rs_issue_chunk<PC, NW, PPW>(P, X, stage[0], 0);
    RowRef cur = load_row_ref<GR>(P, ngran, tc, row);
    rs_store_stage<PC, NW, PPW>(X, stage[0], 0);
#pragma unroll
    for (int k = 1; k <= DEPTH; ++k) rs_issue_chunk<PC, NW, PPW>(P, X, stage[k % DEPTH], k % CPP);
    float raw[KS0][8];
    if (PF) {
        const RowTail tl = load_row_tail<TAILF>(P, cur);
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) load_group(P, cur, tl, 2 * ks + half, raw[ks]);
    }
    RowRef nxt = load_row_ref<GR>(P, ngran, tf, row);
    int ent_f = load_plan_ent<GR>(P, ngran, tf + nwaves, row);  // the entry of the tile after `nxt`: one tile ahead of its index
    sa::f16_guard_t det = 0;                 // fp16 range guard (mlp_act.h), scalar registers

    RW_TICK(0)
    for (int q = 0; q < npass; ++q) {
        RW_COUNT();
        if (!PF) {
            const RowTail tl = load_row_tail<TAILF>(P, cur);
#pragma unroll
            for (int ks = 0; ks < KS0; ++ks) load_group(P, cur, tl, 2 * ks + half, raw[ks]);
        }
        uint4 h0[KS0], l0[KS0];
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) to_planes<PR>(raw[ks], h0[ks], l0[ks], det);
        int ent[32 / GR], cn[32 / GR];           // the tile's plan entries / ball counts, wave-uniform
        tile_entries<GR>(cur, ent, cn);
        RW_TICK(1)

        // ---- hidden layer 0
        uint4 h1[KS1], l1[KS1];
#pragma unroll
        for (int ct = 0; ct < NT1; ++ct) {
            f32x16 ae;
            load_bias_tile(b0, ct, half, ae);
            rs_tile_mma<KS0, G, PR, NW, PPW, DEPTH, CPP, true>(P, X, stage, h0, l0, ct * KS0, ae);
            acc_to_frags<KS1, PR>(ae, ct, h1, l1, det);
        }
        RW_TICK(2)
        // ---- hidden layer 1
        uint4 h2[KS2], l2[KS2];
#pragma unroll
        for (int ct = 0; ct < NT2; ++ct) {
            f32x16 ae;
            load_bias_tile(b1, ct, half, ae);
            rs_tile_mma<KS1, G, PR, NW, PPW, DEPTH, CPP, true>(P, X, stage, h1, l1, KT0 + ct * KS1, ae);
            acc_to_frags<KS2, PR>(ae, ct, h2, l2, det);
        }
        RW_TICK(3)
        // ---- last layer (D form), granule maxima, relu(max + bias) written per ball run (layers_util.py:178-181)
#pragma unroll
        for (int ct = 0; ct < NT3; ++ct) {
            f32x16 ae;
#pragma unroll
            for (int r = 0; r < 16; ++r) ae[r] = 0.0f;
            rs_tile_mma<KS2, G, PR, NW, PPW, DEPTH, CPP, false>(P, X, stage, h2, l2, KT0 + KT1 + ct * KS2, ae);
            const int c = ct * 32 + (lane & 31);
            pool_tile<GR>(ae, ent, cn, b2[c], c, P, lane);
        }
        RW_TICK(4)
        // ---- the next tile's rows: issued here, converted at the top of the next iteration (the other wave of
        //      the SIMD works under their latency; during the last layer the register budget has no room for them)
        tc += nwaves;
        tf += nwaves;
        cur = nxt;
        if (PF) {
            const RowTail tl = load_row_tail<TAILF>(P, cur);
#pragma unroll
            for (int ks = 0; ks < KS0; ++ks) load_group(P, cur, tl, 2 * ks + half, raw[ks]);
        }
        nxt = row_ref_of<GR>(P, ent_f, row);
        ent_f = load_plan_ent<GR>(P, ngran, tf + nwaves, row);
        RW_TICK(5)
    }
    if (PR == 1) sa::f16_overflow_report(det, P.ovf, lane);
    RW_FLUSH(gw)
}
template <int KS0, int NT1, int KS1, int NT2, int KS2, int NT3, int NW, int WPE, int TAILF, int DEPTH, int PR, int CPP, int PF, int GR>
__global__ __launch_bounds__(NW * 64, WPE) void mlp_rs_kernel(RwParams P) {
    rs_body<KS0, NT1, KS1, NT2, KS2, NT3, NW, WPE, TAILF, DEPTH, PR, CPP, PF, GR>(P);
}

// ---- all scales of an SA layer in ONE launch: blockIdx.y picks the scale (its own parameters, its own shape).  The
//      scales of a layer are independent (same inputs, disjoint output slices), a launch costs ~2 us of throughput and
//      these kernels run one or two tiles per wave, so three launches of 14-40 us become one of the longest's length.
struct RwMulti { RwParams p[3]; };
template <int KS0, int NT1, int KS1, int NT2, int KS2, int NT3, int NW, int WPE, int TAILF, int GR>
struct RwBody {
    static constexpr size_t lds = (size_t)(NT1 * KS0 + NT2 * KS1 + NT3 * KS2) * 2048 + (size_t)(NT1 + NT2 + NT3) * 128;
    static __device__ __forceinline__ void run(const RwParams &P) { rw_body<KS0, NT1, KS1, NT2, KS2, NT3, NW, WPE, TAILF, GR>(P); }
};
template <int KS0, int NT1, int KS1, int NT2, int KS2, int NT3, int NW, int WPE, int DEPTH, int PR, int CPP, int PF, int GR>
struct RsBody {
    static constexpr size_t lds = (size_t)2 * (PR == 3 ? 2 : 1) * ((NT1 * KS0 + NT2 * KS1 + NT3 * KS2) / CPP) * 1024 +
                                  (size_t)(NT1 + NT2 + NT3) * 128;
    static __device__ __forceinline__ void run(const RwParams &P) { rs_body<KS0, NT1, KS1, NT2, KS2, NT3, NW, WPE, 0, DEPTH, PR, CPP, PF, GR>(P); }
};
template <class B0, class B1, class B2, int NW, int WPE>
__global__ __launch_bounds__(NW * 64, WPE) void mlp_multi_kernel(RwMulti M) {
    if (blockIdx.y == 0) B0::run(M.p[0]);
    else if (blockIdx.y == 1) B1::run(M.p[1]);
    else B2::run(M.p[2]);
}

int roundup(int x, int q) { return (x + q - 1) / q * q; }

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int KS0, int NT1, int KS1, int NT2, int KS2, int NT3, int NW, int WPE, int TAILF, int GR>
int launch_rw(const RwParams &P, long max_tiles, int wgs_per_cu, hipStream_t stream) {
    constexpr size_t lds = (size_t)(NT1 * KS0 + NT2 * KS1 + NT3 * KS2) * 2048 + (size_t)(NT1 + NT2 + NT3) * 128;
    auto kern = mlp_rw_kernel<KS0, NT1, KS1, NT2, KS2, NT3, NW, WPE, TAILF, GR>;
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipGetLastError();
    }
    long grid = (max_tiles + NW - 1) / NW;      // the densest plan; workgroups without a tile leave at once
    const long cap = (long)num_cus() * wgs_per_cu;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, P);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

template <int KS0, int NT1, int KS1, int NT2, int KS2, int NT3, int NW, int WPE, int TAILF, int DEPTH, int PR, int CPP, int PF, int GR>
int launch_rs(const RwParams &P, long max_tiles, int wgs_per_cu, hipStream_t stream) {
    constexpr int G = (NT1 * KS0 + NT2 * KS1 + NT3 * KS2) / CPP;
    constexpr size_t lds = (size_t)2 * (PR == 3 ? 2 : 1) * G * 1024 + (size_t)(NT1 + NT2 + NT3) * 128;
    auto kern = mlp_rs_kernel<KS0, NT1, KS1, NT2, KS2, NT3, NW, WPE, TAILF, DEPTH, PR, CPP, PF, GR>;
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipGetLastError();
    }
    long grid = (max_tiles + NW - 1) / NW;      // the densest plan; workgroups without a tile leave at once
    const long cap = (long)num_cus() * wgs_per_cu;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, P);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

template <class B0, class B1, class B2, int NW, int WPE>
int launch_multi(const RwParams (&P)[3], const long (&max_tiles)[3], int wgs_per_cu, hipStream_t stream) {
    constexpr size_t l01 = B0::lds > B1::lds ? B0::lds : B1::lds, lds = l01 > B2::lds ? l01 : B2::lds;
    auto kern = mlp_multi_kernel<B0, B1, B2, NW, WPE>;
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipGetLastError();
    }
    long mt = max_tiles[0] > max_tiles[1] ? max_tiles[0] : max_tiles[1];
    if (max_tiles[2] > mt) mt = max_tiles[2];
    long grid = (mt + NW - 1) / NW;             // the densest plan of the widest scale; workgroups without a tile leave at once
    const long cap = (long)num_cus() * wgs_per_cu;
    if (grid > cap) grid = cap;
    RwMulti M;
    for (int i = 0; i < 3; ++i) M.p[i] = P[i];
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, 3), dim3(NW * 64), lds, stream, M);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

struct ScaleSig { int KS0, NT1, KS1, NT2, KS2, NT3; };
bool sig_is(const ScaleSig &s, int a, int b, int c, int d, int e, int f) {
    return s.KS0 == a && s.NT1 == b && s.KS1 == c && s.NT2 == d && s.KS2 == e && s.NT3 == f;
}

}  // namespace

// Returns 1 when the shape has a row-wave instantiation and the launch was issued (status in *st), 0 when the
// caller should take the generic kernel.
// parameters + padded shape of one scale; false when the row-wave kernels cannot take it
static bool rowwave_scale(int b, int n, int m, int ns, int c, const float *xyz, const float *feat, const float *new_xyz,
                          const int *idx, const int *cnt, int nl, const int *dims, const void *const *wpack,
                          const float *const *bias, float *out, int out_stride, int out_off, const int *plan_hdr,
                          const int *plan_gran, long max_tiles, int *overflow, RwParams &P, ScaleSig &S) {
    static const bool enabled = SA_KNOB("SA_MLP_ROWWAVE", 1) != 0;
    if (!enabled || nl != 3) return false;
    if (!(c == 1 || (c > 0 && (c & 7) == 0))) return false;  // input layouts the in-register gather handles
    // 32-bit element offsets everywhere (and a float-exact ball / m)
    const long nb_ = (long)b * m;
    if (nb_ >= (1l << 24) || (long)b * n * (c > 3 ? c : 3) >= (1l << 31) || nb_ * ns >= (1l << 31) ||
        nb_ * out_stride + out_off + dims[3] >= (1l << 31))
        return false;
    S.KS0 = roundup(dims[0], 16) / 16;
    S.NT1 = roundup(dims[1], 32) / 32; S.KS1 = roundup(dims[1], 16) / 16;
    S.NT2 = roundup(dims[2], 32) / 32; S.KS2 = roundup(dims[2], 16) / 16;
    S.NT3 = roundup(dims[3], 32) / 32;
    P = RwParams{};
    P.xyz = xyz; P.feat = feat; P.new_xyz = new_xyz; P.idx = idx; P.cnt = cnt; P.out = out;
    for (int l = 0; l < 3; ++l) { P.w[l] = (const uint4 *)wpack[l]; P.bias[l] = bias[l]; }
    P.n = n; P.m = m; P.ns = ns; P.C = c; P.nballs = (long)b * m;
    P.out_stride = out_stride; P.out_off = out_off;
    P.N3 = dims[3];
    P.hdr = plan_hdr; P.gran = plan_gran; P.ovf = overflow;
    P.m_shift = -1;
    for (int sft = 0; sft < 31; ++sft) if (m == (1 << sft)) P.m_shift = sft;
    P.inv_m = 1.0f / (float)m;
    return max_tiles <= 0x0FFFFFFFl;
}
static bool rowwave_contiguous(const ScaleSig &S, const void *const *wpack, int fp16) {
    const int planes = fp16 ? 1 : 2;         // 1 KiB pieces per (tile, k-step)
    return (const char *)wpack[1] == (const char *)wpack[0] + (size_t)S.NT1 * S.KS0 * planes * 1024 &&
           (const char *)wpack[2] == (const char *)wpack[1] + (size_t)S.NT2 * S.KS1 * planes * 1024;
}

// The three scales of a layer in one launch (mlp_multi_kernel) for the layer shapes of 3dssd.yaml; 0 when the layer
// is not one of them (the caller then launches scale by scale).
// gr[i]: rows per granule of scale i's plan (8, 4 or 2: mlp_plan.h).  Instantiated: all 8, all 4, and per layer the mixed
// combinations that were measured -- layer 1: the inner bands (1-2 points per ball) at 4 or 2 rows, the outer band at 8;
// layer 2: (2, 2, 4); layer 3: scale 0 at 4 or 2.  Any other combination: 0 is returned (the caller goes scale by scale).
int sa_rowwave_try_layer(int b, int n, int m, const int *ns, int c, const float *xyz, const float *feat,
                         const float *new_xyz, const int *const *idx, const int *const *cnt, const int *dims,
                         const void *const *wpack, const float *const *bias, float *out, int out_stride,
                         const int *out_off, const int *const *plan_hdr, const int *const *plan_gran,
                         const long *max_tiles, const int *fp16, const int *gr, int *overflow, hipStream_t stream, int *st) {
    static const bool on = SA_KNOB("SA_MLP_MULTI", 1) != 0;
    static const bool stream_enabled = SA_KNOB("SA_MLP_ROWSTREAM", 1) != 0;
    if (!on) return 0;
    RwParams P[3];
    ScaleSig S[3];
    long mt[3];
    for (int i = 0; i < 3; ++i) {
        if (!rowwave_scale(b, n, m, ns[i], c, xyz, feat, new_xyz, idx[i], cnt[i], 3, dims + 4 * i, wpack + 3 * i,
                           bias + 3 * i, out, out_stride, out_off[i], plan_hdr[i], plan_gran[i], max_tiles[i], overflow, P[i], S[i]))
            return 0;
        if (ns[i] > gr[i] * sa::kPlanMaxOrd) return 0;
        mt[i] = max_tiles[i];
    }
    const bool all16 = fp16[0] && fp16[1] && fp16[2], none16 = !fp16[0] && !fp16[1] && !fp16[2];
    const int gkey = gr[0] * 100 + gr[1] * 10 + gr[2];        // 888, 444, 448, 228, ...
    if (none16 && c == 1 && sig_is(S[0], 1, 1, 1, 1, 1, 1) && sig_is(S[1], 1, 1, 1, 1, 1, 1) && sig_is(S[2], 1, 1, 2, 1, 2, 2)) {
        // layer1: 4 -> 16 -> 16 -> 32 (x2), 4 -> 32 -> 32 -> 64
        if (gkey == 444) *st = launch_multi<RwBody<1, 1, 1, 1, 1, 1, 4, 4, 1, 4>, RwBody<1, 1, 1, 1, 1, 1, 4, 4, 1, 4>, RwBody<1, 1, 2, 1, 2, 2, 4, 4, 1, 4>, 4, 4>(P, mt, 4, stream);
        else if (gkey == 448) *st = launch_multi<RwBody<1, 1, 1, 1, 1, 1, 4, 4, 1, 4>, RwBody<1, 1, 1, 1, 1, 1, 4, 4, 1, 4>, RwBody<1, 1, 2, 1, 2, 2, 4, 4, 1, 8>, 4, 4>(P, mt, 4, stream);
        else if (gkey == 228) *st = launch_multi<RwBody<1, 1, 1, 1, 1, 1, 4, 4, 1, 2>, RwBody<1, 1, 1, 1, 1, 1, 4, 4, 1, 2>, RwBody<1, 1, 2, 1, 2, 2, 4, 4, 1, 8>, 4, 4>(P, mt, 4, stream);
        else if (gkey != 888) return 0;
        else *st = launch_multi<RwBody<1, 1, 1, 1, 1, 1, 4, 4, 1, 8>, RwBody<1, 1, 1, 1, 1, 1, 4, 4, 1, 8>, RwBody<1, 1, 2, 1, 2, 2, 4, 4, 1, 8>, 4, 4>(P, mt, 4, stream);
        return 1;
    }
    if (none16 && c != 1 && sig_is(S[0], 5, 2, 4, 2, 4, 4) && sig_is(S[1], 5, 2, 4, 2, 4, 4) && sig_is(S[2], 5, 2, 4, 3, 6, 4)) {
        // layer2: 67 -> 64 -> 64 -> 128 (x2), 67 -> 64 -> 96 -> 128
        if (gkey == 444) *st = launch_multi<RwBody<5, 2, 4, 2, 4, 4, 8, 2, 0, 4>, RwBody<5, 2, 4, 2, 4, 4, 8, 2, 0, 4>, RwBody<5, 2, 4, 3, 6, 4, 8, 2, 0, 4>, 8, 2>(P, mt, 1, stream);
        else if (gkey == 224) *st = launch_multi<RwBody<5, 2, 4, 2, 4, 4, 8, 2, 0, 2>, RwBody<5, 2, 4, 2, 4, 4, 8, 2, 0, 2>, RwBody<5, 2, 4, 3, 6, 4, 8, 2, 0, 4>, 8, 2>(P, mt, 1, stream);
        else if (gkey != 888) return 0;
        else *st = launch_multi<RwBody<5, 2, 4, 2, 4, 4, 8, 2, 0, 8>, RwBody<5, 2, 4, 2, 4, 4, 8, 2, 0, 8>, RwBody<5, 2, 4, 3, 6, 4, 8, 2, 0, 8>, 8, 2>(P, mt, 1, stream);
        return 1;
    }
    if (all16 && stream_enabled && c != 1 && sig_is(S[0], 9, 4, 8, 4, 8, 8) && sig_is(S[1], 9, 4, 8, 6, 12, 8) &&
        sig_is(S[2], 9, 4, 8, 8, 16, 8) && rowwave_contiguous(S[0], wpack, 1) && rowwave_contiguous(S[1], wpack + 3, 1) &&
        rowwave_contiguous(S[2], wpack + 6, 1)) {
        // layer3, fp16
        if (gkey == 444) *st = launch_multi<RsBody<9, 4, 8, 4, 8, 8, 8, 2, 2, 1, 12, 1, 4>, RsBody<9, 4, 8, 6, 12, 8, 8, 2, 2, 1, 12, 1, 4>, RsBody<9, 4, 8, 8, 16, 8, 8, 2, 2, 1, 12, 1, 4>, 8, 2>(P, mt, 1, stream);
        else if (gkey == 488) *st = launch_multi<RsBody<9, 4, 8, 4, 8, 8, 8, 2, 2, 1, 12, 1, 4>, RsBody<9, 4, 8, 6, 12, 8, 8, 2, 2, 1, 12, 1, 8>, RsBody<9, 4, 8, 8, 16, 8, 8, 2, 2, 1, 12, 1, 8>, 8, 2>(P, mt, 1, stream);
        else if (gkey == 288) *st = launch_multi<RsBody<9, 4, 8, 4, 8, 8, 8, 2, 2, 1, 12, 1, 2>, RsBody<9, 4, 8, 6, 12, 8, 8, 2, 2, 1, 12, 1, 8>, RsBody<9, 4, 8, 8, 16, 8, 8, 2, 2, 1, 12, 1, 8>, 8, 2>(P, mt, 1, stream);
        else if (gkey != 888) return 0;
        else *st = launch_multi<RsBody<9, 4, 8, 4, 8, 8, 8, 2, 2, 1, 12, 1, 8>, RsBody<9, 4, 8, 6, 12, 8, 8, 2, 2, 1, 12, 1, 8>, RsBody<9, 4, 8, 8, 16, 8, 8, 2, 2, 1, 12, 1, 8>, 8, 2>(P, mt, 1, stream);
        return 1;
    }
    return 0;
}

// gr4: the plan holds 4-row granules.  dry: only say whether an instantiation would take the shape (nothing launched).
int sa_rowwave_try(int b, int n, int m, int ns, int c, const float *xyz, const float *feat, const float *new_xyz,
                   const int *idx, const int *cnt, int nl, const int *dims, const void *const *wpack,
                   const float *const *bias, float *out, int out_stride, int out_off, const int *plan_hdr,
                   const int *plan_gran, long max_tiles, int fp16, int gr4, int dry, int *overflow, hipStream_t stream, int *st) {
    RwParams P;
    ScaleSig S;
    if (!rowwave_scale(b, n, m, ns, c, xyz, feat, new_xyz, idx, cnt, nl, dims, wpack, bias, out, out_stride, out_off,
                       plan_hdr, plan_gran, max_tiles, overflow, P, S))
        return 0;
    if (gr4 && ns > 4 * sa::kPlanMaxOrd) return 0;
    const int KS0 = S.KS0, NT1 = S.NT1, KS1 = S.KS1, NT2 = S.NT2, KS2 = S.KS2, NT3 = S.NT3;
#define SA_RW(K0, N1, K1, N2, K2, N3_, NW_, WPE_, WGS)                                              \
    if (!fp16 && KS0 == K0 && NT1 == N1 && KS1 == K1 && NT2 == N2 && KS2 == K2 && NT3 == N3_) {             \
        if (dry) { *st = SA_OK; return 1; }                                                         \
        if (gr4) *st = c == 1 ? launch_rw<K0, N1, K1, N2, K2, N3_, NW_, WPE_, 1, 4>(P, max_tiles, WGS, stream)   \
                              : launch_rw<K0, N1, K1, N2, K2, N3_, NW_, WPE_, 0, 4>(P, max_tiles, WGS, stream);  \
        else *st = c == 1 ? launch_rw<K0, N1, K1, N2, K2, N3_, NW_, WPE_, 1, 8>(P, max_tiles, WGS, stream)       \
                          : launch_rw<K0, N1, K1, N2, K2, N3_, NW_, WPE_, 0, 8>(P, max_tiles, WGS, stream);      \
        return 1;                                                                                   \
    }
    SA_RW(1, 1, 1, 1, 1, 1, 4, 4, 4)      // 4 -> 16 -> 16 -> 32      (layer1 scales 0/1, configs[0])
    SA_RW(1, 1, 2, 1, 2, 2, 4, 4, 4)      // 4 -> 32 -> 32 -> 64      (layer1 scale 2)
    SA_RW(5, 2, 4, 2, 4, 4, 8, 2, 1)      // 67 -> 64 -> 64 -> 128    (layer2 scales 0/1)
    SA_RW(5, 2, 4, 3, 6, 4, 8, 2, 1)      // 67 -> 64 -> 96 -> 128    (layer2 scale 2)
#undef SA_RW
    // the streamed kernel walks the three layers as one linear weight stream: they must be packed back to back
    // (utils/weights.py pack_scale does that); separately allocated layers take the generic kernel
    const bool contiguous = rowwave_contiguous(S, wpack, fp16);
    static const bool stream_enabled = SA_KNOB("SA_MLP_ROWSTREAM", 1) != 0;
#define SA_RS(K0, N1, K1, N2, K2, N3_, NW_, WPE_, WGS, D_, PR_, CPP_, PF_)                              \
    if (stream_enabled && contiguous && c != 1 && (PR_ == 1) == (fp16 != 0) && KS0 == K0 && NT1 == N1 && KS1 == K1 && NT2 == N2 && KS2 == K2 && NT3 == N3_) { \
        if (dry) { *st = SA_OK; return 1; }                                                         \
        *st = gr4 ? launch_rs<K0, N1, K1, N2, K2, N3_, NW_, WPE_, 0, D_, PR_, CPP_, PF_, 4>(P, max_tiles, WGS, stream) \
                  : launch_rs<K0, N1, K1, N2, K2, N3_, NW_, WPE_, 0, D_, PR_, CPP_, PF_, 8>(P, max_tiles, WGS, stream); \
        return 1;                                                                                   \
    }
    // split bf16: 8 waves (2 per SIMD, 256 registers each), 1 workgroup per CU; staging depth as the register budget
    // allows; the widest shape runs 4 waves of 512 registers (the 8-wave form spilled 34 registers per lane)
    SA_RS(9, 4, 8, 4, 8, 8, 8, 2, 1, 2, 3, 12, 1)       // 131 -> 128 -> 128 -> 256   (layer3 scale 0): 132 k-step tiles
    SA_RS(9, 4, 8, 6, 12, 8, 8, 2, 1, 2, 3, 12, 1)      // 131 -> 128 -> 192 -> 256   (layer3 scale 1): 180
    SA_RS(9, 4, 8, 8, 16, 8, 4, 1, 1, 2, 3, 12, 1)      // 131 -> 128 -> 256 -> 256   (layer3 scale 2): 228
    // fp16 (one plane, half the fragment registers: 150-180 VGPRs, no staging-depth compromise).  Layer4 scale 0 runs
    // here too (4 waves of 512 registers: 0.035 ms against 0.050 for the LDS-activation kernel of mlp.hip); the
    // 259 -> 256 -> 512 -> 1024 scale was built in this form as well (its 512-wide hidden layer fits a wave's registers
    // in fp16: 128 for the 32 fragments) and measured SLOWER than group_mlp_wide_kernel (0.125-0.15 ms against 0.106,
    // whatever the chunk size, staging depth or fragment lookahead): one wave per SIMD has nobody to overlap its
    // 1416-MFMA chain with, so that scale stays with the wide kernel.
    SA_RS(9, 4, 8, 4, 8, 8, 8, 2, 1, 2, 1, 12, 1)       // layer3 scale 0
    SA_RS(9, 4, 8, 6, 12, 8, 8, 2, 1, 2, 1, 12, 1)      // layer3 scale 1
    SA_RS(9, 4, 8, 8, 16, 8, 8, 2, 1, 2, 1, 12, 1)      // layer3 scale 2
    // 259 -> 256 -> 256 -> 512 (layer4 scale 0): 520 tiles, 20 per chunk.  Up to one tile per wave of the 4-wave form
    // (1024 tiles on 256 CUs: batch 8) four waves of 512 registers; from two tiles per wave on, eight waves of 256
    // registers (two per SIMD: one wave's gather / conversions / barrier waits run under the other's MFMAs -- the
    // single wave of a SIMD spends 2.5-3x the MFMA time in its layers, tools/rw_phase_prof.py): 125 / 237 / 455 us for the
    // layer4 call at 8 / 16 / 32 frames with four waves, 131 / 217 / 411 with eight
    if ((long)b * m * ns / 32 >= 2048) {
        SA_RS(17, 8, 16, 8, 16, 16, 8, 2, 1, 2, 1, 26, 0)
    }
    SA_RS(17, 8, 16, 8, 16, 16, 4, 1, 1, 2, 1, 26, 0)
#undef SA_RS
    return 0;
}

#ifdef SA_RW_TIMING
extern "C" int sa_debug_rw_prof(unsigned long long *host8, int reset) {
    static unsigned long long *h = (unsigned long long *)calloc(65536 * 8, sizeof(unsigned long long));
    if (host8) {
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rw_prof), 65536 * 8 * sizeof(unsigned long long)) != hipSuccess) return SA_ERR_LAUNCH;
        for (int i = 0; i < 8; ++i) host8[i] = 0;
        unsigned long long waves = 0;
        for (int r = 0; r < 65536; ++r) { if (h[r * 8 + 7]) ++waves; for (int i = 0; i < 8; ++i) host8[i] += h[r * 8 + i]; }
        host8[8] = waves;
    }
    if (reset) {
        void *d = nullptr;
        if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_rw_prof)) != hipSuccess) return SA_ERR_LAUNCH;
        if (hipMemset(d, 0, 65536 * 8 * sizeof(unsigned long long)) != hipSuccess) return SA_ERR_LAUNCH;
    }
    return SA_OK;
}
#endif
