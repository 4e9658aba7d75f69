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

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
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
    int rp;        // rows per ball after padding: 8, 16, 32 or a multiple of 32
    int rp_shift;  // log2(rp) when rp <= 32
    int N3;        // true output channels of the last layer
    int ntiles;    // 32-row tiles
    int tpu;       // tiles per pooling unit (rp / 32 when rp > 32, else 1)
};

__device__ __forceinline__ f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
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

struct RowRef { int pt, ball, cnt; };   // source point (flat index), ball and its count for this lane's row

// (ball, sample) of tile T, row r  ->  source point; both loads issue together
// unit u (a ball when rp > 32, else the tile itself), tile tp of the unit
__device__ __forceinline__ RowRef load_row_ref(const RwParams &P, int u, int tp, int row) {
    long ball;
    int s;
    if (P.rp <= 32) { const int bl = row >> P.rp_shift; ball = ((long)u << (5 - P.rp_shift)) + bl; s = row - (bl << P.rp_shift); }
    else { ball = u; s = tp * 32 + row; }
    if (ball >= P.nballs) ball = P.nballs - 1;
    if (s >= P.ns) s = 0;                                   // padded rows repeat sample 0
    const int a_raw = P.idx[ball * P.ns + s];
    const int c = P.cnt[ball];
    RowRef r;
    r.cnt = c;
    r.ball = (int)ball;
    r.pt = (int)((ball / P.m) * P.n) + (c > 0 ? a_raw : 0); // layers_util.py:157-159
    return r;
}

// Relative coordinates (and, for C == 1, the single feature channel) of this lane's row.
struct RowTail { float t[4]; };
template <int TAILF>
__device__ __forceinline__ RowTail load_row_tail(const RwParams &P, const RowRef &rr) {
    RowTail r;
    const float px = P.xyz[(long)rr.pt * 3 + 0] - P.new_xyz[(long)rr.ball * 3 + 0];
    const float py = P.xyz[(long)rr.pt * 3 + 1] - P.new_xyz[(long)rr.ball * 3 + 1];
    const float pz = P.xyz[(long)rr.pt * 3 + 2] - P.new_xyz[(long)rr.ball * 3 + 2];
    if (TAILF) { r.t[0] = P.feat[(long)rr.pt * P.C + (P.C - 1)]; r.t[1] = px; r.t[2] = py; r.t[3] = pz; }
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
        const float4 f0 = *(const float4 *)(P.feat + (long)rr.pt * P.C + 8 * g);
        const float4 f1 = *(const float4 *)(P.feat + (long)rr.pt * P.C + 8 * g + 4);
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
template <int KSN>
__device__ __forceinline__ void acc_to_frags(const f32x16 &acc, int ct, uint4 (&fh)[KSN], uint4 (&fl)[KSN]) {
#pragma unroll
    for (int hk = 0; hk < 2; ++hk) {
        if (2 * ct + hk < KSN) {
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
        }
    }
}

// max over the rows of each ball for one 32-row tile (D form: reg r of lane (col, h) = row (r&3)+8*(r>>2)+4h)
__device__ __forceinline__ void tile_ball_max(const f32x16 &a, int rp, float (&bm)[4]) {
    float qm[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a0 = sa::fmax_nn(a[4 * q], a[4 * q + 1]);
        const float a1 = sa::fmax_nn(a[4 * q + 2], a[4 * q + 3]);
        const float x = sa::fmax_nn(a0, a1);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        qm[q] = sa::fmax_nn(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (rp == 8) { bm[0] = qm[0]; bm[1] = qm[1]; bm[2] = qm[2]; bm[3] = qm[3]; }
    else if (rp == 16) { bm[0] = sa::fmax_nn(qm[0], qm[1]); bm[1] = sa::fmax_nn(qm[2], qm[3]); bm[2] = bm[3] = 0.f; }
    else { bm[0] = sa::fmax_nn(sa::fmax_nn(qm[0], qm[1]), sa::fmax_nn(qm[2], qm[3])); bm[1] = bm[2] = bm[3] = 0.f; }
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
            if (ct0 + tt < NT) acc_to_frags<KSN>(acc[tt], ct0 + tt, oh, ol);
    }
}

// KS0: k-steps of the gathered input; (NT1, KS1), (NT2, KS2): output tiles of hidden layer 1 / 2 and the
// k-steps the next layer reads of them; NT3: output tiles of the last layer.  NW waves per workgroup.
template <int KS0, int NT1, int KS1, int NT2, int KS2, int NT3, int NW, int WPE, int TAILF>
__global__ __launch_bounds__(NW * 64, WPE) void mlp_rw_kernel(RwParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int nW0 = NT1 * KS0 * 128, nW1 = NT2 * KS1 * 128, nW2 = NT3 * KS2 * 128;   // uint4 counts
    uint4 *W0 = (uint4 *)smem, *W1 = W0 + nW0, *W2 = W1 + nW1;
    float *b0 = (float *)(W2 + nW2), *b1 = b0 + NT1 * 32, *b2 = b1 + NT2 * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < nW0; i += NW * 64) W0[i] = P.w[0][i];
    for (int i = tid; i < nW1; i += NW * 64) W1[i] = P.w[1][i];
    for (int i = tid; i < nW2; i += NW * 64) W2[i] = P.w[2][i];
    for (int i = tid; i < NT1 * 32; i += NW * 64) b0[i] = P.bias[0][i];
    for (int i = tid; i < NT2 * 32; i += NW * 64) b1[i] = P.bias[1][i];
    for (int i = tid; i < NT3 * 32; i += NW * 64) b2[i] = P.bias[2][i];
    __syncthreads();

    const int row = lane & 31, half = lane >> 5;
    const int gw = blockIdx.x * NW + w, nwaves = gridDim.x * NW;
    const int nunits = P.ntiles / P.tpu;
    if (gw >= nunits) return;
    // this wave's tiles: units gw, gw + nwaves, ... ; tiles 0 .. tpu-1 of each.  (u, tp) cursors for the three
    // pipeline stages: compute (uc, tc), feature loads (one tile ahead), index loads (two ahead).
    auto advance = [&](int &u, int &tp) {
        if (tp + 1 < P.tpu) ++tp;
        else if (u + nwaves < nunits) { u += nwaves; tp = 0; }      // past the end: stay on the last tile
    };
    int uc = gw, tc = 0, uf = gw, tf = 0;
    advance(uf, tf);
    int ui = uf, ti = tf;
    advance(ui, ti);

    float raw[KS0][8];
    RowRef cur = load_row_ref(P, uc, tc, row);
    {
        const RowTail tl = load_row_tail<TAILF>(P, cur);
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) load_group(P, cur, tl, 2 * ks + half, raw[ks]);
    }
    RowRef nxt = load_row_ref(P, uf, tf, row);
    float pooled[NT3][4];

    for (bool more = true; more;) {
        const int T_u = uc, tp = tc;
        more = tc + 1 < P.tpu || uc + nwaves < nunits;
        // ---- this tile's input as B-operand fragments
        uint4 h0[KS0], l0[KS0];
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) split8(raw[ks], h0[ks], l0[ks]);
        const int cnt_row = cur.cnt;
        // ---- loads of the next tile (features) and of the one after (indices), then pin them above the math
        cur = nxt;
        {
            const RowTail tl = load_row_tail<TAILF>(P, cur);
#pragma unroll
            for (int ks = 0; ks < KS0; ++ks) load_group(P, cur, tl, 2 * ks + half, raw[ks]);
        }
        nxt = load_row_ref(P, ui, ti, row);
        advance(uc, tc);
        advance(uf, tf);
        advance(ui, ti);
        __builtin_amdgcn_sched_barrier(0);

        uint4 h1[KS1], l1[KS1];
        hidden_layer<KS0, NT1, KS1>(W0, b0, h0, l0, h1, l1, lane);
        uint4 h2[KS2], l2[KS2];
        hidden_layer<KS1, NT2, KS2>(W1, b1, h1, l1, h2, l2, lane);
        // ---- last layer (D form) + max over the rows of each ball
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
                    float bm[4];
                    tile_ball_max(acc[tt], P.rp, bm);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        pooled[ct0 + tt][g] = tp == 0 ? bm[g] : sa::fmax_nn(pooled[ct0 + tt][g], bm[g]);
                }
            }
        }
        // ---- write out after the last tile of the unit: relu(max + bias), zero for empty balls
        //      (layers_util.py:178-181)
        if (tp == P.tpu - 1) {
            const int nb = P.rp <= 32 ? 32 >> P.rp_shift : 1;
            const long ball0 = P.rp <= 32 ? (long)T_u * nb : T_u;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < nb) {
                    const long ball = ball0 + g;
                    const int cg = __builtin_amdgcn_readlane(cnt_row, P.rp <= 32 ? g * P.rp : 0);
                    if (ball < P.nballs && lane < 32) {
#pragma unroll
                        for (int ct = 0; ct < NT3; ++ct) {
                            const int c = ct * 32 + lane;
                            if (c < P.N3) {
                                float v = pooled[ct][g] + b2[c];
                                v = v > 0.0f ? v : 0.0f;
                                if (cg <= 0) v = 0.0f;
                                P.out[ball * P.out_stride + P.out_off + c] = v;
                            }
                        }
                    }
                }
            }
        }
    }
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

template <int KS0, int NT1, int KS1, int NT2, int KS2, int NT3, int NW, int WPE, int TAILF>
int launch_rw(const RwParams &P, int wgs_per_cu, hipStream_t stream) {
    constexpr size_t lds = (size_t)(NT1 * KS0 + NT2 * KS1 + NT3 * KS2) * 2048 + (size_t)(NT1 + NT2 + NT3) * 128;
    auto kern = mlp_rw_kernel<KS0, NT1, KS1, NT2, KS2, NT3, NW, WPE, TAILF>;
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipGetLastError();
    }
    const long nunits = P.ntiles / P.tpu;
    long grid = (nunits + NW - 1) / NW;
    const long cap = (long)num_cus() * wgs_per_cu;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, P);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

}  // namespace

// Returns 1 when the shape has a row-wave instantiation and the launch was issued (status in *st), 0 when the
// caller should take the generic kernel.
int sa_rowwave_try(int b, int n, int m, int ns, int c, const float *xyz, const float *feat, const float *new_xyz,
                   const int *idx, const int *cnt, int nl, const int *dims, const void *const *wpack,
                   const float *const *bias, float *out, int out_stride, int out_off, hipStream_t stream,
                   int *st) {
    static const bool enabled = !(getenv("SA_MLP_ROWWAVE") && atoi(getenv("SA_MLP_ROWWAVE")) == 0);
    if (!enabled || nl != 3) return 0;
    if (!(c == 1 || (c > 0 && (c & 7) == 0))) return 0;      // input layouts the in-register gather handles
    const int KS0 = roundup(dims[0], 16) / 16;
    const int NT1 = roundup(dims[1], 32) / 32, KS1 = roundup(dims[1], 16) / 16;
    const int NT2 = roundup(dims[2], 32) / 32, KS2 = roundup(dims[2], 16) / 16;
    const int NT3 = roundup(dims[3], 32) / 32;
    RwParams P{};
    P.xyz = xyz; P.feat = feat; P.new_xyz = new_xyz; P.idx = idx; P.cnt = cnt; P.out = out;
    for (int l = 0; l < 3; ++l) { P.w[l] = (const uint4 *)wpack[l]; P.bias[l] = bias[l]; }
    P.n = n; P.m = m; P.ns = ns; P.C = c; P.nballs = (long)b * m;
    P.out_stride = out_stride; P.out_off = out_off;
    P.rp = ns <= 8 ? 8 : (ns <= 16 ? 16 : roundup(ns, 32));
    P.N3 = dims[3];
    P.rp_shift = P.rp == 8 ? 3 : (P.rp == 16 ? 4 : 5);
    P.tpu = P.rp <= 32 ? 1 : P.rp / 32;
    const int bpt = P.rp <= 32 ? 32 / P.rp : 1;
    const long ntiles = P.rp <= 32 ? (P.nballs + bpt - 1) / bpt : P.nballs * P.tpu;
    if (ntiles > 0x7FFFFFFFl) return 0;
    P.ntiles = (int)ntiles;
#define SA_RW(K0, N1, K1, N2, K2, N3_, NW_, WPE_, WGS)                                              \
    if (KS0 == K0 && NT1 == N1 && KS1 == K1 && NT2 == N2 && KS2 == K2 && NT3 == N3_) {             \
        *st = c == 1 ? launch_rw<K0, N1, K1, N2, K2, N3_, NW_, WPE_, 1>(P, WGS, stream)             \
                     : launch_rw<K0, N1, K1, N2, K2, N3_, NW_, WPE_, 0>(P, WGS, stream);            \
        return 1;                                                                                   \
    }
    SA_RW(1, 1, 1, 1, 1, 1, 4, 4, 4)      // 4 -> 16 -> 16 -> 32      (layer1 scales 0/1, configs[0])
    SA_RW(1, 1, 2, 1, 2, 2, 4, 4, 4)      // 4 -> 32 -> 32 -> 64      (layer1 scale 2)
    SA_RW(5, 2, 4, 2, 4, 4, 8, 2, 1)      // 67 -> 64 -> 64 -> 128    (layer2 scales 0/1)
    SA_RW(5, 2, 4, 3, 6, 4, 8, 2, 1)      // 67 -> 64 -> 96 -> 128    (layer2 scale 2)
#undef SA_RW
    return 0;
}
