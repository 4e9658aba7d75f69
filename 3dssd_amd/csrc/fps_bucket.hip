// D-FPS with spatial culling at WAVE granularity, n <= 16384 (the layer-1 shape, 16384 -> 4096).
//
// The running min-distance td[k] of a point only changes when the newly selected point is closer to it than
// every earlier pick, i.e. within sqrt(td[k]).  After the first few dozen picks that is a small neighbourhood,
// yet the plain kernel (fps.hip, like the reference) re-evaluates all n distances in each of the m-1 dependent
// iterations (~2500 cycles per iteration on one CU, measured).  Here:
//   * the frame is sorted along a Morton curve in the x-z plane (bitonic sort in LDS, inside the kernel) and cut
//     into 64 buckets of 256 consecutive points; a bucket lives in ONE wave (4 points per lane, registers:
//     coordinates, td and tie keys) and has a bounding box.  Bucket b belongs to wave b mod 8, so that the
//     neighbouring buckets a pick usually touches are re-evaluated by different waves in parallel;
//   * a 64-entry table in LDS holds every bucket's current (max td, tie key of its arg-max, arg-max xyz).  An
//     iteration is: every wave reads the table (lane b <-> bucket b), takes the global arg-max with one wave
//     reduction, tests all 64 boxes against the new point in one lane-parallel pass, and re-evaluates only the
//     buckets it owns whose box lower bound is below their max td -- typically one or two buckets of the whole
//     frame.  One barrier per iteration (the table is double buffered; untouched entries are copied forward by
//     their owner).
// Exactness: the skip test is conservative (the lower bound uses the same monotone fp32 operation chain as the
// distance itself, plus a 1e-5 relative margin), every evaluated distance uses exactly the arithmetic of fps.hip,
// and ties are broken with the reference's (k mod 1024, k) order (tf_sampling_g.cu:142,154-171) on the ORIGINAL
// indices, carried as 32-bit tie keys: the four slots of a lane are ordered by tie key so "first strict maximum"
// inside a lane is the key order, equal wave / table maxima are resolved by the minimum key.  The output is
// bit-identical to the plain kernel and to the oracle -- the spatial order only affects speed.
#include "sa_common.h"

namespace {

constexpr int kW = 8;                  // waves per workgroup (2 per SIMD)
constexpr int kT = kW * 64;            // 512 threads
constexpr int kNB = 64;                // buckets == table entries == lanes of a wave
constexpr int kBPW = kNB / kW;         // buckets owned by a wave
static_assert(kBPW == 8, "the touched-bucket dispatch below names the eight buckets of a wave");
constexpr int kSL = 4;                 // points per lane per bucket
constexpr int kBS = 64 * kSL;          // points per bucket
constexpr int kCap = kNB * kBS;        // 16384 points
constexpr int kPPT = kBPW * kSL;       // 32 points per thread
constexpr float kInitTd = 1e38f;       // tf_sampling_g.cu:136
constexpr float kGone = -3.0e38f;      // padding slots / empty buckets: never selected, never updated
constexpr float kSkipMargin = 1.0f - 1e-5f;
constexpr unsigned kNoKey = 0xFFFFFFFFu;

__device__ __forceinline__ unsigned spread9(unsigned v) {   // 9 bits -> every other bit of 18
    v &= 0x1FFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
__device__ __forceinline__ float wave_allmin_f(float x) { return -sa::wave_allmax(-x); }
// reference tie order (k mod 1024, k) as one unsigned key, and back
__device__ __forceinline__ unsigned tie_key(unsigned k) { return ((k & 1023u) << 16) | (k >> 10); }
__device__ __forceinline__ unsigned tie_key_index(unsigned t) { return ((t >> 16) & 1023u) | ((t & 0xFFFFu) << 10); }
__device__ __forceinline__ float bcast(float x, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}

#ifdef SA_FPSB_PROF
// debug build only (tools/fps_bucket_prof.py): per-phase clocks of wave 0 / of the waves that had work
__device__ unsigned long long g_fpsb_prof[16];
#define FP_T(i) { const unsigned long long n__ = __builtin_readcyclecounter(); pacc[i] += n__ - pt0; pt0 = n__; }
#else
#define FP_T(i)
#endif

struct Table {
    float val[2][kNB];
    unsigned key[2][kNB];
    float4 pt[2][kNB];
};

typedef float vecf __attribute__((ext_vector_type(kPPT)));
typedef unsigned vecu __attribute__((ext_vector_type(kPPT)));

// Re-evaluation of my bucket I (compile-time: its 4 slots per lane are registers) against the new point (ox, oy, oz):
// running minima updated, the bucket's new (max, tie key, point) published into table buffer `par`.
template <int I>
__device__ __forceinline__ void bucket_update(const vecf &X, const vecf &Y, const vecf &Z, vecf &TD, const vecu &TK, float ox,
                                              float oy, float oz, Table &tbl, int par, int w, int lane) {
    float nb = -1.0f;                   // tf_sampling_g.cu:141
    int nj = 0;
    // (packed v_pk_*_f32 arithmetic was tried here: the aligned register pairs it needs made the
    //  kernel spill and it measured 10 % slower)
#pragma unroll
    for (int s = 0; s < kSL; ++s) {
        const int r = I * kSL + s;
        const float dx = X[r] - ox, dy = Y[r] - oy, dz = Z[r] - oz;
        const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));   // as fps.hip
        const float t2 = sa::fmin_nn(d, TD[r]);
        TD[r] = t2;
        const bool g = t2 > nb;         // strict: the smallest tie key among equal maxima
        nb = g ? t2 : nb;
        nj = g ? s : nj;
    }
    const float Mw = sa::wave_allmax(nb);
    unsigned long long cand = __ballot(nb == Mw);
    unsigned tkw = TK[I * kSL];
#pragma unroll
    for (int s = 1; s < kSL; ++s) tkw = nj == s ? TK[I * kSL + s] : tkw;
    if (__builtin_popcountll(cand) > 1) {           // equal maxima in several lanes: minimum tie key
        const unsigned mykey = nb == Mw ? tkw : kNoKey;
        const unsigned kmin = sa::wave_allmin_u32(mykey);
        cand = __ballot(mykey == kmin);             // keys are unique
    }
    const int wl = __builtin_ctzll(cand);
    if (lane == wl) {
        float cx = X[I * kSL], cy = Y[I * kSL], cz = Z[I * kSL];
#pragma unroll
        for (int s = 1; s < kSL; ++s) {
            const bool sel = nj == s;
            cx = sel ? X[I * kSL + s] : cx; cy = sel ? Y[I * kSL + s] : cy; cz = sel ? Z[I * kSL + s] : cz;
        }
        const int b = I * kW + w;
        tbl.val[par][b] = Mw;
        tbl.key[par][b] = tkw;
        tbl.pt[par][b] = make_float4(cx, cy, cz, 0.0f);
    }
}

// STATS (diagnostic entry sa_fps_bucket_stats only): also counts the bucket re-evaluations of the frame.
template <bool STATS>
__global__ __launch_bounds__(kT) void fps3_wave_bucket_kernel(int n, int m, const float *__restrict__ inp,
                                                             long in_bstride, int *__restrict__ out, int out_stride,
                                                             int idx_off, float *__restrict__ ctr, long ctr_bstride,
                                                             unsigned long long *__restrict__ stats) {
    __shared__ unsigned s_sorted[kCap];          // (morton << 14) | original index, ascending; 64 KiB;

    __shared__ float s_red[4][kW];
    __shared__ float s_box[6][kNB];
    __shared__ Table s_tbl;
    const int bidx = blockIdx.x;
    const float *p = inp + (size_t)bidx * (in_bstride ? in_bstride : (long)n * 3);
    int *o = out + (size_t)bidx * out_stride;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- frame bounding box in x and z (the Morton plane; LiDAR frames are flat in y)
    float mnx = 3e38f, mxx = -3e38f, mnz = 3e38f, mxz = -3e38f;
    for (int k = tid; k < n; k += kT) {
        const float x = p[k * 3 + 0], z = p[k * 3 + 2];
        mnx = sa::fmin_nn(mnx, x); mxx = sa::fmax_nn(mxx, x);
        mnz = sa::fmin_nn(mnz, z); mxz = sa::fmax_nn(mxz, z);
    }
    mnx = wave_allmin_f(mnx); mxx = sa::wave_allmax(mxx);
    mnz = wave_allmin_f(mnz); mxz = sa::wave_allmax(mxz);
    if (lane == 0) { s_red[0][w] = mnx; s_red[1][w] = mxx; s_red[2][w] = mnz; s_red[3][w] = mxz; }
    __syncthreads();
    mnx = s_red[0][0]; mxx = s_red[1][0]; mnz = s_red[2][0]; mxz = s_red[3][0];
#pragma unroll
    for (int i = 1; i < kW; ++i) {
        mnx = sa::fmin_nn(mnx, s_red[0][i]); mxx = sa::fmax_nn(mxx, s_red[1][i]);
        mnz = sa::fmin_nn(mnz, s_red[2][i]); mxz = sa::fmax_nn(mxz, s_red[3][i]);
    }
    const float sclx = 511.0f / sa::fmax_nn(mxx - mnx, 1e-20f);
    const float sclz = 511.0f / sa::fmax_nn(mxz - mnz, 1e-20f);

    // ---- Morton keys, bitonic sort in LDS (padding keys 0xFFFFFFFF sort to the end)
    for (int k = tid; k < kCap; k += kT) {
        unsigned key = kNoKey;
        if (k < n) {
            const int qx = min(511, max(0, (int)((p[k * 3 + 0] - mnx) * sclx)));
            const int qz = min(511, max(0, (int)((p[k * 3 + 2] - mnz) * sclz)));
            key = ((spread9((unsigned)qx) | (spread9((unsigned)qz) << 1)) << 14) | (unsigned)k;
        }
        s_sorted[k] = key;
    }
    __syncthreads();
    for (int kk = 2; kk <= kCap; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int q = tid; q < kCap / 2; q += kT) {
                const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1));
                const int l = i | j;
                const unsigned a = s_sorted[i], b = s_sorted[l];
                const bool up = (i & kk) == 0;
                if ((a > b) == up) { s_sorted[i] = b; s_sorted[l] = a; }
            }
            __syncthreads();
        }
    }

    // ---- my points: bucket 8i+w, sorted positions (8i+w)*256 + 4*lane + s; the four slots of a lane are then
    //      re-ordered by tie key (kept in registers).
    vecf X, Y, Z, TD;
    vecu TK;
    const int pos0 = w * kBS + lane * kSL;
#pragma unroll
    for (int i = 0; i < kBPW; ++i) {
        unsigned tk[kSL];
#pragma unroll
        for (int s = 0; s < kSL; ++s) {
            const unsigned pk = s_sorted[pos0 + i * kW * kBS + s];
            tk[s] = pk == kNoKey ? kNoKey : tie_key(pk & 0x3FFFu);
        }
        // 4-element sorting network, ascending
#define SA_CSWAP(a, b) { const unsigned x_ = tk[a], y_ = tk[b]; tk[a] = x_ < y_ ? x_ : y_; tk[b] = x_ < y_ ? y_ : x_; }
        SA_CSWAP(0, 1) SA_CSWAP(2, 3) SA_CSWAP(0, 2) SA_CSWAP(1, 3) SA_CSWAP(1, 2)
#undef SA_CSWAP
        float bx0 = 3e38f, bx1 = -3e38f, by0 = 3e38f, by1 = -3e38f, bz0 = 3e38f, bz1 = -3e38f;
#pragma unroll
        for (int s = 0; s < kSL; ++s) {
            TK[i * kSL + s] = tk[s];
            const bool ok = tk[s] != kNoKey;
            const unsigned k = ok ? tie_key_index(tk[s]) : 0u;
            const float x = p[k * 3 + 0], y = p[k * 3 + 1], z = p[k * 3 + 2];
            X[i * kSL + s] = x; Y[i * kSL + s] = y; Z[i * kSL + s] = z;
            TD[i * kSL + s] = ok ? kInitTd : kGone;
            if (ok) {
                bx0 = sa::fmin_nn(bx0, x); bx1 = sa::fmax_nn(bx1, x);
                by0 = sa::fmin_nn(by0, y); by1 = sa::fmax_nn(by1, y);
                bz0 = sa::fmin_nn(bz0, z); bz1 = sa::fmax_nn(bz1, z);
            }
        }
        bx0 = wave_allmin_f(bx0); bx1 = sa::wave_allmax(bx1);
        by0 = wave_allmin_f(by0); by1 = sa::wave_allmax(by1);
        bz0 = wave_allmin_f(bz0); bz1 = sa::wave_allmax(bz1);
        if (lane == 0) {
            const int b = i * kW + w;
            s_box[0][b] = bx0; s_box[1][b] = bx1; s_box[2][b] = by0;
            s_box[3][b] = by1; s_box[4][b] = bz0; s_box[5][b] = bz1;
        }
    }
    __syncthreads();
    // lane b of EVERY wave holds the box of bucket b (an empty bucket has an inverted box and val = kGone)
    const float bx0 = s_box[0][lane], bx1 = s_box[1][lane], by0 = s_box[2][lane];
    const float by1 = s_box[3][lane], bz0 = s_box[4][lane], bz1 = s_box[5][lane];
    // register copy of the table entry of bucket `lane`
    float val = bx0 <= bx1 ? kInitTd : kGone;
    unsigned key = kNoKey;
    float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool owner = (lane & (kW - 1)) == w;    // buckets w, w+8, ..., w+56

    float ox = p[0], oy = p[1], oz = p[2];         // old = 0, tf_sampling_g.cu:130-133
    if (tid == 0) o[0] = idx_off;

#ifdef SA_FPSB_PROF
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0 = __builtin_readcyclecounter(), pact = 0, pmine = 0, pbusy = 0;
#endif
    unsigned long long nact = 0;
    for (int it = 1; it < m; ++it) {
        const int par = it & 1;
        // ---- which buckets can change?  lower bound of the distance from the new point to each box
        const float ex = sa::fmax_nn(sa::fmax_nn(bx0 - ox, ox - bx1), 0.0f);
        const float ey = sa::fmax_nn(sa::fmax_nn(by0 - oy, oy - by1), 0.0f);
        const float ez = sa::fmax_nn(sa::fmax_nn(bz0 - oz, oz - bz1), 0.0f);
        const float lb = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
        const bool need = lb * kSkipMargin < val;
        const unsigned long long act = __ballot(need);
        const unsigned long long mine = (act >> w) & 0x0101010101010101ull;   // bit 8i <-> my bucket 8i+w
        if (STATS) nact += __builtin_popcountll(act);
#ifdef SA_FPSB_PROF
        pact += __builtin_popcountll(act); pmine += __builtin_popcountll(mine); pbusy += mine != 0;
#endif
        FP_T(0)
        // ---- re-evaluate my active buckets and publish their new entries
#define SA_FPSB_IF(I) if ((mine >> (8 * I)) & 1ull) bucket_update<I>(X, Y, Z, TD, TK, ox, oy, oz, s_tbl, par, w, lane);
        SA_FPSB_IF(0) SA_FPSB_IF(1) SA_FPSB_IF(2) SA_FPSB_IF(3) SA_FPSB_IF(4) SA_FPSB_IF(5) SA_FPSB_IF(6) SA_FPSB_IF(7)
#undef SA_FPSB_IF
        FP_T(1)
        // ---- carry my untouched entries forward into this iteration's buffer
        if (owner && !need) {
            s_tbl.val[par][lane] = val;
            s_tbl.key[par][lane] = key;
            s_tbl.pt[par][lane] = pt;
        }
        FP_T(2)
        __syncthreads();
        FP_T(3)
        // ---- global arg-max over the 64 buckets: maximum value, then minimum tie key
        val = s_tbl.val[par][lane];
        key = s_tbl.key[par][lane];
        pt = s_tbl.pt[par][lane];
        const float M = sa::wave_allmax(val);
        unsigned long long eq = __ballot(val == M);
        if (__builtin_popcountll(eq) > 1) {
            const unsigned kmn = sa::wave_allmin_u32(val == M ? key : kNoKey);
            eq = __ballot(val == M && key == kmn);
        }
        const int bl = __builtin_ctzll(eq);
        ox = bcast(pt.x, bl); oy = bcast(pt.y, bl); oz = bcast(pt.z, bl);
        // VGPR copies: an SGPR source halves the issue rate of the distance instructions (valu_rates.hip)
        asm volatile("" : "+v"(ox), "+v"(oy), "+v"(oz));
        if (tid == 0) o[it] = (int)tie_key_index((unsigned)__builtin_amdgcn_readlane((int)key, bl)) + idx_off;
        FP_T(4)
    }
    if (STATS && tid == 0) { stats[2 * bidx] = nact; stats[2 * bidx + 1] = (unsigned long long)(m - 1); }
    if (ctr) {                                       // the picked points themselves (gather_point fused, layers_util.py:116-119)
        __syncthreads();
        float *cq = ctr + (size_t)bidx * ctr_bstride;
        for (int i = tid; i < m; i += kT) {
            const int k = o[i] - idx_off;
            cq[i * 3 + 0] = p[k * 3 + 0]; cq[i * 3 + 1] = p[k * 3 + 1]; cq[i * 3 + 2] = p[k * 3 + 2];
        }
    }
#ifdef SA_FPSB_PROF
    if (bidx == 0 && lane == 0) {
        if (w == 0) { for (int i = 0; i < 5; ++i) g_fpsb_prof[i] = pacc[i]; g_fpsb_prof[5] = pact; }
        atomicAdd(&g_fpsb_prof[6], pmine); atomicAdd(&g_fpsb_prof[7], pbusy);
        atomicAdd(&g_fpsb_prof[8], pacc[1]); atomicAdd(&g_fpsb_prof[9], pacc[3]);
    }
#endif
}

}  // namespace

// D-FPS on coordinates with wave-bucket culling; same contract as sa_fps_ex2 with c == 3, n <= 16384.
extern "C" int sa_fps_bucket_ex2(int b, int n, int m, const float *inp, long in_bstride, int *out, int out_stride,
                                 int idx_off, float *ctr, long ctr_bstride, hipStream_t stream) {
    if (b <= 0 || n <= 0 || n > kCap || m <= 0 || !inp || !out || out_stride < m) return SA_ERR_INVALID;
    hipLaunchKernelGGL(fps3_wave_bucket_kernel<false>, dim3(b), dim3(kT), 0, stream, n, m, inp, in_bstride, out, out_stride,
                       idx_off, ctr, ctr_bstride, (unsigned long long *)nullptr);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// Diagnostic (bench.py's `evaluated_pairs`, tests): the same picks, plus per frame stats[2f] = number of bucket
// re-evaluations (each = 256 point slots against the new pick) and stats[2f+1] = m - 1 picks.  The reference kernel
// evaluates (m - 1) * n pairs (tf_sampling_g.cu:139-160).
extern "C" int sa_fps_bucket_stats(int b, int n, int m, const float *inp, int *out, unsigned long long *stats,
                                   hipStream_t stream) {
    if (b <= 0 || n <= 0 || n > kCap || m <= 0 || !inp || !out || !stats) return SA_ERR_INVALID;
    hipLaunchKernelGGL(fps3_wave_bucket_kernel<true>, dim3(b), dim3(kT), 0, stream, n, m, inp, 0l, out, m, 0,
                       (float *)nullptr, 0l, stats);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

#ifdef SA_FPSB_PROF
extern "C" int sa_debug_fpsb_prof(unsigned long long *host16, int reset) {
    if (host16 && hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_fpsb_prof), sizeof(g_fpsb_prof)) != hipSuccess) return SA_ERR_LAUNCH;
    if (reset) {
        void *d = nullptr;
        if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_fpsb_prof)) != hipSuccess) return SA_ERR_LAUNCH;
        if (hipMemset(d, 0, sizeof(g_fpsb_prof)) != hipSuccess) return SA_ERR_LAUNCH;
    }
    return SA_OK;
}
#endif
