// F-FPS WITHOUT the distance matrix (round 4).
//
// The reference samples an 'FS' / 'F-FPS' range (lib/utils/layers_util.py:93-104) by building the full
// [n, n] matrix calc_square_dist(concat(xyz, feat)) (model_util.py:144-160) and walking it row by row
// (tf_sampling_g.cu:180-230).  At layer 2 (n = 4096, 3 + 64 channels, 512 picks) that matrix is 64 MiB per frame
// of which the sampler reads 512 rows: 87.5 % of 2.3 GB per 32 frames is written to HBM and never read, and the
// matrix kernel is 28 % of the chip time of a package (bench.py stages, round 4).
//
// Here the row of the matrix that a pick needs is computed when it is needed.  G = n / 1024 workgroups of 512
// threads share a frame; a thread keeps TWO points (all 67 channels, their squared norms and running minima) in
// registers for the whole kernel -- 1024 points x 67 channels = 268 KB, half of a CU's register file -- and per
// pick evaluates d(old, k) = (|old|^2 + |k|^2) - 2 <old, k> for its two points with ONE packed fp32 FMA per channel
// (v_pk_fma_f32: the two halves are the two points, so each point's dot product is the same sequential fmaf chain
// over channels ascending from 0 as the oracle's / csrc/sqdist.hip's -- bit-identical rows).  The arg-max goes
// wave -> workgroup (LDS, one barrier) -> frame: every workgroup publishes {max | pick | tie key} and the squared
// norm of its candidate in two 8-byte words (agent-scope atomic stores) and polls its partners' words (agent-scope
// atomic loads; bounded: a workgroup that gives up raises the sticky error word of sa_common.h and returns); the winner's row is then read from the (static) input.
//
// Semantics = farthest_point_sample_with_distance on that matrix, exactly: thread t of the reference owns
// k = t, t + 1024, ... and keeps its first strict maximum, lower lane / wave wins a tie, i.e. ties go to the lowest
// (k mod 1024, k div 1024).  Here workgroup g owns k div 1024 == g and thread t the points k mod 1024 = t and
// t + 512; every stage compares (value descending, key ascending) explicitly with key = (k mod 1024) << 6 | k div 1024.
//
// Co-residency: the G partners of a frame must run at the same time.  A launch never has more workgroups than the
// device keeps resident for this kernel (queried once; larger batches run as consecutive launches inside the call,
// every frame with its own exchange words), and the CALLER keeps all launches of this kernel -- and of
// fps_coop_kernel, csrc/fps_coop.hip, which explains why -- on ONE stream: the staged executor has a stream for
// exactly that (3dssd_amd/pipeline.py, "fly").
#include <stdint.h>

#include "sa_common.h"

namespace {

constexpr int kT = 512, kW = kT / 64;      // threads / waves per workgroup
constexpr int kPW = 1024;                  // points per workgroup (two per thread)
constexpr float kInit = 1e38f;             // tf_sampling_g.cu:188
constexpr float kAbsent = -3.0e38f;        // slot of a thread that owns no point: never beats best = -1
constexpr unsigned kMaxSpin = 1u << 22;
typedef float f2 __attribute__((ext_vector_type(2)));

struct FlyArgs {
    const float *xyz;  long xyz_bs;        // [b, ., 3]: start of the sampled range, floats between frames
    const float *feat; long feat_bs;       // [b, ., C1]
    unsigned long long *words;             // exchange words: [frame][parity 2][G][2]
    int *out; int out_stride, idx_off;
    float *ctr; long ctr_bs;               // picked xyz rows [b, ., 3] (or null)
    int n, m, gshift;
    int *err_word;                         // sticky error word (sa_common.h), may be null
};

template <int C1>
__global__ __launch_bounds__(kT) void ffps_fly_kernel(FlyArgs A) {
    constexpr int C = 3 + C1;
    __shared__ float s_val[2][kW];
    __shared__ unsigned s_key[2][kW];
    __shared__ float s_sq[2][kW];
    __shared__ int s_old[2];
    __shared__ float s_oldsq[2];
    const int G = 1 << A.gshift;
    const int f = blockIdx.x >> A.gshift, g = blockIdx.x & (G - 1);
    const float *px = A.xyz + (size_t)f * A.xyz_bs, *pf = A.feat + (size_t)f * A.feat_bs;
    int *o = A.out + (size_t)f * A.out_stride;
    unsigned long long *sl = A.words + (size_t)f * 4 * G;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;

    // ---- my two points: channels (xyz first, model_util / layers_util.py:94,102), squared norms, running minima
    f2 x[C];
    float sq[2], td[2];
    unsigned key[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kl = j * (kPW / 2) + t;                  // k mod 1024
        const int k = g * kPW + kl;
        const bool ok = k < A.n;
        const int kk = ok ? k : 0;
        float s = 0.0f;
#pragma unroll
        for (int l = 0; l < C; ++l) {
            const float v = l < 3 ? px[(size_t)kk * 3 + l] : pf[(size_t)kk * C1 + (l - 3)];
            x[l][j] = v;
            s = __builtin_fmaf(v, v, s);                   // |k|^2: fmaf chain ascending (oracle decision E)
        }
        sq[j] = s;
        td[j] = ok ? kInit : kAbsent;
        key[j] = ((unsigned)kl << 6) | (unsigned)g;
    }
    int old = 0;                                           // tf_sampling_g.cu:186
    if (g == 0 && t == 0) o[0] = A.idx_off;
    // |old|^2 of the first pick: its own chain (every later one travels with the exchange)
    float sq_old = 0.0f;
#pragma unroll
    for (int l = 0; l < C; ++l) {
        const float v = l < 3 ? px[l] : pf[l - 3];
        sq_old = __builtin_fmaf(v, v, sq_old);
    }

    for (int it = 1; it < A.m; ++it) {
        // ---- row `old` of the matrix for my two points: <old, k> as one packed chain, then (|old|^2 + |k|^2) - 2 <old, k>
        const float *qx = px + (size_t)old * 3, *qf = pf + (size_t)old * C1;   // uniform addresses
        f2 acc = {0.0f, 0.0f};
#pragma unroll
        for (int l = 0; l < C; ++l) {
            const float ql = l < 3 ? qx[l] : qf[l - 3];
            const f2 qq = {ql, ql};
            acc = __builtin_elementwise_fma(x[l], qq, acc);
        }
        float best = -1.0f;                                // tf_sampling_g.cu:191
        unsigned bkey = 0u;
        float bsq = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float d = (sq_old + sq[j]) - 2.0f * acc[j];
            const float t2 = sa::fmin_nn(d, td[j]);
            td[j] = t2;
            const bool gt = t2 > best;                     // strict: the lower k mod 1024 keeps a tie
            best = gt ? t2 : best;
            bkey = gt ? key[j] : bkey;
            bsq = gt ? sq[j] : bsq;
        }
        // ---- wave: maximum, then minimum key among the lanes that hold it
        const float wmax = sa::wave_allmax(best);
        unsigned long long cand = __ballot(best == wmax);
        if (__builtin_popcountll(cand) > 1) {
            const unsigned kmin = sa::wave_allmin_u32(best == wmax ? bkey : 0xFFFFFFFFu);
            cand = __ballot(best == wmax && bkey == kmin);
        }
        const int par = it & 1;
        if (lane == __builtin_ctzll(cand)) {
            s_val[par][w] = wmax;
            s_key[par][w] = bkey;
            s_sq[par][w] = bsq;
        }
        __syncthreads();
        // ---- workgroup: the same over the 8 wave entries (every lane reads entry lane & 7)
        const float v8 = s_val[par][lane & (kW - 1)];
        const unsigned k8 = s_key[par][lane & (kW - 1)];
        const float q8 = s_sq[par][lane & (kW - 1)];
        const float M = sa::row16_allmax(v8);
        const unsigned kwin = sa::row16_allmin_u32(v8 == M ? k8 : 0xFFFFFFFFu);
        const int ewin = __builtin_ctzll(__ballot(v8 == M && k8 == kwin));
        const float sqwin = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q8), ewin));
        const unsigned kw_u = (unsigned)__builtin_amdgcn_readfirstlane((int)kwin);
        const float M_u = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(M)));
        unsigned long long *mine = sl + ((size_t)par * G + g) * 2;
        if (t < 2) {
            const unsigned hi = t == 0 ? __float_as_uint(M_u) : __float_as_uint(sqwin);
            const unsigned lo = ((unsigned)it << 16) | (t == 0 ? kw_u : 0u);
            __hip_atomic_store(mine + t, ((unsigned long long)hi << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- frame: ONE wave polls the 2 G words of this pick and hands (winner, |winner|^2) to the others through LDS + a
        //      second barrier (round 5, as in fps_coop.hip: eight polling waves per workgroup fought over the same lines)
        if (w == 0) {
            const int nw = 2 * G;
            const unsigned long long *sp = sl + (size_t)par * G * 2 + (lane < nw ? lane : 0);
            unsigned long long wv;
            unsigned spins = 0;
            int winner = -1;                               // -1: partners lost
            float wsq = 0.0f;
            for (;;) {
                wv = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool okw = ((((unsigned)wv) >> 16) & 0xFFFFu) == (unsigned)it;
                if (__ballot(okw || lane >= nw) == ~0ull) {
                    const bool isval = lane < nw && (lane & 1) == 0;   // even lanes: {max | pick | key}, odd: {|candidate|^2 | pick}
                    const float gv = isval ? __uint_as_float((unsigned)(wv >> 32)) : -3.4e38f;
                    const float GM = sa::row16_allmax(gv);
                    const unsigned gk = sa::row16_allmin_u32((isval && gv == GM) ? ((unsigned)wv & 0xFFFFu) : 0xFFFFFFFFu);
                    const unsigned gk_u = (unsigned)__builtin_amdgcn_readfirstlane((int)gk);
                    const int gwin = (int)(gk_u & 63u);
                    wsq = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)(unsigned)(wv >> 32), 2 * gwin + 1));
                    winner = (int)(gk_u >> 6) + kPW * gwin;
                    break;
                }
                if (++spins > kMaxSpin) {                  // partners lost: sticky error word (sa_common.h); every wave leaves below
                    sa::coop_raise(A.err_word, sa::kCoopErrFfps);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) { s_old[par] = winner; s_oldsq[par] = wsq; }
        }
        __syncthreads();
        old = s_old[par];
        sq_old = s_oldsq[par];
        if (old < 0) return;
        if (g == 0 && t == 0) o[it] = old + A.idx_off;
    }
    if (A.ctr && g == 0) {                                 // the picked points themselves (layers_util.py:116-119)
        __syncthreads();
        float *c = A.ctr + (size_t)f * A.ctr_bs;
        for (int i = t; i < A.m; i += kT) {
            const int k = o[i] - A.idx_off;
            c[i * 3 + 0] = px[(size_t)k * 3 + 0]; c[i * 3 + 1] = px[(size_t)k * 3 + 1]; c[i * 3 + 2] = px[(size_t)k * 3 + 2];
        }
    }
}

int g_cap64 = 0;   // resident workgroups of ffps_fly_kernel<64> (0 = not yet queried, < 0 = unusable); atomic accesses

}  // namespace

// Bytes of exchange scratch sa_ffps_fly_ex needs for b frames of n points.
extern "C" size_t sa_ffps_fly_ws_bytes(int b, int n) {
    if (b <= 0 || n <= 0) return 0;
    const int G = (n + kPW - 1) / kPW;
    return (size_t)b * 4 * (size_t)(G < 1 ? 1 : G) * sizeof(unsigned long long);
}

// F-FPS of `m` points on rows [0, n) of (xyz, feat) per frame, on the fly.  out[f, 0..m) = idx_off + pick;
// ctr (optional) receives the picked xyz rows.  Supported: feature width 64, n = 1024 / 2048 / 4096 (1 / 2 / 4
// workgroups per frame), m <= 65535; SA_ERR_UNSUPPORTED otherwise (the caller then takes the matrix path).
extern "C" int sa_ffps_fly_ex(int b, int n, int c1, int m, const float *xyz, long xyz_bstride, const float *feat,
                              long feat_bstride, void *workspace, int *out, int out_stride, int idx_off, float *ctr,
                              long ctr_bstride, hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || !xyz || !feat || !out || !workspace || out_stride < m) return SA_ERR_INVALID;
    if (c1 != 64 || m > 65535 || (n != 1024 && n != 2048 && n != 4096) || m > n) return SA_ERR_UNSUPPORTED;
    int gshift = 0;
    while ((kPW << gshift) < n) ++gshift;
    const int G = 1 << gshift;
    int *err_word = sa::coop_error_word();
    if (err_word && __atomic_load_n(err_word, __ATOMIC_RELAXED) != 0) return SA_ERR_PARTNERS;   // sticky: an earlier launch lost partners
    int cap = __atomic_load_n(&g_cap64, __ATOMIC_ACQUIRE);
    if (cap == 0) {                                         // first use (two threads may both query: same answer)
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)ffps_fly_kernel<64>, kT, 0) != hipSuccess)
            cap = -1;
        else
            cap = cus * per_cu > 0 ? cus * per_cu : -1;
        (void)hipGetLastError();
        __atomic_store_n(&g_cap64, cap, __ATOMIC_RELEASE);
    }
    if (cap < G) return SA_ERR_UNSUPPORTED;
    if (!err_word) {                                        // no error word (its allocation is refused during a capture): never run unchecked under one
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return SA_ERR_UNSUPPORTED; }
    }
    const int per_launch = cap / G;
    if (sa::zero_async(workspace, sa_ffps_fly_ws_bytes(b, n), stream) != hipSuccess) return SA_ERR_LAUNCH;
    for (int f0 = 0; f0 < b; f0 += per_launch) {
        const int nf = b - f0 < per_launch ? b - f0 : per_launch;
        FlyArgs A;
        A.xyz = xyz + (size_t)f0 * (xyz_bstride ? xyz_bstride : (long)n * 3);
        A.xyz_bs = xyz_bstride ? xyz_bstride : (long)n * 3;
        A.feat = feat + (size_t)f0 * (feat_bstride ? feat_bstride : (long)n * c1);
        A.feat_bs = feat_bstride ? feat_bstride : (long)n * c1;
        A.words = (unsigned long long *)workspace + (size_t)f0 * 4 * G;
        A.out = out + (size_t)f0 * out_stride; A.out_stride = out_stride; A.idx_off = idx_off;
        A.ctr = ctr ? ctr + (size_t)f0 * ctr_bstride : nullptr; A.ctr_bs = ctr_bstride;
        A.n = n; A.m = m; A.gshift = gshift;
        A.err_word = err_word;
        hipLaunchKernelGGL(ffps_fly_kernel<64>, dim3(nf * G), dim3(kT), 0, stream, A);
        SA_CHECK_LAUNCH();
    }
    return SA_OK;
}
