// D-FPS with spatial culling, n <= 16384 (the layer-1 shape, 16384 -> 4096).
//
// The running min-distance td[k] of a point only changes when the newly selected point is closer to it
// than every earlier pick, i.e. within sqrt(td[k]).  After the first few dozen picks that is a small
// neighbourhood, yet the plain kernel (fps.hip, like the reference) re-evaluates all n distances in each
// of the m-1 dependent iterations.  Here the frame is first sorted along a Morton curve (bitonic sort in
// LDS, inside the same kernel) and every thread owns 16 CONSECUTIVE sorted points -- a compact bucket with
// its own bounding box.  A thread re-evaluates its bucket in an iteration only if the box's lower-bound
// distance to the new point is below the bucket's current maximum of td; otherwise its cached
// (max, arg-max) is still exact.  A wave whose 64 buckets are all skipped spends ~15 VALU instructions in
// that iteration instead of ~180.
//
// Exactness: the skip test is conservative (relative margin 1e-5 against the <= 8 ulp error of the fp32
// chains), every evaluated distance uses exactly the arithmetic of fps.hip, and ties are broken with the
// reference's (k mod 1024, k) order on the ORIGINAL indices carried as 32-bit tie keys (each thread keeps
// its 16 points sorted by that key, so "first strict maximum" inside a thread is the key order), so the
// output is bit-identical to the plain kernel and to the oracle -- the spatial order only affects speed.
#include "sa_common.h"

namespace {

constexpr int kBW = 16;                // waves
constexpr int kBThreads = kBW * 64;    // 1024
constexpr int kPPT = 16;               // points per thread = bucket size
constexpr int kCap = kBThreads * kPPT; // 16384 points
constexpr float kInitTd = 1e38f;       // tf_sampling_g.cu:136
constexpr float kGone = -3.0e38f;      // padding slots: never selected, never updated
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

__global__ __launch_bounds__(kBThreads) void fps3_bucket_kernel(int n, int m, const float *__restrict__ inp,
                                                                int *__restrict__ out, int out_stride,
                                                                int idx_off) {
    __shared__ unsigned s_sorted[kCap];          // (morton << 14) | original index, ascending; 64 KiB;
                                                 // afterwards: tie key of (thread t, slot j) at [t*16 + j]
    __shared__ float s_red[4][kBW];
    __shared__ float s_val[2][kBW];
    __shared__ unsigned s_key[2][kBW];
    __shared__ float4 s_pt[2][kBW];
    const int bidx = blockIdx.x;
    const float *p = inp + (size_t)bidx * n * 3;
    int *o = out + (size_t)bidx * out_stride;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

    // ---- frame bounding box in x and z (the Morton plane; LiDAR frames are flat in y)
    float mnx = 3e38f, mxx = -3e38f, mnz = 3e38f, mxz = -3e38f;
    for (int k = tid; k < n; k += kBThreads) {
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
    for (int i = 1; i < kBW; ++i) {
        mnx = sa::fmin_nn(mnx, s_red[0][i]); mxx = sa::fmax_nn(mxx, s_red[1][i]);
        mnz = sa::fmin_nn(mnz, s_red[2][i]); mxz = sa::fmax_nn(mxz, s_red[3][i]);
    }
    const float sclx = 511.0f / sa::fmax_nn(mxx - mnx, 1e-20f);
    const float sclz = 511.0f / sa::fmax_nn(mxz - mnz, 1e-20f);

    // ---- Morton keys, bitonic sort in LDS (padding keys 0xFFFFFFFF sort to the end)
    for (int k = tid; k < kCap; k += kBThreads) {
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
            for (int q = tid; q < kCap / 2; q += kBThreads) {
                const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1));
                const int l = i | j;
                const unsigned a = s_sorted[i], b = s_sorted[l];
                const bool up = (i & kk) == 0;
                if ((a > b) == up) { s_sorted[i] = b; s_sorted[l] = a; }
            }
            __syncthreads();
        }
    }

    // ---- my bucket: sorted positions tid*16 .. +15, re-ordered inside the thread by tie key
    unsigned TK[kPPT];
#pragma unroll
    for (int j = 0; j < kPPT; ++j) {
        const unsigned pk = s_sorted[tid * kPPT + j];
        TK[j] = pk == kNoKey ? kNoKey : tie_key(pk & 0x3FFFu);
    }
#pragma unroll
    for (int r = 0; r < kPPT; ++r) {               // odd-even transposition sort, ascending tie key
#pragma unroll
        for (int j = (r & 1); j + 1 < kPPT; j += 2) {
            const unsigned a = TK[j], b = TK[j + 1];
            TK[j] = a < b ? a : b;
            TK[j + 1] = a < b ? b : a;
        }
    }
    __syncthreads();                               // everybody has read its Morton keys
#pragma unroll
    for (int j = 0; j < kPPT; ++j) s_sorted[tid * kPPT + j] = TK[j];   // now: tie keys
    const unsigned anyk = TK[0] == kNoKey ? 0u : tie_key_index(TK[0]);  // a real point of this bucket (or 0)
    float X[kPPT], Y[kPPT], Z[kPPT], TD[kPPT];
    float bx0 = 3e38f, bx1 = -3e38f, by0 = 3e38f, by1 = -3e38f, bz0 = 3e38f, bz1 = -3e38f;
#pragma unroll
    for (int j = 0; j < kPPT; ++j) {
        const bool ok = TK[j] != kNoKey;
        const unsigned k = ok ? tie_key_index(TK[j]) : anyk;
        X[j] = p[k * 3 + 0];
        Y[j] = p[k * 3 + 1];
        Z[j] = p[k * 3 + 2];
        TD[j] = ok ? kInitTd : kGone;
        bx0 = sa::fmin_nn(bx0, X[j]); bx1 = sa::fmax_nn(bx1, X[j]);
        by0 = sa::fmin_nn(by0, Y[j]); by1 = sa::fmax_nn(by1, Y[j]);
        bz0 = sa::fmin_nn(bz0, Z[j]); bz1 = sa::fmax_nn(bz1, Z[j]);
    }
    // bucket state: its maximum of td and where it is (slot with the smallest tie key among equals)
    float best = TK[0] != kNoKey ? kInitTd : kGone;
    int bj = 0;

    float ox = p[0], oy = p[1], oz = p[2];         // old = 0, tf_sampling_g.cu:130-133
    if (tid == 0) o[0] = idx_off;
    // this wave's published candidate (wave-uniform), re-derived only when one of its buckets changed
    float pubM = kGone, pubX = 0.f, pubY = 0.f, pubZ = 0.f;
    unsigned pubKey = kNoKey;
    bool first = true;

    for (int it = 1; it < m; ++it) {
        // ---- can my bucket change?  lower bound of the distance from the new point to the bucket's box
        const float ex = sa::fmax_nn(sa::fmax_nn(bx0 - ox, ox - bx1), 0.0f);
        const float ey = sa::fmax_nn(sa::fmax_nn(by0 - oy, oy - by1), 0.0f);
        const float ez = sa::fmax_nn(sa::fmax_nn(bz0 - oz, oz - bz1), 0.0f);
        const float lb = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
        const bool need = lb * kSkipMargin < best;
        const bool touched = __ballot(need) != 0ull;
        if (touched) {
            if (need) {
                float nb = -1.0f;                   // tf_sampling_g.cu:141
                int nj = 0;
#pragma unroll
                for (int j = 0; j < kPPT; ++j) {
                    const float dx = X[j] - ox, dy = Y[j] - oy, dz = Z[j] - oz;
                    const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));   // as fps.hip
                    const float t2 = sa::fmin_nn(d, TD[j]);
                    TD[j] = t2;
                    const bool g = t2 > nb;         // strict: the smallest tie key among equal maxima
                    nb = g ? t2 : nb;
                    nj = g ? j : nj;
                }
                best = nb;
                bj = nj;
            }
        }
        if (touched || first) {
            // ---- this wave's candidate: maximum td, ties by tie key
            first = false;
            const float Mw = sa::wave_allmax(best);
            const unsigned long long cand = __ballot(best == Mw);
            unsigned mykey = kNoKey;
            if (best == Mw) mykey = s_sorted[tid * kPPT + bj];
            const unsigned kmin = sa::wave_allmin_u32(mykey);
            const int wl = __builtin_ctzll(__ballot(mykey == kmin && best == Mw) | (cand == 0ull ? 1ull : 0ull));
            float cx = X[0], cy = Y[0], cz = Z[0];
#pragma unroll
            for (int j = 1; j < kPPT; ++j) {
                const bool sel = bj == j;
                cx = sel ? X[j] : cx; cy = sel ? Y[j] : cy; cz = sel ? Z[j] : cz;
            }
            pubM = Mw;
            pubKey = kmin;
            pubX = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cx), wl));
            pubY = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cy), wl));
            pubZ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cz), wl));
        }
        const int par = it & 1;
        if (lane == 0) {
            s_val[par][w] = pubM;
            s_key[par][w] = pubKey;
            s_pt[par][w] = make_float4(pubX, pubY, pubZ, 0.0f);
        }
        __syncthreads();
        // ---- cross-wave: maximum value, then minimum tie key among the waves that hold it
        const float v = s_val[par][lane & (kBW - 1)];
        const unsigned kq = s_key[par][lane & (kBW - 1)];
        const float M = sa::row16_allmax(v);
        const unsigned kmn = sa::row16_allmin_u32(v == M ? kq : kNoKey);
        const unsigned long long eq = __ballot(v == M && kq == kmn);
        const int ws = __builtin_ctzll(eq) & (kBW - 1);
        const float4 wp = s_pt[par][ws];
        ox = wp.x; oy = wp.y; oz = wp.z;
        if (tid == 0) o[it] = (int)tie_key_index(__builtin_amdgcn_readfirstlane(kmn)) + idx_off;
    }
}

}  // namespace

// D-FPS on coordinates with bucket culling; same contract as sa_fps_ex with c == 3, n <= 16384.
extern "C" int sa_fps_bucket_ex(int b, int n, int m, const float *inp, int *out, int out_stride, int idx_off,
                                hipStream_t stream) {
    if (b <= 0 || n <= 0 || n > kCap || m <= 0 || !inp || !out || out_stride < m) return SA_ERR_INVALID;
    hipLaunchKernelGGL(fps3_bucket_kernel, dim3(b), dim3(kBThreads), 0, stream, n, m, inp, out, out_stride,
                       idx_off);
    SA_CHECK_LAUNCH();
    return SA_OK;
}
