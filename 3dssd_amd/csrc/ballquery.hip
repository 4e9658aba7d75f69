// Ball query (plain and dilated) for gfx950, all radius bands of an SA layer in ONE pass.
//
// Reference: lib/utils/tf_ops/grouping/tf_grouping_g.cu:215-255 (query_ball_point_gpu) and :308-357
// (query_ball_point_dilated_gpu): one thread per query scans the n points serially and keeps the
// FIRST nsample hits in index order; an SA layer launches it once per radius band.
//
// Here a wave64 scans 64 points per step for one query: hits are compacted in index order with
// __ballot + prefix popcount (same "first nsample" result as the serial scan), each wave keeps a
// chunk of points in VGPRs and reuses it for kQW queries, and the 2-3 bands of a layer (same xyz,
// same centres, layers_util.py:134-147) share one distance evaluation per pair.
//
// Distance: d2 = fma(dz,dz, fma(dx,dx, dy*dy)) (decision B of oracle/sa_oracle.c).  The reference
// compares sqrtf(d2) with the radii; sqrtf is monotone and correctly rounded, so each comparison
// is replaced by an exactly equivalent comparison of d2 with a host-computed threshold
// T(r) = min{x >= 0 : sqrtf(x) >= r}:   sqrtf(d2) < r  <=>  d2 < T(r),   sqrtf(d2) >= r  <=>  d2 >= T(r).
#include <math.h>

#include "sa_common.h"

namespace {

constexpr int kMaxBands = 4;
#ifndef SA_BQ_QW
#define SA_BQ_QW 8
#endif
#ifndef SA_BQ_CH
#define SA_BQ_CH 8
#endif
#ifndef SA_BQ_PREFETCH
#define SA_BQ_PREFETCH 1
#endif
constexpr int kQWmax = SA_BQ_QW;     // queries per wave of the throughput form (a small-problem form uses 2, see the launcher)
constexpr int kWavesPerWG = 4;
constexpr int kCH = SA_BQ_CH;        // 64-point steps held in registers per chunk
constexpr int kRow = 64;      // max nsample of the fused kernel

struct Bands {
    float tlo[kMaxBands];   // hit needs d2 >= tlo (dilated only)
    float thi[kMaxBands];   // hit needs d2 <  thi
    int ns[kMaxBands];
    int *idx[kMaxBands];    // [b,m,ns_i]
    int *cnt[kMaxBands];    // [b,m]
    float thi_max;
    int nbands;
    int dilated;            // 1: d2 == 0 is always a hit (tf_grouping_g.cu:337)
};

template <int kQW>
__global__ __launch_bounds__(kWavesPerWG * 64) void ball_query_kernel(
    int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2, Bands B) {
    __shared__ int s_rows[kWavesPerWG][kQW][kMaxBands][kRow];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q0 = (blockIdx.x * kWavesPerWG + w) * kQW;   // first query of this wave
    if (q0 >= m) return;                                   // whole wave out of range (no barriers used)
    const float *P = xyz1 + (size_t)b * n * 3;
    const float *C = xyz2 + ((size_t)b * m + q0) * 3;
    const int nq = min(kQW, m - q0);

    // lane q < nq holds centre q
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (lane < nq) { cx = C[lane * 3 + 0]; cy = C[lane * 3 + 1]; cz = C[lane * 3 + 2]; }
    int cntv = 0;   // lane q*kMaxBands+i holds the hit count of (query q, band i): read/written with
                    // v_readlane / a lane select, no LDS round trip on the scan path
    unsigned active = (1u << nq) - 1u;                     // queries with at least one band not full

#if SA_BQ_PREFETCH
    // register chunk of 64*kCH points, double buffered: chunk c+1 is requested before chunk c is scanned
    float x1[kCH], y1[kCH], z1[kCH], xn[kCH], yn[kCH], zn[kCH];
#pragma unroll
    for (int s = 0; s < kCH; ++s) {
        const int k = s * 64 + lane;
        const int kk = k < n ? k : n - 1;
        xn[s] = P[kk * 3 + 0]; yn[s] = P[kk * 3 + 1]; zn[s] = P[kk * 3 + 2];
    }
    for (int base = 0; base < n && active != 0u; base += 64 * kCH) {
#pragma unroll
        for (int s = 0; s < kCH; ++s) { x1[s] = xn[s]; y1[s] = yn[s]; z1[s] = zn[s]; }
        if (base + 64 * kCH < n) {
#pragma unroll
            for (int s = 0; s < kCH; ++s) {
                const int k = base + 64 * kCH + s * 64 + lane;
                const int kk = k < n ? k : n - 1;
                xn[s] = P[kk * 3 + 0]; yn[s] = P[kk * 3 + 1]; zn[s] = P[kk * 3 + 2];
            }
        }
#else
    float x1[kCH], y1[kCH], z1[kCH];
    for (int base = 0; base < n && active != 0u; base += 64 * kCH) {
#pragma unroll
        for (int s = 0; s < kCH; ++s) {
            const int k = base + s * 64 + lane;
            const int kk = k < n ? k : n - 1;
            x1[s] = P[kk * 3 + 0]; y1[s] = P[kk * 3 + 1]; z1[s] = P[kk * 3 + 2];
        }
#endif
        for (int q = 0; q < nq; ++q) {
            if (!((active >> q) & 1u)) continue;
            const float x2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cx), q));
            const float y2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cy), q));
            const float z2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cz), q));
            // distances of this query to the whole register chunk; one wave-wide test decides whether any
            // of the 512 points can be a hit at all (d2 == 0 < thi_max is covered by the same test)
            float d2[kCH];
            float dmin = 3.0e38f;
#pragma unroll
            for (int s = 0; s < kCH; ++s) {
                const float dx = x2 - x1[s], dy = y2 - y1[s], dz = z2 - z1[s];
                d2[s] = __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
                dmin = sa::fmin_nn(dmin, d2[s]);
            }
            if (__ballot(dmin < B.thi_max) == 0ull) continue;
#pragma unroll
            for (int s = 0; s < kCH; ++s) {
                const int k = base + s * 64 + lane;
                const bool valid = k < n;
                const bool near = valid && (d2[s] < B.thi_max);
                if (__ballot(near) == 0ull) continue;
                bool all_full = true;
#pragma unroll
                for (int i = 0; i < kMaxBands; ++i) {
                    if (i >= B.nbands) break;
                    int c = __builtin_amdgcn_readlane(cntv, q * kMaxBands + i);
                    const int nsi = B.ns[i];
                    if (c >= nsi) continue;
                    bool hit = valid && (B.dilated ? (d2[s] == 0.0f || (d2[s] >= B.tlo[i] && d2[s] < B.thi[i]))
                                                   : (d2[s] < B.thi[i]));
                    unsigned long long hm = __ballot(hit);
                    if (hm != 0ull) {
                        int pos = c + __popcll(hm & ((1ull << lane) - 1ull));
                        if (hit && pos < nsi) s_rows[w][q][i][pos] = k;
                        c = min(nsi, c + (int)__popcll(hm));
                        cntv = (lane == q * kMaxBands + i) ? c : cntv;
                    }
                    all_full = all_full && (c >= nsi);
                }
                if (all_full) { active &= ~(1u << q); break; }   // tf_grouping_g.cu:237-239
            }
        }
    }
    // rows: slot l < cnt keeps its hit, slots >= cnt repeat the first hit (tf_grouping_g.cu:245-248),
    // empty balls are zero-filled (oracle decision D).
    for (int q = 0; q < nq; ++q) {
#pragma unroll
        for (int i = 0; i < kMaxBands; ++i) {
            if (i >= B.nbands) break;
            const int c = __builtin_amdgcn_readlane(cntv, q * kMaxBands + i);
            const int nsi = B.ns[i];
            const size_t qi = (size_t)b * m + q0 + q;
            if (lane < nsi) {
                int v = 0;
                if (c > 0) v = s_rows[w][q][i][lane < c ? lane : 0];
                B.idx[i][qi * nsi + lane] = v;
            }
            if (lane == 0) B.cnt[i][qi] = c;
        }
    }
}

// Small frames (n <= 2048: layer3 / layer4 of the backbone, 1 024 / 512 points): ONE query per wave with the whole frame
// in registers (NS steps of 64 points per lane), four times the waves of the chunked form above and no serial part
// beyond the query itself.  The distances of a query are evaluated once, branch-free; a band is then NS ballots, a
// running scalar count and one predicated store per step straight into the output row (hit h of the scan lands at slot
// rank(h) = hits before it, the same "first nsample in index order" as the serial scan; slots >= cnt repeat the first
// hit, tf_grouping_g.cu:245-248).  No LDS, no cross-lane traffic except the ballots.
template <int NS>
__global__ __launch_bounds__(kWavesPerWG * 64) void ball_query_small_kernel(
    int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2, Bands B) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * kWavesPerWG + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (q >= m) return;
    const float *P = xyz1 + (size_t)b * n * 3;
    float x1[NS], y1[NS], z1[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = s * 64 + lane;
        const int kk = k < n ? k : n - 1;
        x1[s] = P[kk * 3 + 0]; y1[s] = P[kk * 3 + 1]; z1[s] = P[kk * 3 + 2];
    }
    const size_t qi = (size_t)b * m + q;
    const float x2 = xyz2[qi * 3 + 0], y2 = xyz2[qi * 3 + 1], z2 = xyz2[qi * 3 + 2];
    float d2[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float dx = x2 - x1[s], dy = y2 - y1[s], dz = z2 - z1[s];
        d2[s] = __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
        if (s * 64 + lane >= n) d2[s] = __builtin_inff();            // never a hit: not 0, not below any threshold
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int i = 0; i < B.nbands; ++i) {
        // one predicate for both forms: dilated  d2 == 0 || (tlo <= d2 < thi)  ==  d2 < thi && (d2 >= tlo || d2 == 0)
        // because thi > 0 always (rmax > 0); plain  d2 < thi  is the same with tlo = -inf
        const float tlo = B.dilated ? B.tlo[i] : -__builtin_inff(), thi = B.thi[i];
        const int nsi = B.ns[i];
        int *row = B.idx[i] + qi * nsi;
        int c = 0, first = 0;
#pragma unroll
        for (int g = 0; g < NS / 4; ++g) {
            if (c < nsi) {                                               // tf_grouping_g.cu:237-239 (uniform)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int s = g * 4 + u;
                    const bool hit = (d2[s] < thi) & ((d2[s] >= tlo) | (d2[s] == 0.0f));
                    const unsigned long long hm = __ballot(hit);
                    const int pos = c + (int)__popcll(hm & below);
                    if (hit && pos < nsi) row[pos] = s * 64 + lane;
                    first = (c == 0 && hm != 0ull) ? s * 64 + (int)__builtin_ctzll(hm) : first;
                    c += (int)__popcll(hm);
                }
            }
        }
        c = c < nsi ? c : nsi;
        if (lane >= c && lane < nsi) row[lane] = first;               // first == 0 for an empty ball (oracle decision D)
        if (lane == 0) B.cnt[i][qi] = c;
    }
}

// Fallback for nsample > 64: the reference's own shape, one thread per query, serial scan.
__global__ void ball_query_serial_kernel(int b, int n, int m, float tlo, float thi, int dilated,
                                         int nsample, const float *__restrict__ xyz1,
                                         const float *__restrict__ xyz2, int *__restrict__ idx,
                                         int *__restrict__ pts_cnt) {
    const long total = (long)b * m;
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < total;
         q += (long)gridDim.x * blockDim.x) {
        const int bi = (int)(q / m);
        const float *P = xyz1 + (size_t)bi * n * 3;
        const float x2 = xyz2[q * 3 + 0], y2 = xyz2[q * 3 + 1], z2 = xyz2[q * 3 + 2];
        int *ci = idx + (size_t)q * nsample;
        int cnt = 0;
        for (int k = 0; k < n && cnt < nsample; ++k) {
            const float dx = x2 - P[k * 3 + 0], dy = y2 - P[k * 3 + 1], dz = z2 - P[k * 3 + 2];
            const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
            const bool hit = dilated ? (d2 == 0.0f || (d2 >= tlo && d2 < thi)) : (d2 < thi);
            if (hit) {
                if (cnt == 0)
                    for (int l = 0; l < nsample; ++l) ci[l] = k;
                ci[cnt++] = k;
            }
        }
        if (cnt == 0)
            for (int l = 0; l < nsample; ++l) ci[l] = 0;
        pts_cnt[q] = cnt;
    }
}

// query_ball_point_withidx (tf_grouping_g.cu:259-304): the scan visits the points in the order sort_idx[q, :]
// gives (an argsort from the caller) instead of index order.  One wave per query, 64 candidates per step.
__global__ __launch_bounds__(256) void ball_query_withidx_kernel(int n, int m, long total, float thi, int nsample,
                                                                 const float *__restrict__ xyz1,
                                                                 const float *__restrict__ xyz2,
                                                                 const int *__restrict__ sort_idx,
                                                                 int *__restrict__ idx, int *__restrict__ pts_cnt) {
    const int lane = threadIdx.x & 63;
    const long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= total) return;
    const float *P = xyz1 + (size_t)(q / m) * n * 3;
    const int *order = sort_idx + (size_t)q * n;
    const float cx = xyz2[q * 3 + 0], cy = xyz2[q * 3 + 1], cz = xyz2[q * 3 + 2];
    int *o = idx + (size_t)q * nsample;
    int cnt = 0, first = 0;
    for (int i0 = 0; i0 < n && cnt < nsample; i0 += 64) {
        const int i = i0 + lane;
        bool in = false;
        int k = 0;
        if (i < n) {
            k = order[i];
            const float dx = cx - P[k * 3 + 0], dy = cy - P[k * 3 + 1], dz = cz - P[k * 3 + 2];
            in = __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy)) < thi;      // decision B; d < radius
        }
        const unsigned long long hit = __ballot(in);
        if (hit) {
            if (cnt == 0) first = __builtin_amdgcn_readlane(k, __builtin_ctzll(hit));
            const int pos = cnt + __builtin_popcountll(hit & ((1ull << lane) - 1ull));
            if (in && pos < nsample) o[pos] = k;
            cnt += __builtin_popcountll(hit);
        }
    }
    cnt = cnt < nsample ? cnt : nsample;
    for (int s = cnt + lane; s < nsample; s += 64) o[s] = first;    // empty ball: zeros (decision D)
    if (lane == 0) pts_cnt[q] = cnt;
}

// T(r) = min{x >= 0 : sqrtf(x) >= r}; +inf if no float qualifies.
float sqrt_ge_threshold(float r) {
    if (!(r > 0.0f)) return 0.0f;
    float x = r * r;
    if (isinf(x)) return sqrtf(3.402823466e+38f) >= r ? 3.402823466e+38f : INFINITY;
    while (sqrtf(x) < r) x = nextafterf(x, INFINITY);
    while (x > 0.0f && sqrtf(nextafterf(x, 0.0f)) >= r) x = nextafterf(x, 0.0f);
    return x;
}

}  // namespace

// Fused entry: all bands of one SA layer.  rmin/rmax/ns are host arrays of length nbands;
// idx[i] is [b,m,ns[i]] and cnt[i] is [b,m] (device).  dilated=0 ignores rmin
// (query_ball_point: hit iff max(sqrt(d2),1e-20) < rmax, tf_grouping_g.cu:243-244).
extern "C" int sa_query_ball_point_multi(int b, int n, int m, int nbands, const float *rmin,
                                         const float *rmax, const int *ns, int dilated,
                                         const float *xyz1, const float *xyz2, int *const *idx,
                                         int *const *cnt, hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || nbands <= 0 || !xyz1 || !xyz2 || !idx || !cnt) return SA_ERR_INVALID;
    bool fused_ok = nbands <= kMaxBands;
    for (int i = 0; i < nbands; ++i) {
        if (ns[i] <= 0 || !(rmax[i] > 0.0f)) return SA_ERR_INVALID;   // tf_grouping.cpp:274-278
        if (dilated && rmin[i] < 0.0f) return SA_ERR_INVALID;         // tf_grouping.cpp:368
        if (ns[i] > kRow) fused_ok = false;
    }
    if (fused_ok) {
        Bands B;
        B.nbands = nbands;
        B.dilated = dilated ? 1 : 0;
        B.thi_max = 0.0f;
        for (int i = 0; i < kMaxBands; ++i) {
            const bool on = i < nbands;
            B.tlo[i] = on && dilated ? sqrt_ge_threshold(rmin[i]) : 0.0f;
            // non-dilated: max(d,1e-20f) < r is never true for r <= 1e-20f
            B.thi[i] = on ? ((!dilated && rmax[i] <= 1e-20f) ? 0.0f : sqrt_ge_threshold(rmax[i])) : 0.0f;
            B.ns[i] = on ? ns[i] : 0;
            B.idx[i] = on ? idx[i] : nullptr;
            B.cnt[i] = on ? cnt[i] : nullptr;
            if (on && B.thi[i] > B.thi_max) B.thi_max = B.thi[i];
        }
        // Queries per wave: 8 amortise a register chunk of points over many queries (the throughput form); the layers
        // that take this kernel in the backbone are small (512 / 256 centres per frame: 64 waves per frame at 8), where
        // the launch is one dependent scan per wave -- 2 queries per wave give 4x the waves and a quarter of the scan
        // (layer3 0.068 -> 0.025 ms, layer4 0.032 -> 0.013 ms).
        static const bool small_form = SA_KNOB("SA_BQ_SMALL", 1) != 0;
        if (small_form && n <= 2048) {                       // one query per wave, the frame in registers
            dim3 grid((m + kWavesPerWG - 1) / kWavesPerWG, b);
            if (n <= 512) hipLaunchKernelGGL(ball_query_small_kernel<8>, grid, dim3(kWavesPerWG * 64), 0, stream, n, m, xyz1, xyz2, B);
            else if (n <= 1024) hipLaunchKernelGGL(ball_query_small_kernel<16>, grid, dim3(kWavesPerWG * 64), 0, stream, n, m, xyz1, xyz2, B);
            else hipLaunchKernelGGL(ball_query_small_kernel<32>, grid, dim3(kWavesPerWG * 64), 0, stream, n, m, xyz1, xyz2, B);
            SA_CHECK_LAUNCH();
            return SA_OK;
        }
        static const int cus = [] { int d = 0; hipDeviceProp_t pr; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
        const bool small = (long)b * ((m + kQWmax - 1) / kQWmax) < 16l * cus;
        const int qpw = (small ? 2 : kQWmax) * kWavesPerWG;
        dim3 grid((m + qpw - 1) / qpw, b);
        if (small) hipLaunchKernelGGL(ball_query_kernel<2>, grid, dim3(kWavesPerWG * 64), 0, stream, n, m, xyz1, xyz2, B);
        else hipLaunchKernelGGL(ball_query_kernel<kQWmax>, grid, dim3(kWavesPerWG * 64), 0, stream, n, m, xyz1, xyz2, B);
        SA_CHECK_LAUNCH();
    } else {
        for (int i = 0; i < nbands; ++i) {
            const float tlo = dilated ? sqrt_ge_threshold(rmin[i]) : 0.0f;
            const float thi = (!dilated && rmax[i] <= 1e-20f) ? 0.0f : sqrt_ge_threshold(rmax[i]);
            const long total = (long)b * m;
            const int grid = (int)((total + 63) / 64 < 2048 ? (total + 63) / 64 : 2048);
            hipLaunchKernelGGL(ball_query_serial_kernel, dim3(grid), dim3(64), 0, stream, b, n, m, tlo, thi,
                               dilated ? 1 : 0, ns[i], xyz1, xyz2, idx[i], cnt[i]);
            SA_CHECK_LAUNCH();
        }
    }
    return SA_OK;
}

// Reference launcher signatures (lib/utils/tf_ops/grouping/tf_grouping.cpp:270,363) + stream.
extern "C" int sa_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1,
                                   const float *xyz2, int *idx, int *pts_cnt, hipStream_t stream) {
    const float rmin = 0.0f;
    return sa_query_ball_point_multi(b, n, m, 1, &rmin, &radius, &nsample, 0, xyz1, xyz2, &idx, &pts_cnt,
                                     stream);
}
extern "C" int sa_query_ball_point_dilated(int b, int n, int m, float min_radius, float max_radius,
                                           int nsample, const float *xyz1, const float *xyz2, int *idx,
                                           int *pts_cnt, hipStream_t stream) {
    return sa_query_ball_point_multi(b, n, m, 1, &min_radius, &max_radius, &nsample, 1, xyz1, xyz2, &idx,
                                     &pts_cnt, stream);
}

// queryBallPointWithidxLauncher(b,n,m,radius,nsample,xyz1,xyz2,sort_idx,idx,pts_cnt) -- tf_grouping.cpp:314.
// sort_idx [b,m,n]: visiting order of the n points for every query.
extern "C" int sa_query_ball_point_withidx(int b, int n, int m, float radius, int nsample, const float *xyz1,
                                           const float *xyz2, const int *sort_idx, int *idx, int *pts_cnt,
                                           hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0 || !(radius > 0.0f) || !xyz1 || !xyz2 || !sort_idx || !idx || !pts_cnt)
        return SA_ERR_INVALID;
    const long total = (long)b * m;
    if ((total + 3) / 4 > 0x7FFFFFFF || (long)n * 3 > 0x7FFFFFFF) return SA_ERR_UNSUPPORTED;
    // hit iff max(sqrtf(d2), 1e-20f) < radius  <=>  d2 < T(radius) for radius > 1e-20 (never otherwise)
    const float thi = radius <= 1e-20f ? 0.0f : sqrt_ge_threshold(radius);
    hipLaunchKernelGGL(ball_query_withidx_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, stream, n, m, total,
                       thi, nsample, xyz1, xyz2, sort_idx, idx, pts_cnt);
    SA_CHECK_LAUNCH();
    return SA_OK;
}
