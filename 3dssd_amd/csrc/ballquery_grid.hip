// Ball query through a uniform x-z grid: same outputs as ballquery.hip (first nsample hits in index order,
// padding with the first hit, zero rows for empty balls), ~10x fewer distance evaluations on large frames.
//
// The brute-force kernel evaluates all n x m pairs because "the first nsample in index order" looks inherently
// serial.  It is not: the result of a band is the set of the nsample SMALLEST indices among the points inside
// the band.  So: (1) per frame, bucket the points into grid cells at least r_max wide (one workgroup: LDS
// histogram, scan, scatter -- order inside a cell is irrelevant); (2) per query, visit the 3 x 3 cells around it,
// evaluate exactly the same d2 and thresholds as ballquery.hip on those candidates only, collect the hits of
// every band in LDS, and place hit h at output slot rank(h) = #{hits with a smaller index} if rank < nsample.
// Round 6, the SORTING form (bq_grid_sort_kernel, further down; the default whenever sum(nsample) <= 192): a pass with one
// THREAD per query writes the query's record (centre, the three candidate ranges of its 3 x 3 cells); the query's wave
// walks the candidates in blocks of four 64-candidate steps, appends the hits of ANY band to one key list
// (index << 4 | band mask), sorts that list once in registers (bitonic network over the wave) and reads every band's slot
// as the prefix count of its mask bit.  A list about to overflow (near the sensor, `dense` frames) is cut to the keys that
// can still be output -- the nsample-th smallest index of a full band, found by bisection on wave ballots, becomes the
// band's admission bound for the rest of the walk.
// The LIST form of rounds 4-5 (bq_grid_query_kernel: per-band LDS lists, ranking by v_readlane loops) stays for larger
// nsample sums: a band with more hits than its LDS list holds (kCap) is cut down IN PLACE to its nsample smallest entries
// (bisection on wave ballots) and that value becomes the band's admission bound; pts_cnt = min(total, nsample) is kept
// separately.  (Rounds 2-3 redid such a query by an ordered full scan of all n points: 17x slower on dense frames.)
//
// Cell geometry: cell = clamp(int((coord - min) * inv), 0, kNX-1) with cell size >= r_max * (1 + 1e-4): monotone in
// the coordinate, so |dx| <= r_max implies a cell difference of at most 1 (the margin absorbs the fp32 rounding of
// the product), and clamping only merges cells.  y is not gridded (LiDAR frames are flat); it enters through d2.
#include <math.h>
#include <string.h>

#include "sa_common.h"

namespace {

constexpr int kNX = 128;                 // grid is kNX x kNX cells
constexpr int kNC = kNX * kNX;
constexpr int kMaxBands = 4;
constexpr int kCap = 320;                // hits per band kept in LDS per query (more: cut down to the nsample smallest, in place); nsample <= kCap - 64 = 256 on this path
constexpr int kQWaves = 4;               // waves (= queries in flight) per workgroup

struct GBands {
    float tlo[kMaxBands], thi[kMaxBands];
    int ns[kMaxBands];
    int *idx[kMaxBands];
    int *cnt[kMaxBands];
    float thi_max;
    int nbands, dilated;
    unsigned blo[kMaxBands], bwd[kMaxBands];     // the dilated band test on the bits of d2 (bq_grid_sort_kernel): bits(tlo), bits(thi) - bits(tlo)
};

__device__ __forceinline__ float wave_allmin_f(float x) { return -sa::wave_allmax(-x); }

// workspace per frame (in ints): cell_start[kNC + 1] (+3 padding) | sorted[n] float4 = (x, y, z, index bits) in cell
// order | params[4] floats (minx, minz, inv, cell size) | one float4 "point" at 1e30 (sorted[n + 1]: what the lanes past a
// query's last candidate read -- its distance is +inf, no band takes it; round 6).  Round 5: the cell lists hold the POINTS, not their indices -- a query's
// chain was bounds -> sorted index -> point (a scattered 12-byte gather); it is now bounds -> one coalesced 16-byte read.
constexpr int kCellInts = kNC + 4;
constexpr int kPrepInts = 12;            // dwords of a query's record (bq_grid_prep_kernel)
__host__ __device__ __forceinline__ size_t ws_stride(int n) { return (size_t)kCellInts + 4 * (size_t)n + 8; }

// reuse != 0: the workspace is stated to hold the grid of these very points from an earlier call; it is kept if its cells
// are at least cell_min wide (params[3] = the cell size it was built with), rebuilt otherwise -- decided here, on the
// device, because the cell size depends on the frame's extent.
// prep != nullptr (round 6): the workgroup also writes the records of its frame's m queries (kPrepInts dwords each, see
// bq_grid_prep_kernel) straight from the cell starts it holds in LDS -- one launch less per call.
__global__ __launch_bounds__(1024) void bq_grid_build_kernel(int n, float cell_min, int reuse, const float *__restrict__ xyz1,
                                                             int *__restrict__ ws, int m, const float *__restrict__ xyz2,
                                                             int *__restrict__ prep) {
    __shared__ int s_cnt[kNC];
    __shared__ float s_red[4][16];
    __shared__ int s_wsum[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float *p = xyz1 + (size_t)b * n * 3;
    int *cell_start = ws + (size_t)b * ws_stride(n);
    float4 *sorted = (float4 *)(cell_start + kCellInts);
    float *params = (float *)(sorted + n);
    if (reuse && params[3] >= cell_min) return;          // (the same word for every thread of the workgroup)

    // Frames of at most 16384 points (every layer of 3dssd.yaml): a thread keeps its 16 points in registers for all passes,
    // and the cell lists leave through LDS (round 6).  The direct scatter below wrote 16 bytes per point to a random place of
    // the frame's 256 KB list: 45 of the kernel's 51 us on spread-out frames (5 us on `dense` ones, whose few cells make the
    // same writes consecutive).  Here every point's position is taken first (the same LDS cursors), then the list is
    // assembled in LDS a quarter (4096 points = the 64 KB the counters no longer need) at a time and copied out in whole
    // 1 KB pieces.  The order INSIDE a cell may differ from the scatter's (both are arbitrary: atomics); no reader depends on it.
    const bool small = n <= 16 * 1024;
    float qx[16], qy[16], qz[16];
    if (small) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = min(tid + u * 1024, n - 1);
            qx[u] = p[k * 3 + 0]; qy[u] = p[k * 3 + 1]; qz[u] = p[k * 3 + 2];
        }
    }
    float mnx = 3e38f, mxx = -3e38f, mnz = 3e38f, mxz = -3e38f;
    if (small) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {                         // (a repeated point n - 1 changes no extreme)
            mnx = sa::fmin_nn(mnx, qx[u]); mxx = sa::fmax_nn(mxx, qx[u]);
            mnz = sa::fmin_nn(mnz, qz[u]); mxz = sa::fmax_nn(mxz, qz[u]);
        }
    } else {
        for (int k = tid; k < n; k += 1024) {
            const float x = p[k * 3 + 0], z = p[k * 3 + 2];
            mnx = sa::fmin_nn(mnx, x); mxx = sa::fmax_nn(mxx, x);
            mnz = sa::fmin_nn(mnz, z); mxz = sa::fmax_nn(mxz, z);
        }
    }
    mnx = wave_allmin_f(mnx); mxx = sa::wave_allmax(mxx);
    mnz = wave_allmin_f(mnz); mxz = sa::wave_allmax(mxz);
    if (lane == 0) { s_red[0][w] = mnx; s_red[1][w] = mxx; s_red[2][w] = mnz; s_red[3][w] = mxz; }
    for (int i = tid; i < kNC; i += 1024) s_cnt[i] = 0;
    __syncthreads();
    mnx = s_red[0][0]; mxx = s_red[1][0]; mnz = s_red[2][0]; mxz = s_red[3][0];
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        mnx = sa::fmin_nn(mnx, s_red[0][i]); mxx = sa::fmax_nn(mxx, s_red[1][i]);
        mnz = sa::fmin_nn(mnz, s_red[2][i]); mxz = sa::fmax_nn(mxz, s_red[3][i]);
    }
    // cell size: at least cell_min (= r_max with margin), and large enough that the frame spans <= kNX-2 cells
    float cs = sa::fmax_nn(cell_min, sa::fmax_nn(mxx - mnx, mxz - mnz) / (float)(kNX - 2));
    cs = sa::fmax_nn(cs, 1e-20f);
    const float inv = 1.0f / cs;
    if (tid == 0) {
        params[0] = mnx; params[1] = mnz; params[2] = inv; params[3] = cs;
        sorted[n + 1] = make_float4(1e30f, 1e30f, 1e30f, 0.0f);
    }

    int cell[16];
    if (small) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int cx = min(kNX - 1, max(0, (int)((qx[u] - mnx) * inv)));
            const int cz = min(kNX - 1, max(0, (int)((qz[u] - mnz) * inv)));
            cell[u] = cz * kNX + cx;
            if (tid + u * 1024 < n) atomicAdd(&s_cnt[cell[u]], 1);
        }
    } else {
        for (int k = tid; k < n; k += 1024) {
            const int cx = min(kNX - 1, max(0, (int)((p[k * 3 + 0] - mnx) * inv)));
            const int cz = min(kNX - 1, max(0, (int)((p[k * 3 + 2] - mnz) * inv)));
            atomicAdd(&s_cnt[cz * kNX + cx], 1);
        }
    }
    __syncthreads();
    // exclusive scan of the 16384 counters: 16 consecutive cells per thread
    int loc[16], sum = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { loc[i] = s_cnt[tid * 16 + i]; sum += loc[i]; }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
    if (lane == 63) s_wsum[w] = incl;
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < w; ++i) woff += s_wsum[i];
    int run = woff + incl - sum;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        cell_start[tid * 16 + i] = run;
        s_cnt[tid * 16 + i] = run;          // becomes the scatter cursor
        run += loc[i];
    }
    if (tid == 1023) cell_start[kNC] = run;
    __syncthreads();
    if (prep) {                                            // s_cnt holds the cell starts (and cell kNC starts at n)
        for (int q = tid; q < m; q += 1024) {
            const size_t qi = (size_t)b * m + q;
            const float x2 = xyz2[qi * 3 + 0], y2 = xyz2[qi * 3 + 1], z2 = xyz2[qi * 3 + 2];
            const int cx = min(kNX - 1, max(0, (int)((x2 - mnx) * inv)));
            const int cz = min(kNX - 1, max(0, (int)((z2 - mnz) * inv)));
            const int x_lo = max(cx - 1, 0), x_hi = min(cx + 1, kNX - 1);
            int rs[3], rc[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int iz = cz - 1 + r;
                const bool in = iz >= 0 && iz < kNX;
                const int izc = in ? iz : cz;
                const int ie = izc * kNX + x_hi + 1;
                const int st = s_cnt[izc * kNX + x_lo], e = ie < kNC ? s_cnt[ie] : n;
                rs[r] = st;
                rc[r] = in ? e - st : 0;
            }
            const int c01 = rc[0] + rc[1], T = c01 + rc[2];
            int4 *o = (int4 *)(prep + qi * kPrepInts);
            o[0] = make_int4(__float_as_int(x2), __float_as_int(y2), __float_as_int(z2), rs[0]);
            o[1] = make_int4(rs[1] - rc[0], rs[2] - c01, rc[0], c01);
            o[2] = make_int4(T, 0, 0, 0);
        }
        __syncthreads();                                   // (the cursors move next)
    }
    if (small) {
        int pos[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) pos[u] = tid + u * 1024 < n ? atomicAdd(&s_cnt[cell[u]], 1) : -1;
        __syncthreads();                                       // every cursor was read: the counters' 64 KB become the staging buffer
        float4 *stage = (float4 *)s_cnt;
        for (int q = 0; q * 4096 < n; ++q) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if ((pos[u] >> 12) == q) stage[pos[u] & 4095] = make_float4(qx[u], qy[u], qz[u], __int_as_float(tid + u * 1024));
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int at = q * 4096 + i * 1024 + tid;
                if (at < n) sorted[at] = stage[i * 1024 + tid];
            }
            __syncthreads();
        }
        return;
    }
    for (int k = tid; k < n; k += 1024) {
        const int cx = min(kNX - 1, max(0, (int)((p[k * 3 + 0] - mnx) * inv)));
        const int cz = min(kNX - 1, max(0, (int)((p[k * 3 + 2] - mnz) * inv)));
        const int pos = atomicAdd(&s_cnt[cz * kNX + cx], 1);
        sorted[pos] = make_float4(p[k * 3 + 0], p[k * 3 + 1], p[k * 3 + 2], __int_as_float(k));
    }
}

// list[0..H) (H <= kCap, distinct point indices) -> its `nth` smallest value (1 <= nth <= H).  One wave; every lane
// holds up to kCap / 64 entries; bisection on the value range with wave ballots (<= 31 rounds, usually ~14).
__device__ __forceinline__ int select_nth(const int *list, int H, int nth, int lane) {
    int v[kCap / 64];
    int mn = 0x7FFFFFFF, mx = 0;
#pragma unroll
    for (int s = 0; s < kCap / 64; ++s) {
        const int h = s * 64 + lane;
        v[s] = h < H ? list[h] : 0x7FFFFFFF;
        mn = min(mn, v[s]);
        mx = max(mx, h < H ? v[s] : 0);
    }
    int lo = (int)sa::wave_allmin_u32((unsigned)mn);
    int hi = (int)~sa::wave_allmin_u32(~(unsigned)mx);
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int s = 0; s < kCap / 64; ++s) c += (int)__popcll(__ballot(v[s] <= mid));
        if (c >= nth) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Cut list[0..H) down to its `keep` smallest entries (H > keep), compacted to list[0..keep); returns the largest kept
// value (the admission bound from now on).
__device__ __forceinline__ int keep_smallest(int *list, int H, int keep, int lane) {
    const int tau = select_nth(list, H, keep, lane);
    int v[kCap / 64];
#pragma unroll
    for (int s = 0; s < kCap / 64; ++s) {
        const int h = s * 64 + lane;
        v[s] = h < H ? list[h] : 0x7FFFFFFF;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int base = 0;
#pragma unroll
    for (int s = 0; s < kCap / 64; ++s) {
        const bool k = v[s] <= tau;
        const unsigned long long mk = __ballot(k);
        if (k) list[base + (int)__popcll(mk & ((1ull << lane) - 1ull))] = v[s];
        base += (int)__popcll(mk);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return tau;
}

// NB / DIL: number of bands and the dilated form as COMPILE-TIME constants (round 5).  The kernel is bound by the CU's one
// scalar pipe (PMC at the layer-1 shape: 1 070 scalar + 1 000 vector instructions per query on ring-structured frames,
// 0.65 scalar instructions per cycle and CU): every run-time band test and form select was scalar work per band and step,
// and so is the per-band bookkeeping (ballot -> branch -> popcount -> list append -> overflow test) when it runs for every
// band at every 64-candidate step.  So a step only APPENDS its hits -- of any band -- to ONE list per query, as
// `index | band mask << 27` (band membership is vector arithmetic on the lane's own d2: no ballots), and the per-band
// lists with their in-place cut are fed from that list in batches: when it is nearly full, and at the end of the walk.
// A query with 54 hits among 290 candidates (ring-structured frames) runs the per-band code once per band instead of
// 4.5 times, a sparse one (1-10 hits) skips it for the bands that got nothing.
constexpr int kListCap = 256;            // entries of the per-query hit list (flushed into the band lists beyond kListCap - 64)
constexpr int kMaskShift = 27;           // host: n <= 2^27

template <int NB, bool DIL>
__global__ __launch_bounds__(kQWaves * 64, 8) void bq_grid_query_kernel(int n, int m, const float *__restrict__ xyz1,
                                                                     const float *__restrict__ xyz2,
                                                                     const int *__restrict__ ws, GBands B) {
    __shared__ int s_hits[kQWaves][NB][kCap];
    __shared__ int s_list[kQWaves][kListCap];
    int b = blockIdx.y, bx = blockIdx.x;
    if ((gridDim.x & 7) == 0 && (gridDim.y & 7) == 0) {
        // XCD-aware (block L is observed to run on XCD L % 8, sa_common.h): the queries of frame f run on XCD f % 8, whose L2
        // then holds that frame's points, cell table and sorted list alone
        const unsigned L = blockIdx.y * gridDim.x + blockIdx.x, G8 = 8u * gridDim.x;
        b = (int)(8u * (L / G8) + (L & 7u));
        bx = (int)((L % G8) >> 3);
    }
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // wave-uniform: the per-query loads are scalar
    const int *cell_start = ws + (size_t)b * ws_stride(n);
    const float4 *sorted = (const float4 *)(cell_start + kCellInts);
    const float *params = (const float *)(sorted + n);
    const float mnx = params[0], mnz = params[1], inv = params[2];
    int (*hits)[kCap] = s_hits[w];
    int *list = s_list[w];

    for (int q = bx * kQWaves + w; q < m; q += gridDim.x * kQWaves) {
        const size_t qi = (size_t)b * m + q;
        const float x2 = xyz2[qi * 3 + 0], y2 = xyz2[qi * 3 + 1], z2 = xyz2[qi * 3 + 2];
        const int cx = min(kNX - 1, max(0, (int)((x2 - mnx) * inv)));
        const int cz = min(kNX - 1, max(0, (int)((z2 - mnz) * inv)));
        const int x_lo = max(cx - 1, 0), x_hi = min(cx + 1, kNX - 1);
        int cnts[NB], tot[NB], tau[NB];            // entries in the band's LDS list | hits so far (pts_cnt = min(tot, nsample)) | admission bound once a list was cut
#pragma unroll
        for (int i = 0; i < NB; ++i) { cnts[i] = 0; tot[i] = 0; tau[i] = 0x7FFFFFFF; }
        int nlist = 0;                             // entries in the per-query hit list
        // the hit list -> the per-band lists: band i takes the entries that carry its bit, 64 per step, with the same
        // bookkeeping the walk used to do per candidate step (total count, admission bound, in-place cut when nearly full)
        auto flush = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int c0 = 0; c0 < nlist; c0 += 64) {
                const bool live = c0 + lane < nlist;
                const int e = live ? list[c0 + lane] : 0;
                const int k = e & ((1 << kMaskShift) - 1);
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const bool hit = live && ((e >> (kMaskShift + i)) & 1) != 0;
                    const unsigned long long hall = __ballot(hit);
                    if (hall != 0ull) {
                        tot[i] += (int)__popcll(hall);
                        const bool take = hit && k <= tau[i];              // beyond the bound: cannot be among the nsample smallest
                        const unsigned long long hm = __ballot(take);
                        const int at = cnts[i] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
                        if (take) hits[i][at] = k;                         // at < kCap: the list had >= 64 free entries
                        cnts[i] += (int)__popcll(hm);
                        if (cnts[i] > kCap - 64) {                         // no room for another step: keep the nsample smallest
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            tau[i] = keep_smallest(hits[i], cnts[i], B.ns[i], lane);
                            cnts[i] = B.ns[i];
                        }
                    }
                }
            }
            nlist = 0;
        };
        // the three z-rows of the 3 x 3 neighbourhood are three index ranges of `sorted` (3 cells each, contiguous
        // in x).  Their bounds are fetched together and the candidates are walked as ONE flattened list, 64 per
        // step: per query the dependent chain is bounds -> cell list, once, instead of once per row.
        int rs[3], rc[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iz = cz - 1 + r;
            const bool in = iz >= 0 && iz < kNX;
            const int izc = in ? iz : cz;
            const int s = cell_start[izc * kNX + x_lo], e = cell_start[izc * kNX + x_hi + 1];
            rs[r] = s;
            rc[r] = in ? e - s : 0;
        }
        const int c01 = rc[0] + rc[1], T = c01 + rc[2];
        // a step reads its 64 candidates (point + index, 16 bytes each, consecutive in the cell lists) with ONE coalesced
        // load, requested one step ahead of its use
        auto cand = [&](int base) -> float4 {              // lane's candidate of the step at `base` (clamped past T)
            const int j = base + lane;
            const int jj = j < T ? j : 0;
            const int pos = jj < rc[0] ? rs[0] + jj : (jj < c01 ? rs[1] + (jj - rc[0]) : rs[2] + (jj - c01));
            return sorted[pos];
        };
        float4 nxt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (T > 0) nxt = cand(0);
        for (int base = 0; base < T; base += 64) {
            const bool valid = base + lane < T;
            const float4 cur = nxt;
            if (base + 64 < T) nxt = cand(base + 64);       // (wave-uniform: a one-step query requests nothing ahead)
            const int k = __float_as_int(cur.w);
            const float dx = x2 - cur.x, dy = y2 - cur.y, dz = z2 - cur.z;
            const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));   // as ballquery.hip
            // band membership of this lane's candidate as a bit mask: the same comparisons as ballquery.hip, as 0 / 1 integers
            unsigned mask = 0u;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                unsigned bit;
                if (DIL) bit = ((d2 >= B.tlo[i] ? 1u : 0u) & (d2 < B.thi[i] ? 1u : 0u)) | (d2 == 0.0f ? 1u : 0u);
                else bit = d2 < B.thi[i] ? 1u : 0u;
                mask |= bit << i;
            }
            mask = valid ? mask : 0u;
            const unsigned long long hm = __ballot(mask != 0u);
            if (hm == 0ull) continue;
            const int at = nlist + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
            if (mask != 0u) list[at] = k | (int)(mask << kMaskShift);           // at < kListCap: the list had >= 64 free entries
            nlist += (int)__popcll(hm);
            if (nlist > kListCap - 64) flush();
        }
        flush();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int nsi = B.ns[i];
            if (cnts[i] > nsi) {                                           // more entries than outputs: cut to the smallest
                keep_smallest(hits[i], cnts[i], nsi, lane);
                cnts[i] = nsi;
            }
            const int H = cnts[i];
            const int c = min(tot[i], nsi);                                // == H
            int *row = B.idx[i] + qi * nsi;
            if (H == 0) {
                for (int l = lane; l < nsi; l += 64) row[l] = 0;           // empty ball: zero row
            } else {
                // rank of every hit among the hits of this band; the nsample smallest indices fill slots 0..c-1
                int kmin = 0x7FFFFFFF;
                if (H <= 64) {
                    // the usual case: one hit per lane, the others' values broadcast through v_readlane (an SGPR)
                    const int k = lane < H ? hits[i][lane] : 0x7FFFFFFF;
                    int rank = 0;
                    for (int j = 0; j < H; ++j) rank += __builtin_amdgcn_readlane(k, j) < k ? 1 : 0;
                    if (lane < H && rank < nsi) row[rank] = k;
                    kmin = k;
                } else {
                    for (int h0 = 0; h0 < H; h0 += 64) {
                        const int h = h0 + lane;
                        const int k = h < H ? hits[i][h] : 0x7FFFFFFF;
                        int rank = 0;
                        for (int j = 0; j < H; ++j) rank += hits[i][j] < k ? 1 : 0;
                        if (h < H && rank < nsi) row[rank] = k;
                        kmin = min(kmin, k);
                    }
                }
                kmin = (int)sa::wave_allmin_u32((unsigned)kmin);
                for (int l = c + lane; l < nsi; l += 64) row[l] = kmin;     // padding: the first hit
            }
            if (lane == 0) B.cnt[i][qi] = c;
        }
    }
}

// ---- round 6: the SORTING form of the query kernel -------------------------------------------------------------------
// The kernel above spends ~900 of its ~1 000 vector instructions per query (ring-structured frames: 290 candidates, 54 hits)
// on bookkeeping: the hits go from the per-query list into per-band lists (ballot -> count -> append per band and
// 64-entry chunk), and every band then RANKS its hits with a serial loop (v_readlane + compare + add-with-carry + loop
// control per hit: 8 instructions, 5 of them on the scalar side).  Here the query keeps ONE list of keys
// `index << 4 | band mask`, and when the walk is over the list is SORTED once, in registers, by a bitonic network over the
// wave (63 vector instructions for 64 keys whatever the bands; only the phases the entry count needs; up to 256 keys in
// four registers per lane).  In index order a band's output slot is a prefix count of its mask bit (ballot + mbcnt): no
// ranking loop, no per-band lists (15 KB of LDS per workgroup -> 4), no scalar work per hit.  A list about to overflow
// (> 192 keys; near the sensor, or `dense` frames) is cut to the keys that can still be output: a band holding nsample keys
// or more keeps its nsample smallest -- their largest index, found by bisection on wave ballots, is the band's admission
// bound from then on, as in the list form.  Needs sum(nsample) <= 192 (3dssd.yaml: 128) and 32-bit row offsets; the kernel
// above stays for the rest.  Measured at the layer-1 shape (128 frames; default / rings64, `dense` per 32 frames), list form
// 366 / 1 093 / 1 571 us: sorting 290 / 841 / 1 439; + the set-up pass, 32-bit offsets, rows through an LDS compaction,
// one-hot band test 222 / 774 / 1 282; + consecutive queries per wave, blocks of four steps 200 / 723 / 1 052; + cuts by
// bisection instead of a full sort 194 / 683 / 986 (profiles/r06_ballquery_sort_ab.txt).
constexpr int kSortCap = 256;              // keys per query list = 4 per lane
constexpr int kKeyShift = 4;               // key = index << 4 | band mask (kMaxBands bits)
constexpr unsigned kKeySentinel = 0xFFFFFFF0u;     // sorts behind every key, belongs to no band

template <int BIT> __device__ __forceinline__ constexpr unsigned long long lane_bit_mask() {
    return BIT == 0 ? 0xAAAAAAAAAAAAAAAAull : BIT == 1 ? 0xCCCCCCCCCCCCCCCCull : BIT == 2 ? 0xF0F0F0F0F0F0F0F0ull
         : BIT == 3 ? 0xFF00FF00FF00FF00ull : BIT == 4 ? 0xFFFF0000FFFF0000ull : 0xFFFFFFFF00000000ull;
}
// compare-exchange with the partner's value p: lanes whose bit BIT is clear keep the smaller, the others the larger
// (v_min_u32_dpp + v_max_u32_dpp + v_cndmask on an SGPR-pair constant when p is a DPP move of x)
template <int BIT> __device__ __forceinline__ unsigned key_ce(unsigned x, unsigned p) {
    const unsigned lo = x < p ? x : p, hi = x < p ? p : x;
    return __builtin_amdgcn_inverse_ballot_w64(lane_bit_mask<BIT>()) ? hi : lo;
}
template <int CTRL> __device__ __forceinline__ unsigned key_dpp(unsigned x) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}
template <int PATTERN> __device__ __forceinline__ unsigned key_swz(unsigned x) {
    return (unsigned)__builtin_amdgcn_ds_swizzle((int)x, PATTERN);
}
// DPP controls: quad_perm [1,0,3,2] = 0xB1 (lane ^ 1), [2,3,0,1] = 0x4E (lane ^ 2), [3,2,1,0] = 0x1B (mirror of 4);
// row_half_mirror 0x141 (mirror of 8), row_mirror 0x140 (mirror of 16), row_ror:8 0x128 (lane ^ 8).  ds_swizzle bit
// mode and | or << 5 | xor << 10: lane ^ 4 = 0x101F, lane ^ 16 = 0x401F, mirror of 32 (lane ^ 31) = 0x7C1F.
__device__ __forceinline__ unsigned key_tail2(unsigned x) {            // the half-cleaners of distance 2, 1
    x = key_ce<1>(x, key_dpp<0x4E>(x));
    return key_ce<0>(x, key_dpp<0xB1>(x));
}
// 64 keys, one per lane (lanes >= U hold the sentinel), ascending by lane.  Bitonic network in the "mirror" form (the
// first step of a merge phase compares i with its mirror image inside the block: every exchange is ascending); a phase
// over blocks of B lanes is only needed when more than B / 2 lanes hold keys.  mir64: (63 - lane) << 2.
__device__ __forceinline__ unsigned key_sort64(unsigned x, int U, unsigned mir64) {
    if (U > 1) x = key_ce<0>(x, key_dpp<0xB1>(x));
    if (U > 2) { x = key_ce<1>(x, key_dpp<0x1B>(x)); x = key_ce<0>(x, key_dpp<0xB1>(x)); }
    if (U > 4) { x = key_ce<2>(x, key_dpp<0x141>(x)); x = key_tail2(x); }
    if (U > 8) { x = key_ce<3>(x, key_dpp<0x140>(x)); x = key_ce<2>(x, key_swz<0x101F>(x)); x = key_tail2(x); }
    if (U > 16) {
        x = key_ce<4>(x, key_swz<0x7C1F>(x));
        x = key_ce<3>(x, key_dpp<0x128>(x)); x = key_ce<2>(x, key_swz<0x101F>(x)); x = key_tail2(x);
    }
    if (U > 32) {
        x = key_ce<5>(x, (unsigned)__builtin_amdgcn_ds_bpermute((int)mir64, (int)x));
        x = key_ce<4>(x, key_swz<0x401F>(x));
        x = key_ce<3>(x, key_dpp<0x128>(x)); x = key_ce<2>(x, key_swz<0x101F>(x)); x = key_tail2(x);
    }
    return x;
}
// a BITONIC sequence of 64 keys -> ascending: half-cleaners of distance 32 (v_permlane32_swap: [0] = the lower half's
// keys in both halves, [1] = the upper half's), 16, 8, 4, 2, 1
__device__ __forceinline__ unsigned key_clean64(unsigned x) {
    const auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    const unsigned a = sw[0], b = sw[1];
    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    x = __builtin_amdgcn_inverse_ballot_w64(lane_bit_mask<5>()) ? hi : lo;
    x = key_ce<4>(x, key_swz<0x401F>(x));
    x = key_ce<3>(x, key_dpp<0x128>(x)); x = key_ce<2>(x, key_swz<0x101F>(x));
    return key_tail2(x);
}
// two ascending registers -> ascending over (a, b)
__device__ __forceinline__ void key_merge2(unsigned &a, unsigned &b, unsigned mir64) {
    const unsigned bm = (unsigned)__builtin_amdgcn_ds_bpermute((int)mir64, (int)b);
    const unsigned lo = a < bm ? a : bm, hi = a < bm ? bm : a;
    a = key_clean64(lo); b = key_clean64(hi);
}
// v[0..4): the first U entries of `list` (U <= kSortCap) ascending over (register, lane), sentinels behind them
__device__ __forceinline__ void key_sort_list(const unsigned *list, int U, int lane, unsigned mir64, unsigned (&v)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = kKeySentinel;
    if (lane < U) v[0] = list[lane];
    v[0] = key_sort64(v[0], U, mir64);
    if (U <= 64) return;
    if (64 + lane < U) v[1] = list[64 + lane];
    v[1] = key_sort64(v[1], U - 64, mir64);
    key_merge2(v[0], v[1], mir64);
    if (U <= 128) return;
    if (128 + lane < U) v[2] = list[128 + lane];
    v[2] = key_sort64(v[2], U - 128, mir64);
    if (U > 192) {
        if (192 + lane < U) v[3] = list[192 + lane];
        v[3] = key_sort64(v[3], U - 192, mir64);
    }
    key_merge2(v[2], v[3], mir64);
    // (v0, v1) and (v2, v3) ascending: against the reversed second pair, then two bitonic sequences of 128
    const unsigned m3 = (unsigned)__builtin_amdgcn_ds_bpermute((int)mir64, (int)v[3]);
    const unsigned m2 = (unsigned)__builtin_amdgcn_ds_bpermute((int)mir64, (int)v[2]);
    const unsigned l0 = v[0] < m3 ? v[0] : m3, h0 = v[0] < m3 ? m3 : v[0];
    const unsigned l1 = v[1] < m2 ? v[1] : m2, h1 = v[1] < m2 ? m2 : v[1];
    v[0] = key_clean64(l0 < l1 ? l0 : l1); v[1] = key_clean64(l0 < l1 ? l1 : l0);
    v[2] = key_clean64(h0 < h1 ? h0 : h1); v[3] = key_clean64(h0 < h1 ? h1 : h0);
}

// candidate j of a query's flattened 3 x 3 neighbourhood (three index ranges of `sorted`; the far point past T): its byte offset
__device__ __forceinline__ unsigned key_cand_off(int j, int rc0, int c01, int T, int offA, int offB, int offC, int sent) {
    int off = j >= rc0 ? offB : offA;
    off = j >= c01 ? offC : off;
    const int pos = j < T ? j + off : sent;
    return (unsigned)pos * 16u;
}
// Per-query set-up as a pass of its own (one THREAD per query): the centre's cell, the bounds of the three z-rows of its
// 3 x 3 neighbourhood and the offsets that flatten them into one candidate list -- 70 scalar + 10 vector instructions and
// two dependent scalar-load round trips per query when the query's WAVE did them (the kernel is bound by instruction
// issue: 0.87 of the scalar pipe on sparse frames).  Record (12 dwords, behind the frames' grids in the workspace):
//   x, y, z | offA, offB, offC (candidate j lives at sorted[j + off], off by range) | rc0, c01, T (range ends) | 3 unused
__global__ __launch_bounds__(256) void bq_grid_prep_kernel(int n, int m, const float *__restrict__ xyz2,
                                                           const int *__restrict__ ws, int *__restrict__ prep) {
    const int b = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    if (q >= m) return;
    const int *cell_start = ws + (size_t)b * ws_stride(n);
    const float *params = (const float *)(cell_start + kCellInts + 4 * (size_t)n);
    const float mnx = params[0], mnz = params[1], inv = params[2];
    const size_t qi = (size_t)b * m + q;
    const float x2 = xyz2[qi * 3 + 0], y2 = xyz2[qi * 3 + 1], z2 = xyz2[qi * 3 + 2];
    const int cx = min(kNX - 1, max(0, (int)((x2 - mnx) * inv)));
    const int cz = min(kNX - 1, max(0, (int)((z2 - mnz) * inv)));
    const int x_lo = max(cx - 1, 0), x_hi = min(cx + 1, kNX - 1);
    int rs[3], rc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int iz = cz - 1 + r;
        const bool in = iz >= 0 && iz < kNX;
        const int izc = in ? iz : cz;
        const int s = cell_start[izc * kNX + x_lo], e = cell_start[izc * kNX + x_hi + 1];
        rs[r] = s;
        rc[r] = in ? e - s : 0;
    }
    const int c01 = rc[0] + rc[1], T = c01 + rc[2];
    int4 *o = (int4 *)(prep + qi * kPrepInts);
    o[0] = make_int4(__float_as_int(x2), __float_as_int(y2), __float_as_int(z2), rs[0]);
    o[1] = make_int4(rs[1] - rc[0], rs[2] - c01, rc[0], c01);
    o[2] = make_int4(T, 0, 0, 0);
}

// CONTIG (dilated only): the bands are [0, t1), [t1, t2), ... -- the reference's dilated groups -- so a candidate's mask is
// one-hot: start at band 0 and move up once per threshold passed (two instructions per threshold instead of three per band)
template <int NB, bool DIL, bool CONTIG>
__global__ __launch_bounds__(kQWaves * 64, 8) void bq_grid_sort_kernel(int n, int m, int qpw, const int *__restrict__ ws,
                                                                    const int *__restrict__ prep, GBands B) {
    __shared__ unsigned s_keys[kQWaves][kSortCap];
    int b = blockIdx.y, bx = blockIdx.x;
    if ((gridDim.x & 7) == 0 && (gridDim.y & 7) == 0) {                 // XCD-aware, as above
        const unsigned L = blockIdx.y * gridDim.x + blockIdx.x, G8 = 8u * gridDim.x;
        b = (int)(8u * (L / G8) + (L & 7u));
        bx = (int)((L % G8) >> 3);
    }
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char *sorted = (const char *)(ws + (size_t)b * ws_stride(n) + kCellInts);
    unsigned *list = s_keys[w];
    const unsigned mir64 = (unsigned)(63 - lane) << 2;
    // band tests on the BITS of d2 (d2 >= +0 or NaN: unsigned order = float order, a NaN passes no test either way):
    // dilated: tlo <= d2 < thi  <=>  bits(d2) - bits(tlo) < bits(thi) - bits(tlo) (unsigned; empty when thi <= tlo)
    // (B.blo / B.bwd, from the host)
    constexpr unsigned kAll = (1u << NB) - 1u;
    const int sent = n + 1;                               // float4 index of the far "point" behind the params
    // (host: b * m * max(kPrepInts, nsample) * 4 < 2^32 -- query records and output rows are addressed with 32-bit offsets)
    const unsigned q0 = (unsigned)b * (unsigned)m;

    // a wave takes qpw CONSECUTIVE queries: their counts leave as one store per band and 64 queries (a 4-byte store per
    // query and band was 3 of the 6 store instructions of a query)
    const int q_first = (bx * kQWaves + w) * qpw, q_end = min(q_first + qpw, m);
    for (int qs0 = q_first; qs0 < q_end; qs0 += 64) {
    int cntv[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) cntv[i] = 0;
    const int nq = min(64, q_end - qs0);
    for (int t = 0; t < nq; ++t) {
        const unsigned qi = q0 + (unsigned)(qs0 + t);
        const int4 *rec = (const int4 *)((const char *)prep + qi * (unsigned)(kPrepInts * 4));
        const int4 r0 = rec[0], r1 = rec[1];
        const int T = rec[2].x;
        asm volatile("" ::"s"(r0.x), "s"(r1.x), "s"(T));        // one wait for the whole record (hipcc sinks the first two loads behind a test of T)
        const float x2 = __int_as_float(r0.x), y2 = __int_as_float(r0.y), z2 = __int_as_float(r0.z);
        const int offA = r0.w, offB = r1.x, offC = r1.y, rc0 = r1.z, c01 = r1.w;
        int tau[NB];                                       // admission bound of a band whose first nsample keys are known
#pragma unroll
        for (int i = 0; i < NB; ++i) tau[i] = 0x7FFFFFFF;
        bool filtered = false;
        int nlist = 0;
        // The list is about to overflow (193..256 keys): a band that holds nsample keys or more keeps its nsample smallest --
        // their largest index, found by BISECTION on the index range with wave ballots (4 compares + ~14 scalar instructions
        // per round: a full sort of the list is 430 vector instructions, and `dense` frames cut 3-4 times per query), is the
        // band's admission bound from now on; keys left without a band are dropped.  The list stays unsorted.
        auto cut = [&]() {
            __builtin_amdgcn_wave_barrier();
            unsigned v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = r * 64 + lane < nlist ? list[r * 64 + lane] : kKeySentinel;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int nsi = B.ns[i];
                unsigned kb[4];                              // the band's keys as indices, 0xFFFFFFFF elsewhere
                int have = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool inb = ((v[r] >> i) & 1u) != 0u;
                    kb[r] = inb ? v[r] >> kKeyShift : 0xFFFFFFFFu;
                    have += (int)__popcll(__ballot(inb));
                }
                if (have < nsi) continue;                    // every key of the band can still be output
                unsigned lo = 0u, hi = (unsigned)n - 1u;     // the nsi-th smallest index lies in [lo, hi]
                while (lo < hi) {
                    const unsigned mid = lo + ((hi - lo) >> 1);
                    int c = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) c += (int)__popcll(__ballot(kb[r] <= mid));
                    if (c >= nsi) hi = mid; else lo = mid + 1u;
                }
                tau[i] = (int)lo;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = kb[r] != 0xFFFFFFFFu && kb[r] > lo ? v[r] & ~(1u << i) : v[r];
            }
            __builtin_amdgcn_wave_barrier();
            int cb = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool keep = (v[r] & ((1u << kKeyShift) - 1u)) != 0u;      // (the sentinel carries no band)
                const unsigned long long bal = __ballot(keep);
                const int pos = cb + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                if (keep) list[pos] = v[r];
                cb += (int)__popcll(bal);
            }
            nlist = cb;
            filtered = true;
            __builtin_amdgcn_wave_barrier();
        };

        // one step = 64 candidates: distances, band bits, the hits appended to the key list
        auto step = [&](const float4 &cur) {
            const int k = __float_as_int(cur.w);
            const float dx = x2 - cur.x, dy = y2 - cur.y, dz = z2 - cur.z;
            const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));   // as ballquery.hip
            const unsigned u = __float_as_uint(d2);
            unsigned mask = 0u;
            if (CONTIG) {
                // one-hot: the band the candidate lies in, and -- once lists were cut -- THAT band's admission bound with it
                mask = 1u;
                // (readfirstlane: the bounds as VALUES -- a select between the array's elements makes hipcc move the array to LDS)
                int tsel = __builtin_amdgcn_readfirstlane(tau[0]);
#pragma unroll
                for (int i = 1; i < NB; ++i) {
                    const bool up = u >= B.blo[i];
                    const int ti = __builtin_amdgcn_readfirstlane(tau[i]);
                    mask = up ? (1u << i) : mask;
                    if (filtered) tsel = up ? ti : tsel;
                }
                mask = u >= B.blo[NB - 1] + B.bwd[NB - 1] ? 0u : mask;
                if (filtered) mask = k > tsel ? 0u : mask;
                mask = u == 0u ? kAll : mask;                 // the centre itself: every band (never filtered: admitting a key is always safe)
            } else {
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    if (DIL) mask |= (u - B.blo[i]) < B.bwd[i] ? (1u << i) : 0u;
                    else mask |= d2 < B.thi[i] ? (1u << i) : 0u;
                }
                if (DIL) mask = u == 0u ? kAll : mask;
                if (filtered) {
#pragma unroll
                    for (int i = 0; i < NB; ++i) mask = k > tau[i] ? (mask & ~(1u << i)) : mask;
                }
            }
            const bool hit = mask != 0u;
            const unsigned long long hm = __ballot(hit);
            if (hm != 0ull) {
                const int at = nlist + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
                if (hit) list[at] = ((unsigned)k << kKeyShift) | mask;                    // at < kSortCap: >= 64 entries were free
                nlist += (int)__popcll(hm);
            }
        };
        // The walk, in blocks of up to four steps whose candidates are requested together (a step's ~35 instructions do
        // not cover a load's round trip).  Wherever the next 256 candidates lie inside ONE of the three ranges (near the
        // sensor; `dense` frames: thousands of candidates) they are four loads off one address (immediate offsets) and the
        // steps need no range select, bounds test or far-point select.  A list about to overflow is cut at the top of
        // the loop (ONE site) and the walk resumes where it stopped.
        // (SA_BQ_DBG_NOCUT / _NOSORT / _NOOUT: ablation switches of variant builds, tools/build_variant.sh -- results are wrong
        //  with any of them; profiles/r06_ballquery_sort_ab.txt has what each stage costs.)
        int base = 0;
        if (T <= 64) {                                        // sparse frames: one step, no loop (and no cut: <= 64 keys)
            step(*(const float4 *)(sorted + key_cand_off(lane, rc0, c01, T, offA, offB, offC, sent)));
            base = 64;
        }
        while (base < T) {
#ifdef SA_BQ_DBG_NOCUT
            if (nlist > kSortCap - 64) nlist = kSortCap - 64;
#else
            if (nlist > kSortCap - 64) cut();
#endif
            const int off = base >= c01 ? offC : (base >= rc0 ? offB : offA);
            const int end = base >= c01 ? T : (base >= rc0 ? c01 : rc0);
            // byte offsets of the four steps' candidates first, then ONE unconditional set of loads (loads under tests, or one
            // set per form, made hipcc wait for a load where it was issued to copy it into the registers the paths share);
            // steps past T read the far point and are not evaluated
            unsigned o0, o1, o2, o3;
            if (base + 256 <= end) {
                o0 = (unsigned)(base + lane + off) * 16u;
                o1 = o0 + 1024u; o2 = o0 + 2048u; o3 = o0 + 3072u;
            } else {
                o0 = key_cand_off(base + lane, rc0, c01, T, offA, offB, offC, sent);
                o1 = key_cand_off(base + 64 + lane, rc0, c01, T, offA, offB, offC, sent);
                o2 = key_cand_off(base + 128 + lane, rc0, c01, T, offA, offB, offC, sent);
                o3 = key_cand_off(base + 192 + lane, rc0, c01, T, offA, offB, offC, sent);
            }
            const float4 a0 = *(const float4 *)(sorted + o0), a1 = *(const float4 *)(sorted + o1);
            const float4 a2 = *(const float4 *)(sorted + o2), a3 = *(const float4 *)(sorted + o3);
            step(a0); base += 64;
            if (base >= T || nlist > kSortCap - 64) continue;
            step(a1); base += 64;
            if (base >= T || nlist > kSortCap - 64) continue;
            step(a2); base += 64;
            if (base >= T || nlist > kSortCap - 64) continue;
            step(a3); base += 64;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- the keys in index order; a band's output slot = the number of its keys in front
        unsigned v[4];
#ifdef SA_BQ_DBG_NOSORT
        for (int r = 0; r < 4; ++r) v[r] = r * 64 + lane < nlist ? list[r * 64 + lane] : kKeySentinel;
#else
        key_sort_list(list, nlist, lane, mir64, v);
#endif
        const int R = (nlist + 63) >> 6;
#ifdef SA_BQ_DBG_NOOUT
        cntv[0] = lane == t ? (int)(v[0] + v[1] + v[2] + v[3]) + R : cntv[0];
        continue;
#endif
        // a band's keys are compacted through the (now free) LDS list -- slot = prefix count of its bit -- and its row leaves
        // as whole 256-byte pieces: lane l writes entry l, or the first hit behind the last one (0 for an empty ball)
        int *ilist = (int *)list;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int nsi = B.ns[i];
            int base = 0;
            auto place = [&](unsigned key) {
                const bool inb = ((key >> i) & 1u) != 0u;
                const unsigned long long bal = __ballot(inb);
                if (bal != 0ull) {
                    const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    if (inb && slot < nsi) ilist[slot] = (int)(key >> kKeyShift);
                    base += (int)__popcll(bal);
                }
            };
            place(v[0]);
            if (R > 1) {
                place(v[1]);
                if (R > 2) { place(v[2]); place(v[3]); }
            }
            const int c = min(base, nsi);
            if (c == 0 && lane == 0) ilist[0] = 0;
            __builtin_amdgcn_wave_barrier();
            char *row = (char *)B.idx[i];
            const unsigned rbase = qi * (unsigned)nsi;
#pragma unroll
            for (int p = 0; p < (kSortCap - 64) / 64; ++p) {
                if (p * 64 >= nsi) break;
                const int l = p * 64 + lane;
                if (l < nsi) *(int *)(row + ((rbase + (unsigned)l) << 2)) = ilist[l < c ? l : 0];
            }
            cntv[i] = lane == t ? c : cntv[i];
            __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
        if (lane < nq) *(int *)((char *)B.cnt[i] + ((q0 + (unsigned)(qs0 + lane)) << 2)) = cntv[i];
    }
}

float sqrt_ge_threshold_g(float r) {
    if (!(r > 0.0f)) return 0.0f;
    float x = r * r;
    if (isinf(x)) return sqrtf(3.402823466e+38f) >= r ? 3.402823466e+38f : INFINITY;
    while (sqrtf(x) < r) x = nextafterf(x, INFINITY);
    while (x > 0.0f && sqrtf(nextafterf(x, 0.0f)) >= r) x = nextafterf(x, 0.0f);
    return x;
}

}  // namespace

// Bytes of device workspace sa_query_ball_point_grid needs for (b, n, m).
extern "C" size_t sa_query_ball_point_grid_ws_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    return ((size_t)b * ws_stride(n) + (size_t)b * m * kPrepInts) * sizeof(int);      // the frames' grids | the queries' records
}

extern "C" int sa_query_ball_point_multi(int b, int n, int m, int nbands, const float *rmin, const float *rmax,
                                         const int *ns, int dilated, const float *xyz1, const float *xyz2,
                                         int *const *idx, int *const *cnt, hipStream_t stream);

// Same contract as sa_query_ball_point_multi (all bands of one SA layer), through the grid.  `workspace` is
// caller-owned device memory of sa_query_ball_point_grid_ws_bytes(b, n, m) bytes.  nbands <= 4.
// flags bit 0: `workspace` still holds the grid an earlier call of this function built over the SAME xyz1 contents (same b, n)
// and that call is ordered before this one: the build is skipped when that grid's cells are wide enough for these radii
// (checked on the device), e.g. the per-band calls of the reference's stand-alone ops over one point set.
extern "C" int sa_query_ball_point_grid_ex(int b, int n, int m, int nbands, const float *rmin, const float *rmax,
                                           const int *ns, int dilated, const float *xyz1, const float *xyz2,
                                           int *const *idx, int *const *cnt, void *workspace, int flags, hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || nbands <= 0 || nbands > kMaxBands || !xyz1 || !xyz2 || !idx || !cnt || !workspace)
        return SA_ERR_INVALID;
    // the per-query hit lists (LDS) hold kCap entries per band and must keep nsample of them plus one step of 64:
    // larger nsample goes to the plain scan kernels (found by tests/fuzz_ops.py: nsample = 300 rows were cut at 256)
    for (int i = 0; i < nbands; ++i)
        if (ns[i] > kCap - 64 || n > (1 << kMaskShift)) return sa_query_ball_point_multi(b, n, m, nbands, rmin, rmax, ns, dilated, xyz1, xyz2, idx, cnt, stream);
    GBands B;
    B.nbands = nbands;
    B.dilated = dilated ? 1 : 0;
    B.thi_max = 0.0f;
    float rmax_all = 0.0f;
    for (int i = 0; i < kMaxBands; ++i) {
        const bool on = i < nbands;
        if (on && (ns[i] <= 0 || !(rmax[i] > 0.0f))) return SA_ERR_INVALID;
        if (on && dilated && rmin[i] < 0.0f) return SA_ERR_INVALID;
        B.tlo[i] = on && dilated ? sqrt_ge_threshold_g(rmin[i]) : 0.0f;
        B.thi[i] = on ? ((!dilated && rmax[i] <= 1e-20f) ? 0.0f : sqrt_ge_threshold_g(rmax[i])) : 0.0f;
        B.ns[i] = on ? ns[i] : 0;
        {
            unsigned lo_bits, hi_bits;
            memcpy(&lo_bits, &B.tlo[i], 4); memcpy(&hi_bits, &B.thi[i], 4);
            B.blo[i] = lo_bits;
            B.bwd[i] = B.thi[i] > B.tlo[i] ? hi_bits - lo_bits : 0u;
        }
        B.idx[i] = on ? idx[i] : nullptr;
        B.cnt[i] = on ? cnt[i] : nullptr;
        if (on && B.thi[i] > B.thi_max) B.thi_max = B.thi[i];
        if (on && rmax[i] > rmax_all) rmax_all = rmax[i];
    }
    const float cell_min = rmax_all * 1.0001f + 1e-6f;
    // (launched further down: with the sorting form and no kept grid it also writes the queries' records)
    // queries per wave: a wave's set-up (block -> frame mapping, the frame's grid parameters: a dependent scalar load) is
    // paid once per wave; with enough work to fill the chip several times over, a wave takes kQPW queries
    static const int qpw = SA_KNOB("SA_BQ_QPW", 8);
    const long waves_total = (long)b * ((m + kQWaves - 1) / kQWaves) * kQWaves;
    int per_wave = qpw;                                                   // ... while >= 4 rounds of the chip's 8192 wave slots remain
    while (per_wave > 1 && waves_total / per_wave < 4l * 8192) per_wave >>= 1;
    int gx = (m + kQWaves * per_wave - 1) / (kQWaves * per_wave);
    if (gx > 4096) gx = 4096;
    if (gx > 8) gx = (gx + 7) & ~7;                                       // multiples of 8: the XCD-aware frame mapping
    // the sorting form (round 6) when the bands' nsample fit its key list after a cut; SA_BQ_SORT=0: the list form, for A/B runs
#ifdef SA_BQ_FORCE_LIST
    static const bool sort_on = false;
#else
    static const bool sort_on = SA_KNOB("SA_BQ_SORT", 1) != 0;
#endif
    int ns_sum = 0;
    for (int i = 0; i < nbands; ++i) ns_sum += ns[i];
    int ns_max = kPrepInts;
    for (int i = 0; i < nbands; ++i) ns_max = ns[i] > ns_max ? ns[i] : ns_max;
    const bool sorting = sort_on && ns_sum <= kSortCap - 64 && (size_t)b * m * ns_max * 4 < ((size_t)1 << 32);
    int *prep = (int *)workspace + (size_t)b * ws_stride(n);
    // a kept grid (flags bit 0) may or may not be rebuilt -- decided on the device -- so its queries' records come from the
    // pass of their own; otherwise the build writes them
    const bool fused_prep = sorting && !(flags & 1);
    hipLaunchKernelGGL(bq_grid_build_kernel, dim3(b), dim3(1024), 0, stream, n, cell_min, flags & 1, xyz1, (int *)workspace, m, xyz2,
                       fused_prep ? prep : (int *)nullptr);
    SA_CHECK_LAUNCH();
    if (sorting && !fused_prep) {
        hipLaunchKernelGGL(bq_grid_prep_kernel, dim3((m + 255) / 256, b), dim3(256), 0, stream, n, m, xyz2, (const int *)workspace, prep);
        SA_CHECK_LAUNCH();
    }
    bool contig = dilated != 0 && B.blo[0] == 0u;            // bands [0, t1), [t1, t2), ...: the one-hot band test
    for (int i = 0; i < nbands; ++i) {
        contig = contig && B.bwd[i] != 0u;
        if (i > 0) contig = contig && B.blo[i] == B.blo[i - 1] + B.bwd[i - 1];
    }
    const int sort_qpw = (m + gx * kQWaves - 1) / (gx * kQWaves);           // consecutive queries per wave (sorting form)
#define SA_BQ_LAUNCH(NB_, DIL_)                                                                                         \
    do {                                                                                                                \
        if (sorting && DIL_ && contig)                                                                                  \
            hipLaunchKernelGGL((bq_grid_sort_kernel<NB_, DIL_, DIL_>), dim3(gx, b), dim3(kQWaves * 64), 0, stream, n,   \
                               m, sort_qpw, (const int *)workspace, (const int *)prep, B);                              \
        else if (sorting)                                                                                               \
            hipLaunchKernelGGL((bq_grid_sort_kernel<NB_, DIL_, false>), dim3(gx, b), dim3(kQWaves * 64), 0, stream, n,  \
                               m, sort_qpw, (const int *)workspace, (const int *)prep, B);                              \
        else                                                                                                            \
            hipLaunchKernelGGL((bq_grid_query_kernel<NB_, DIL_>), dim3(gx, b), dim3(kQWaves * 64), 0, stream, n, m,     \
                               xyz1, xyz2, (const int *)workspace, B);                                                  \
    } while (0)
    switch (nbands * 2 + (dilated ? 1 : 0)) {
        case 2: SA_BQ_LAUNCH(1, false); break;
        case 3: SA_BQ_LAUNCH(1, true); break;
        case 4: SA_BQ_LAUNCH(2, false); break;
        case 5: SA_BQ_LAUNCH(2, true); break;
        case 6: SA_BQ_LAUNCH(3, false); break;
        case 7: SA_BQ_LAUNCH(3, true); break;
        case 8: SA_BQ_LAUNCH(4, false); break;
        default: SA_BQ_LAUNCH(4, true); break;
    }
#undef SA_BQ_LAUNCH
    SA_CHECK_LAUNCH();
    return SA_OK;
}

extern "C" int sa_query_ball_point_grid(int b, int n, int m, int nbands, const float *rmin, const float *rmax,
                                        const int *ns, int dilated, const float *xyz1, const float *xyz2,
                                        int *const *idx, int *const *cnt, void *workspace, hipStream_t stream) {
    return sa_query_ball_point_grid_ex(b, n, m, nbands, rmin, rmax, ns, dilated, xyz1, xyz2, idx, cnt, workspace, 0, stream);
}
