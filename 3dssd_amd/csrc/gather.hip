// Row gathers for gfx950: gather_point (tf_sampling_g.cu:320-331) and group_point
// (tf_grouping_g.cu:362-379).  Both are pure HBM traffic: the output is written once, fully
// coalesced (16 B per lane when the channel count allows), and the source rows are read in
// row-sized contiguous pieces.  The reference uses one thread per scalar on a fixed 512x64 grid.
#include "sa_common.h"

namespace {

// out[r, :] = src[batch(r), idx[r], :]   with r over b*rows_per_batch rows of c floats.
// NEG1_ZERO: idx == -1 produces a zero row (group_point, tf_grouping_g.cu:373-375).
template <typename VT, bool NEG1_ZERO>
__global__ __launch_bounds__(256) void gather_rows_kernel(long total_vec, int vec_per_row, int n,
                                                          long rows_per_batch,
                                                          const VT *__restrict__ src,
                                                          const int *__restrict__ idx,
                                                          VT *__restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total_vec;
         i += (long)gridDim.x * blockDim.x) {
        const long r = i / vec_per_row;
        const int v = (int)(i - r * vec_per_row);
        const long bi = r / rows_per_batch;
        const int a = idx[r];
        VT val;
        if (NEG1_ZERO && a == -1) {
            val = VT{};
        } else {
            val = src[((size_t)bi * n + a) * vec_per_row + v];
        }
        out[i] = val;
    }
}

// ---- row gathers at HBM speed (round 4; VERDICT r3 item 5: group_point ran at 2.2 TB/s overall) ---------------
// What the grid-stride kernel above spends its time on is not memory: two 64-bit divisions per 16-byte element, the
// index re-read by every lane of a row, and one load in flight per lane.  Here a wave owns 64 consecutive OUTPUT rows
// at a time: one coalesced index load (lane j <-> row j), the row's index handed to its lanes by a cross-lane read, no
// division at all when a 64-row block lies inside one frame, four independent 16-byte row pieces in flight per lane,
// and streaming (non-temporal) stores -- the grouped tensor is written once and not read again by this kernel, it
// should not evict the source frame (1 MB, re-read ~32 times) from the L2.  Waves of one XCD take contiguous ranges of
// the row list (sa::xcd_block), so a frame's source rows are fetched into one or two L2s instead of all eight.
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef int vi4 __attribute__((ext_vector_type(4)));

// c = 4 << LOG2V floats per row (LOG2V = 0..6: c = 4..256), 16-byte aligned tensors.
template <int LOG2V, bool NEG1_ZERO>
__global__ __launch_bounds__(256) void gather_rows64_kernel(long rows, int n, long rows_per_batch, int nblk_wg,
                                                            const vf4 *__restrict__ src, const int *__restrict__ idx,
                                                            vf4 *__restrict__ out) {
    constexpr int VPR = 1 << LOG2V;            // lanes per row
    constexpr int RPI = 64 >> LOG2V;           // rows per wave per step
    constexpr int UN = VPR >= 4 ? 4 : VPR;     // steps in flight
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int stride;
    const int pos = sa::xcd_block(blockIdx.x, gridDim.x, nblk_wg, stride);
    if (pos < 0) return;
    const long nblk = (rows + 63) >> 6;        // 64-row blocks
    for (long blk = (long)pos * 4 + wv; blk < nblk; blk += (long)stride * 4) {
        const long r0 = blk << 6;
        const long rj = r0 + lane;
        const int a = rj < rows ? idx[rj] : 0;
        // frame of the block's first and last row: equal -> uniform, no per-row division
        const long last = (r0 + 63 < rows ? r0 + 63 : rows - 1);
        const long f0 = r0 / rows_per_batch, f1 = last / rows_per_batch;
        const int sub = lane >> LOG2V, v = lane & (VPR - 1);
#pragma unroll
        for (int it0 = 0; it0 < VPR; it0 += UN) {
            vf4 val[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int jr = (it0 + u) * RPI + sub;                       // row of the block this lane serves
                const int ar = __builtin_amdgcn_ds_bpermute(jr << 2, a);    // its index (from lane jr)
                const long r = r0 + jr;
                const long f = f0 == f1 ? f0 : r / rows_per_batch;
                const bool live = r < rows && !(NEG1_ZERO && ar == -1);
                val[u] = live ? src[(((size_t)f * n + (live ? ar : 0)) << LOG2V) + v] : vf4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const long r = r0 + (it0 + u) * RPI + sub;
                if (r < rows) __builtin_nontemporal_store(val[u], out + ((size_t)r << LOG2V) + v);
            }
        }
    }
}

// c == 3 (grouped xyz): a lane packs FOUR rows (48 B = three 16-byte stores; a wave writes 3 KB contiguous).
// rows % 4 == 0, rows_per_batch % 4 == 0, out and idx 16-byte aligned.
template <bool NEG1_ZERO>
__global__ __launch_bounds__(256) void gather_rows3x4_kernel(long quads, int n, long quads_per_batch, int nblk_wg,
                                                             const float *__restrict__ src, const vi4 *__restrict__ idx4,
                                                             vf4 *__restrict__ out) {
    int stride;
    const int pos = sa::xcd_block(blockIdx.x, gridDim.x, nblk_wg, stride);
    if (pos < 0) return;
    for (long q = (long)pos * 256 + threadIdx.x; q < quads; q += (long)stride * 256) {
        const vi4 a = idx4[q];
        const float *fr = src + (size_t)(q / quads_per_batch) * n * 3;
        float p[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool live = !(NEG1_ZERO && a[k] == -1);
            const float *row = fr + (size_t)(live ? a[k] : 0) * 3;
            p[3 * k + 0] = live ? row[0] : 0.f; p[3 * k + 1] = live ? row[1] : 0.f; p[3 * k + 2] = live ? row[2] : 0.f;
        }
        vf4 *o = out + q * 3;
        __builtin_nontemporal_store(vf4{p[0], p[1], p[2], p[3]}, o);
        __builtin_nontemporal_store(vf4{p[4], p[5], p[6], p[7]}, o + 1);
        __builtin_nontemporal_store(vf4{p[8], p[9], p[10], p[11]}, o + 2);
    }
}

// c == 1 (the intensity channel of layer 1): four rows per lane, one 16-byte store
template <bool NEG1_ZERO>
__global__ __launch_bounds__(256) void gather_rows1x4_kernel(long quads, int n, long quads_per_batch, int nblk_wg,
                                                             const float *__restrict__ src, const vi4 *__restrict__ idx4,
                                                             vf4 *__restrict__ out) {
    int stride;
    const int pos = sa::xcd_block(blockIdx.x, gridDim.x, nblk_wg, stride);
    if (pos < 0) return;
    for (long q = (long)pos * 256 + threadIdx.x; q < quads; q += (long)stride * 256) {
        const vi4 a = idx4[q];
        const float *fr = src + (size_t)(q / quads_per_batch) * n;
        vf4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (NEG1_ZERO && a[k] == -1) ? 0.f : fr[a[k] == -1 ? 0 : a[k]];
        __builtin_nontemporal_store(v, out + q);
    }
}

template <bool NEG1_ZERO>
int launch_gather(int b, int n, int c, long rows_per_batch, const float *src, const int *idx, float *out,
                  hipStream_t stream) {
    const long rows = (long)b * rows_per_batch;
    const bool al16 = (((uintptr_t)src | (uintptr_t)out) % 16 == 0);
    // ---- 64-row-block kernels: c = 4, 8, ..., 256 (a power of two)
    if (al16 && c >= 4 && c <= 256 && (c & (c - 1)) == 0) {
        const long nblk = (rows + 63) >> 6;
        long wgs = (nblk + 3) / 4;
        const int need = (int)(wgs > 4096 ? 4096 : wgs);
        const unsigned grid = (unsigned)((need + 7) & ~7);
#define SA_G64(L)                                                                                                    \
    hipLaunchKernelGGL((gather_rows64_kernel<L, NEG1_ZERO>), dim3(grid), dim3(256), 0, stream, rows, n, rows_per_batch, \
                       need, (const vf4 *)src, idx, (vf4 *)out)
        switch (c) {
            case 4: SA_G64(0); break;
            case 8: SA_G64(1); break;
            case 16: SA_G64(2); break;
            case 32: SA_G64(3); break;
            case 64: SA_G64(4); break;
            case 128: SA_G64(5); break;
            default: SA_G64(6); break;
        }
#undef SA_G64
        SA_CHECK_LAUNCH();
        return SA_OK;
    }
    if (c == 3 && rows_per_batch % 4 == 0 && (((uintptr_t)idx | (uintptr_t)out) % 16 == 0)) {
        const long quads = rows / 4;
        long wgs = (quads + 255) / 256;
        const int need = (int)(wgs > 4096 ? 4096 : wgs);
        const unsigned grid = (unsigned)((need + 7) & ~7);
        hipLaunchKernelGGL((gather_rows3x4_kernel<NEG1_ZERO>), dim3(grid), dim3(256), 0, stream, quads, n,
                           rows_per_batch / 4, need, src, (const vi4 *)idx, (vf4 *)out);
        SA_CHECK_LAUNCH();
        return SA_OK;
    }
    if (c == 1 && rows_per_batch % 4 == 0 && (((uintptr_t)idx | (uintptr_t)out) % 16 == 0)) {
        const long quads = rows / 4;
        long wgs = (quads + 255) / 256;
        const int need = (int)(wgs > 4096 ? 4096 : wgs);
        const unsigned grid = (unsigned)((need + 7) & ~7);
        hipLaunchKernelGGL((gather_rows1x4_kernel<NEG1_ZERO>), dim3(grid), dim3(256), 0, stream, quads, n,
                           rows_per_batch / 4, need, src, (const vi4 *)idx, (vf4 *)out);
        SA_CHECK_LAUNCH();
        return SA_OK;
    }
    const bool v4 = (c % 4 == 0) && al16;
    const int vpr = v4 ? c / 4 : c;
    const long total = rows * vpr;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (v4)
        hipLaunchKernelGGL((gather_rows_kernel<float4, NEG1_ZERO>), dim3((unsigned)blocks), dim3(256), 0,
                           stream, total, vpr, n, rows_per_batch, (const float4 *)src, idx, (float4 *)out);
    else
        hipLaunchKernelGGL((gather_rows_kernel<float, NEG1_ZERO>), dim3((unsigned)blocks), dim3(256), 0,
                           stream, total, vpr, n, rows_per_batch, src, idx, out);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

}  // namespace

// ---- up to four strided block copies in ONE launch (tf.slice of layers_util.py:85-86 and
//      single_stage_detector.py:117-118: the [B,n,3+C] input into xyz / features, a prefix range of xyz and points)
namespace {
constexpr int kMaxCopyJobs = 4;
struct CopyJob {
    const float *src;
    float *dst;
    int frames, rows, cols;              // dst[f, r, 0:cols] = src[f, r, 0:cols]
    long src_fs, src_rs, dst_fs, dst_rs; // frame / row strides in floats
};
struct CopyJobs { CopyJob j[kMaxCopyJobs]; };
__global__ __launch_bounds__(256) void copy_blocks_kernel(CopyJobs J) {
    const CopyJob job = J.j[blockIdx.y];
    const long per_frame = (long)job.rows * job.cols, total = per_frame * job.frames;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long f = i / per_frame, q = i - f * per_frame;
        const long r = q / job.cols;
        const int c = (int)(q - r * job.cols);
        job.dst[f * job.dst_fs + r * job.dst_rs + c] = job.src[f * job.src_fs + r * job.src_rs + c];
    }
}
// ---- the executor's package fill: up to 32 dense batches (equal size, 16-byte multiples) into consecutive parts of one
//      buffer in ONE launch (3dssd_amd/pipeline.py: a submit only notes the source; the copies of a package used to be 16
//      launches in front of its sampling stage)
constexpr int kMaxBatches = 32;
struct BatchSrcs { const float4 *src[kMaxBatches]; };
__global__ __launch_bounds__(256) void copy_batches_kernel(BatchSrcs S, float4 *__restrict__ dst, long vec_per_batch) {
    const float4 *__restrict__ s = S.src[blockIdx.y];
    float4 *__restrict__ d = dst + (size_t)blockIdx.y * vec_per_batch;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < vec_per_batch; i += (long)gridDim.x * blockDim.x) d[i] = s[i];
}
}  // namespace

// jobs: host array of njobs (<= 4) records of 9 longs {src, dst, frames, rows, cols, src_frame_stride,
// src_row_stride, dst_frame_stride, dst_row_stride} (pointers as integers, strides in floats).
extern "C" int sa_copy_blocks(int njobs, const long *jobs, hipStream_t stream) {
    if (njobs < 1 || njobs > kMaxCopyJobs || !jobs) return SA_ERR_INVALID;
    CopyJobs J{};
    long most = 0;
    for (int i = 0; i < njobs; ++i) {
        const long *q = jobs + 9 * i;
        if (!q[0] || !q[1] || q[2] <= 0 || q[3] <= 0 || q[4] <= 0) return SA_ERR_INVALID;
        J.j[i].src = (const float *)q[0]; J.j[i].dst = (float *)q[1];
        J.j[i].frames = (int)q[2]; J.j[i].rows = (int)q[3]; J.j[i].cols = (int)q[4];
        J.j[i].src_fs = q[5]; J.j[i].src_rs = q[6]; J.j[i].dst_fs = q[7]; J.j[i].dst_rs = q[8];
        const long t = q[2] * q[3] * q[4];
        if (t > most) most = t;
    }
    long blocks = (most + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy_blocks_kernel, dim3((unsigned)blocks, njobs), dim3(256), 0, stream, J);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// n dense batches of `bytes_per_batch` bytes each (a multiple of 16; srcs[i] and dst 16-byte aligned) -> dst, back to back.
extern "C" int sa_copy_batches(int n, const void *const *srcs, void *dst, long bytes_per_batch, hipStream_t stream) {
    if (n < 1 || n > kMaxBatches || !srcs || !dst || bytes_per_batch <= 0 || (bytes_per_batch & 15) || ((uintptr_t)dst & 15))
        return SA_ERR_INVALID;
    BatchSrcs S{};
    for (int i = 0; i < n; ++i) {
        if (!srcs[i] || ((uintptr_t)srcs[i] & 15)) return SA_ERR_INVALID;
        S.src[i] = (const float4 *)srcs[i];
    }
    const long vec = bytes_per_batch / 16;
    long blocks = (vec + 256 * 8 - 1) / (256 * 8);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(copy_batches_kernel, dim3((unsigned)blocks, n), dim3(256), 0, stream, S, (float4 *)dst, vec);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// lib/utils/tf_ops/sampling/tf_sampling.cpp:235  gatherpointLauncher(b,n,m,c,inp,idx,out)
extern "C" int sa_gather_point(int b, int n, int m, int c, const float *inp, const int *idx, float *out,
                               hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || c <= 0 || !inp || !idx || !out) return SA_ERR_INVALID;
    return launch_gather<false>(b, n, c, m, inp, idx, out, stream);
}

// lib/utils/tf_ops/grouping/tf_grouping.cpp:446  groupPointLauncher(b,n,c,m,nsample,points,idx,out)
extern "C" int sa_group_point(int b, int n, int c, int m, int nsample, const float *points,
                              const int *idx, float *out, hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || c <= 0 || nsample <= 0 || !points || !idx || !out)
        return SA_ERR_INVALID;
    return launch_gather<true>(b, n, c, (long)m * nsample, points, idx, out, stream);
}

// ---- the gradients of the two gathers and gather_by_mask ---------------------------------
namespace {

// dst[batch(r), idx[r], :] += g[r, :]  (idx == -1 rows skipped when NEG1_SKIP).  Float atomics like the reference
// (tf_sampling_g.cu:339-351, tf_grouping_g.cu:384-400): the summation order of rows that hit the same point is not
// defined there either.
template <bool NEG1_SKIP>
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(long total, int c, int n, long rows_per_batch,
                                                               const float *__restrict__ g,
                                                               const int *__restrict__ idx, float *dst) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c;
        const int ch = (int)(i - r * c);
        const long bi = r / rows_per_batch;
        const int a = idx[r];
        if (NEG1_SKIP && a == -1) continue;
        atomicAdd(dst + ((size_t)bi * n + a) * c + ch, g[i]);
    }
}

template <bool NEG1_SKIP>
int launch_scatter(int b, int n, int c, long rows_per_batch, const float *g, const int *idx, float *dst,
                   hipStream_t stream) {
    if (sa::zero_async(dst, (size_t)b * n * c * sizeof(float), stream) != hipSuccess) return SA_ERR_LAUNCH;
    const long total = (long)b * rows_per_batch * c;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scatter_add_rows_kernel<NEG1_SKIP>, dim3((unsigned)blocks), dim3(256), 0, stream, total, c, n,
                       rows_per_batch, g, idx, dst);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

constexpr int kMaskBlock = 256;

// One workgroup per frame: the first proposal_num points with int(mask) != 0, in point order; rows past the number
// found repeat the first one (tf_sampling_g.cu:356-384).  No selected point at all: zero rows (the reference leaves
// the output uninitialised).
__global__ __launch_bounds__(kMaskBlock) void gather_by_mask_kernel(int n, int c, int proposal_num,
                                                                    const float *__restrict__ inp,
                                                                    const float *__restrict__ mask,
                                                                    float *__restrict__ out, int *__restrict__ sel) {
    __shared__ int s_wave[kMaskBlock / 64];
    __shared__ int s_total;
    const int bi = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float *mk = mask + (size_t)bi * n;
    int *sl = sel + (size_t)bi * proposal_num;
    if (t == 0) s_total = 0;
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += kMaskBlock) {
        const int base = s_total;
        if (base >= proposal_num) break;                         // uniform: s_total is read after a barrier
        const int k = k0 + t;
        const bool on = k < n && (int)mk[k] != 0;                // tf_sampling_g.cu:366
        const unsigned long long hit = __ballot(on);
        if (lane == 0) s_wave[w] = __builtin_popcountll(hit);
        __syncthreads();
        int before = base, all = 0;
        for (int i = 0; i < kMaskBlock / 64; ++i) {
            before += i < w ? s_wave[i] : 0;
            all += s_wave[i];
        }
        const int pos = before + __builtin_popcountll(hit & ((1ull << lane) - 1ull));
        if (on && pos < proposal_num) sl[pos] = k;
        __syncthreads();
        if (t == 0) s_total = base + all;
        __syncthreads();
    }
    const int cnt = s_total < proposal_num ? s_total : proposal_num;
    __threadfence_block();
    const float *src = inp + (size_t)bi * n * c;
    float *dst = out + (size_t)bi * proposal_num * c;
    const long total = (long)proposal_num * c;
    for (long i = t; i < total; i += kMaskBlock) {
        const int r = (int)(i / c);
        const int ch = (int)(i - (long)r * c);
        dst[i] = cnt == 0 ? 0.0f : src[(size_t)sl[r < cnt ? r : 0] * c + ch];
    }
}

}  // namespace

// scatteraddpointLauncher(b,n,m,c,out_g,idx,inp_g) -- tf_sampling.cpp:261; inp_g is zeroed here (the op's cudaMemset,
// tf_sampling.cpp:285)
extern "C" int sa_gather_point_grad(int b, int n, int m, int c, const float *out_g, const int *idx, float *inp_g,
                                    hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || c <= 0 || !out_g || !idx || !inp_g) return SA_ERR_INVALID;
    return launch_scatter<false>(b, n, c, m, out_g, idx, inp_g, stream);
}

// groupPointGradLauncher(b,n,c,m,nsample,grad_out,idx,grad_points) -- tf_grouping.cpp:479 (+ the memset of :509)
extern "C" int sa_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                                   float *grad_points, hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || c <= 0 || nsample <= 0 || !grad_out || !idx || !grad_points)
        return SA_ERR_INVALID;
    return launch_scatter<true>(b, n, c, (long)m * nsample, grad_out, idx, grad_points, stream);
}

// GatherByMaskLauncher(b,n,c,proposal_num,inp,mask,out) -- tf_sampling.cpp:293.  `sel` [b,proposal_num] int scratch
// (receives the selected point indices).
extern "C" int sa_gather_by_mask(int b, int n, int c, int proposal_num, const float *inp, const float *mask,
                                 float *out, int *sel, hipStream_t stream) {
    if (b <= 0 || n <= 0 || c <= 0 || proposal_num <= 0 || !inp || !mask || !out || !sel) return SA_ERR_INVALID;
    hipLaunchKernelGGL(gather_by_mask_kernel, dim3(b), dim3(kMaskBlock), 0, stream, n, c, proposal_num, inp, mask, out,
                       sel);
    SA_CHECK_LAUNCH();
    return SA_OK;
}
