// points_pooling (lib/utils/tf_ops/points_pooling/tf_points_pooling_g.cu:36-118; second-stage pooling of
// lib/builder/points_pooler.py:81): the points gathered for a proposal are binned into an l x h x w grid over the
// proposal's box, each voxel keeps its FIRST sample_num points in point order (features copied, index recorded), plus
// the voxel counts and the voxel centres.  The reference runs one THREAD per proposal over all its points.
// Here a proposal belongs to one wave64: 64 points per step; a lane's slot in its voxel is the voxel's running count
// (in LDS) plus the number of earlier lanes of the step that fell into the same voxel, which reproduces the serial
// order exactly; counts saturate at sample_num like the reference's `continue`.
// Arithmetic as written in the reference: xmin = cx - l/2. and the centres xmin + (i + 0.5)*interval in double, rounded
// on the store; the voxel index from float subtract / divide / floor.  The reference offsets `pillars` by
// batch_inds*l*h*w floats instead of *3 (tf_points_pooling_g.cu:66), so its proposals overwrite each other's centres;
// the intended [bs, proposal_num, l, h, w, 3] layout is produced here.
#include <math.h>

#include "sa_common.h"

namespace {

constexpr int kPoolWaves = 4;
constexpr int kPoolMaxVox = 2048;     // voxels per proposal held in LDS (7 x 7 x 7 = 343 in the reference's use)

__global__ __launch_bounds__(64 * kPoolWaves) void points_pooling_kernel(
    int total, int point_num, int c, int l, int h, int w, int sample_num, const float *__restrict__ pc,
    const float *__restrict__ box_3d, const float *__restrict__ pc_loc, float *__restrict__ out_features,
    int *__restrict__ out_idx, int *__restrict__ sampled_num, float *__restrict__ pillars) {
    __shared__ int s_cnt[kPoolWaves][kPoolMaxVox];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = blockIdx.x * kPoolWaves + wv;
    if (q >= total) return;
    const int nvox = l * h * w;
    int *cnt = s_cnt[wv];
    for (int v = lane; v < nvox; v += 64) cnt[v] = 0;
    const float *bx = box_3d + (size_t)q * 6;
    const float cx = bx[0], by = bx[1], cz = bx[2], bl = bx[3], bh = bx[4], bw = bx[5];
    const float il = bl / (float)l, ih = bh / (float)h, iw = bw / (float)w;      // :57-59
    const float xmin = (float)((double)cx - (double)bl / 2.0);                    // :61-63
    const float ymin = by - bh;
    const float zmin = (float)((double)cz - (double)bw / 2.0);
    float *pl = pillars + (size_t)q * nvox * 3;
    for (int v = lane; v < nvox; v += 64) {
        const int i = v / (h * w), j = (v / w) % h, k = v % w;
        pl[v * 3 + 0] = (float)((double)xmin + ((double)i + 0.5) * (double)il);  // :74-76
        pl[v * 3 + 1] = (float)((double)ymin + ((double)j + 0.5) * (double)ih);
        pl[v * 3 + 2] = (float)((double)zmin + ((double)k + 0.5) * (double)iw);
    }
    const float *loc = pc_loc + (size_t)q * point_num * 3;
    const float *feat = pc + (size_t)q * point_num * c;
    float *of = out_features + (size_t)q * nvox * sample_num * c;
    int *oi = out_idx + (size_t)q * nvox * sample_num;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int p0 = 0; p0 < point_num; p0 += 64) {
        const int p = p0 + lane;
        int v = -1;
        if (p < point_num) {
            const int xi = min(max((int)floorf((loc[p * 3 + 0] - xmin) / il), 0), l - 1);   // :91-93
            const int yi = min(max((int)floorf((loc[p * 3 + 1] - ymin) / ih), 0), h - 1);
            const int zi = min(max((int)floorf((loc[p * 3 + 2] - zmin) / iw), 0), w - 1);
            v = xi * h * w + yi * w + zi;
        }
        int rank = 0, same = 0;
        for (int s = 0; s < 64; ++s) {
            const int vs = __builtin_amdgcn_readlane(v, s);
            const bool eqv = vs == v;
            same += eqv ? 1 : 0;
            rank += (eqv && s < lane) ? 1 : 0;
        }
        if (v >= 0) {
            const int base = cnt[v];
            const int slot = base + rank;
            if (slot < sample_num) {                                             // :96-97
                const size_t g = (size_t)v * sample_num + slot;
                oi[g] = p;
                for (int ch = 0; ch < c; ++ch) of[g * c + ch] = feat[(size_t)p * c + ch];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // every lane has read cnt[v]
            if (rank == 0) cnt[v] = min(sample_num, base + same);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    int *sn = sampled_num + (size_t)q * nvox;
    for (int v = lane; v < nvox; v += 64) sn[v] = cnt[v];
}

// pc_grad[q, out_idx[q, v, s], ch] += features_grad[q, v, s, ch] for s < sampled_num[q, v]   (:131-153)
__global__ __launch_bounds__(256) void points_pooling_grad_kernel(long total, int point_num, int c, int nvox,
                                                                  int sample_num, const int *__restrict__ out_idx,
                                                                  const int *__restrict__ sampled_num,
                                                                  const float *__restrict__ features_grad,
                                                                  float *pc_grad) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long g = e / c;                       // (proposal, voxel, slot)
        const int ch = (int)(e - g * c);
        const long qv = g / sample_num;             // (proposal, voxel)
        const int s = (int)(g - qv * sample_num);
        if (s >= sampled_num[qv]) continue;
        const long q = qv / nvox;
        atomicAdd(pc_grad + ((size_t)q * point_num + out_idx[g]) * c + ch, features_grad[e]);
    }
}

}  // namespace

// pointsPoolingLauncher(bs,proposal_num,point_num,channel_num,l,h,w,sample_num,pc,box_3d,pc_loc,out_features,out_idx,
// sampled_num_lists,pillars) -- tf_points_pooling.cpp:24; the four outputs are zeroed here (the op's cudaMemsets, :133-140)
extern "C" int sa_points_pooling(int bs, int proposal_num, int point_num, int channel_num, int l, int h, int w,
                                 int sample_num, const float *pc, const float *box_3d, const float *pc_loc,
                                 float *out_features, int *out_idx, int *sampled_num_lists, float *pillars,
                                 hipStream_t stream) {
    if (bs <= 0 || proposal_num <= 0 || point_num <= 0 || channel_num <= 0 || l <= 0 || h <= 0 || w <= 0 ||
        sample_num <= 0 || !pc || !box_3d || !pc_loc || !out_features || !out_idx || !sampled_num_lists || !pillars)
        return SA_ERR_INVALID;
    const long nvox = (long)l * h * w, total = (long)bs * proposal_num;
    if (nvox > kPoolMaxVox || total > 0x7FFFFFFF || (long)point_num * channel_num > 0x7FFFFFFF) return SA_ERR_UNSUPPORTED;
    const size_t slots = (size_t)total * nvox * sample_num;
    if (hipMemsetAsync(out_features, 0, slots * channel_num * sizeof(float), stream) != hipSuccess ||
        hipMemsetAsync(out_idx, 0, slots * sizeof(int), stream) != hipSuccess)
        return SA_ERR_LAUNCH;
    hipLaunchKernelGGL(points_pooling_kernel, dim3((unsigned)((total + kPoolWaves - 1) / kPoolWaves)),
                       dim3(64 * kPoolWaves), 0, stream, (int)total, point_num, channel_num, l, h, w, sample_num, pc, box_3d,
                       pc_loc, out_features, out_idx, sampled_num_lists, pillars);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// pointsPoolingGradLauncher(bs,proposal_num,point_num,channel_num,l,h,w,sample_num,pc,out_idx,sampled_num_lists,
// features_grad,pc_grad) -- tf_points_pooling.cpp:150 (pc only gives the shape); pc_grad is zeroed here (:219)
extern "C" int sa_points_pooling_grad(int bs, int proposal_num, int point_num, int channel_num, int l, int h, int w,
                                      int sample_num, const int *out_idx, const int *sampled_num_lists,
                                      const float *features_grad, float *pc_grad, hipStream_t stream) {
    if (bs <= 0 || proposal_num <= 0 || point_num <= 0 || channel_num <= 0 || l <= 0 || h <= 0 || w <= 0 ||
        sample_num <= 0 || !out_idx || !sampled_num_lists || !features_grad || !pc_grad)
        return SA_ERR_INVALID;
    const long nvox = (long)l * h * w;
    if (hipMemsetAsync(pc_grad, 0, (size_t)bs * proposal_num * point_num * channel_num * sizeof(float), stream) != hipSuccess)
        return SA_ERR_LAUNCH;
    const long total = (long)bs * proposal_num * nvox * sample_num * channel_num;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(points_pooling_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, total, point_num, channel_num,
                       (int)nvox, sample_num, out_idx, sampled_num_lists, features_grad, pc_grad);
    SA_CHECK_LAUNCH();
    return SA_OK;
}
