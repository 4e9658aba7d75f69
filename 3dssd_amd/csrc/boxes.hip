// Point-in-box operators of lib/utils/tf_ops/grouping (used by the second stage and the
// target assigner: lib/builder/points_pooler.py:34,123, lib/builder/target_assigner.py:116):
//   query_boxes_3d_points  first nsample points inside each box            tf_grouping_g.cu:44-95
//   query_boxes_3d_mask    inside-flag of every (box, point) pair          tf_grouping_g.cu:98-134
//   query_points_iou       |P in A and G| / |P in A or G| per (anchor, gt) tf_grouping_g.cu:137-209
// The reference runs one THREAD per box / pair over all n points on a fixed 512 x 64 grid.  Here a box (or a pair)
// belongs to one wave64: 64 points per step, one ballot, the ordered prefix of the hits written by the lanes that
// own them; the mask kernel is a plain coalesced elementwise pass (one int per (box, point), HBM-write bound).
//
// Box test = point_inside_box_3d (tf_grouping_g.cu:27-41), restated with the arithmetic spelled out:
//   md   = max(sqrtf((float)((l/2.)*(l/2.) + (w/2.)*(w/2.))), 1e-20f)      (the halves and their squares in double)
//   out  if |x-cx| > md  or  y > by  or  (by - y) > h  or  |z-cz| > md
//   cosr = cos(ry), sinr = sin(ry): correctly rounded float values (computed in double here and in the oracle)
//   u = (x-cx)*cosr - (z-cz)*sinr ; v = (x-cx)*sinr + (z-cz)*cosr contracted like the ball-query distance (oracle
//       decision B: the LEFT product of an add/sub is the fused one): u = fma(dx, cosr, -(dz*sinr)),
//       v = fma(dx, sinr, dz*cosr)
//   in   iff -l/2 <= u <= l/2 and -w/2 <= v <= w/2                            (l/2, w/2 exact in float)
#include <math.h>

#include "sa_common.h"

namespace {

struct Box {
    float cx, by, cz, h, hl, hw, md, cosr, sinr;
};

__device__ __forceinline__ Box load_box(const float *__restrict__ q) {
    Box bx;
    bx.cx = q[0]; bx.by = q[1]; bx.cz = q[2];
    const float l = q[3], w = q[5], ry = q[6];
    bx.h = q[4];
    const double hl = (double)l / 2.0, hw = (double)w / 2.0;
    bx.md = fmaxf(sqrtf((float)(hl * hl + hw * hw)), 1e-20f);
    bx.hl = l * 0.5f;
    bx.hw = w * 0.5f;
    bx.cosr = (float)cos((double)ry);
    bx.sinr = (float)sin((double)ry);
    return bx;
}

__device__ __forceinline__ bool inside_box(const Box &bx, float x, float y, float z) {
    const float dx = x - bx.cx, dz = z - bx.cz;
    if (fabsf(dx) > bx.md || y > bx.by || (bx.by - y) > bx.h || fabsf(dz) > bx.md) return false;
    const float u = __builtin_fmaf(dx, bx.cosr, -(dz * bx.sinr));
    const float v = __builtin_fmaf(dx, bx.sinr, dz * bx.cosr);
    return u >= -bx.hl && u <= bx.hl && v >= -bx.hw && v <= bx.hw;
}

constexpr int kWavesPerBlock = 4;

// one wave per box: idx [b,m,nsample], cnt [b,m]
__global__ __launch_bounds__(64 * kWavesPerBlock) void boxes_points_kernel(int n, int m, int total, int nsample,
                                                                          const float *__restrict__ xyz,
                                                                          const float *__restrict__ boxes,
                                                                          int *__restrict__ idx,
                                                                          int *__restrict__ pts_cnt) {
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (q >= total) return;
    const int bi = q / m;
    const float *p = xyz + (size_t)bi * n * 3;
    const Box bx = load_box(boxes + (size_t)q * 7);
    int *o = idx + (size_t)q * nsample;
    int cnt = 0, first = 0;
    for (int k0 = 0; k0 < n && cnt < nsample; k0 += 64) {
        const int k = k0 + lane;
        bool in = false;
        if (k < n) in = inside_box(bx, p[k * 3 + 0], p[k * 3 + 1], p[k * 3 + 2]);
        const unsigned long long hit = __ballot(in);
        if (hit) {
            if (cnt == 0) first = k0 + __builtin_ctzll(hit);
            const int pos = cnt + __builtin_popcountll(hit & ((1ull << lane) - 1ull));
            if (in && pos < nsample) o[pos] = k;
            cnt += __builtin_popcountll(hit);
        }
    }
    cnt = cnt < nsample ? cnt : nsample;
    // rows shorter than nsample repeat the first hit (tf_grouping_g.cu:84-87); an empty box gives zeros (the
    // reference leaves the row uninitialised)
    for (int s = cnt + lane; s < nsample; s += 64) o[s] = first;
    if (lane == 0) pts_cnt[q] = cnt;
}

// mask [b,m,n]: grid (ceil(n/256), b*m)
__global__ __launch_bounds__(256) void boxes_mask_kernel(int n, int m, const float *__restrict__ xyz,
                                                         const float *__restrict__ boxes, int *__restrict__ mask) {
    const int q = blockIdx.y;
    const int bi = q / m;
    const Box bx = load_box(boxes + (size_t)q * 7);
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const float *p = xyz + ((size_t)bi * n + k) * 3;
    mask[(size_t)q * n + k] = inside_box(bx, p[0], p[1], p[2]) ? 1 : 0;
}

// one wave per (anchor, gt) pair
__global__ __launch_bounds__(64 * kWavesPerBlock) void points_iou_kernel(int n, int anchors_num, int gt_num,
                                                                        long total, const float *__restrict__ xyz,
                                                                        const float *__restrict__ anchors,
                                                                        const float *__restrict__ gt,
                                                                        const float *__restrict__ iou_matrix,
                                                                        float *__restrict__ iou_points) {
    const int lane = threadIdx.x & 63;
    const long q = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (q >= total) return;
    if (iou_matrix[q] < 1e-3f) {                       // tf_grouping_g.cu:146-150
        if (lane == 0) iou_points[q] = 0.0f;
        return;
    }
    const long bi = q / ((long)anchors_num * gt_num);
    const long ai = q / gt_num;                         // anchor row over [b, anchors_num]
    const int gi = (int)(q % gt_num);
    const float *p = xyz + (size_t)bi * n * 3;
    const Box ba = load_box(anchors + (size_t)ai * 7);
    const Box bg = load_box(gt + ((size_t)bi * gt_num + gi) * 7);
    int in = 0, un = 0;
    for (int k = lane; k < n; k += 64) {
        const float x = p[k * 3 + 0], y = p[k * 3 + 1], z = p[k * 3 + 2];
        const bool a = inside_box(ba, x, y, z), g = inside_box(bg, x, y, z);
        in += (a && g) ? 1 : 0;
        un += (a || g) ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) {
        in += __shfl_xor(in, off);
        un += __shfl_xor(un, off);
    }
    if (lane == 0) iou_points[q] = (float)in / (float)(un > 1 ? un : 1);
}

}  // namespace

// queryBoxes3dPointsLauncher(b,n,m,nsample,xyz,proposals,idx,pts_cnt) -- tf_grouping.cpp:228
extern "C" int sa_query_boxes_3d_points(int b, int n, int m, int nsample, const float *xyz, const float *proposals,
                                        int *idx, int *pts_cnt, hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || nsample <= 0 || !xyz || !proposals || !idx || !pts_cnt) return SA_ERR_INVALID;
    const long total = (long)b * m;
    if (total > 0x7FFFFFFF || (long)n * 3 > 0x7FFFFFFF) return SA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(boxes_points_kernel, dim3((unsigned)((total + kWavesPerBlock - 1) / kWavesPerBlock)),
                       dim3(64 * kWavesPerBlock), 0, stream, n, m, (int)total, nsample, xyz, proposals, idx, pts_cnt);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// queryBoxes3dMaskLauncher(b,n,m,xyz,boxes_3d,mask) -- tf_grouping.cpp:151
extern "C" int sa_query_boxes_3d_mask(int b, int n, int m, const float *xyz, const float *boxes_3d, int *mask,
                                      hipStream_t stream) {
    if (b <= 0 || n <= 0 || m <= 0 || !xyz || !boxes_3d || !mask) return SA_ERR_INVALID;
    if ((long)b * m > 65535) return SA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(boxes_mask_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)(b * m)), dim3(256), 0, stream, n,
                       m, xyz, boxes_3d, mask);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// queryPointsIouLauncher(b,n,anchors_num,gt_num,xyz,anchors_3d,gt_boxes_3d,iou_matrix,iou_points) -- tf_grouping.cpp:182
extern "C" int sa_query_points_iou(int b, int n, int anchors_num, int gt_num, const float *xyz, const float *anchors_3d,
                                   const float *gt_boxes_3d, const float *iou_matrix, float *iou_points,
                                   hipStream_t stream) {
    if (b <= 0 || n <= 0 || anchors_num <= 0 || gt_num <= 0 || !xyz || !anchors_3d || !gt_boxes_3d || !iou_matrix ||
        !iou_points)
        return SA_ERR_INVALID;
    const long total = (long)b * anchors_num * gt_num;
    const long blocks = (total + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > 0x7FFFFFFF || (long)n * 3 > 0x7FFFFFFF) return SA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(points_iou_kernel, dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), 0, stream, n, anchors_num,
                       gt_num, total, xyz, anchors_3d, gt_boxes_3d, iou_matrix, iou_points);
    SA_CHECK_LAUNCH();
    return SA_OK;
}
