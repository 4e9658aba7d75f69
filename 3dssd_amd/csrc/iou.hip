// calc_iou / calc_iou_match of lib/utils/tf_ops/evaluation (tf_evaluate.cpp:142-217 -> evaluate.cpp:461-537,
// 1161-1227): bird's-eye-view and 3-D IoU of rotated boxes (t1, t2, t3, l, h, w, ry), used by the target assigner and
// the IoU loss (lib/builder/target_assigner.py:110, loss_builder.py:151).  The reference is a CPU op on
// boost::geometry polygons in double; here one thread per pair clips rectangle A against the four edges of rectangle B
// (Sutherland-Hodgman, at most 8 vertices) in double and takes the shoelace area -- the same quantities, not the same
// arithmetic: agreement is to rounding (~1e-6 relative), not bit for bit.
//   corners  (+-l/2, +-w/2) turned by [[cos ry, sin ry], [-sin ry, cos ry]] and moved to (t1, t3)        evaluate.cpp:461-485
//   iou_bev  inter / (area A + area B - inter)             (union polygon area of two overlapping convex polygons) :487-507
//   iou_3d   inter * max(0, min(t2) - max(t2 - h)) / (vol A + vol B - that)                                         :510-536
// Pairs whose union has no area (both boxes degenerate) give 0 (0/0 in the reference).
#include <math.h>

#include "sa_common.h"

namespace {

struct P2 {
    double x, y;
};

__device__ __forceinline__ void box_corners(const float *q, P2 (&c)[4]) {
    const double t1 = q[0], t3 = q[2], l = q[3], w = q[5], ry = q[6];
    const double cs = cos(ry), sn = sin(ry);
    const double hx = l / 2, hy = w / 2;
    const double dx[4] = {hx, hx, -hx, -hx}, dy[4] = {hy, -hy, -hy, hy};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c[i].x = cs * dx[i] + sn * dy[i] + t1;
        c[i].y = -sn * dx[i] + cs * dy[i] + t3;
    }
}

__device__ __forceinline__ double poly_area(const P2 *p, int n) {
    double a = 0.0;
    for (int i = 0; i < n; ++i) {
        const int j = i + 1 < n ? i + 1 : 0;
        a += p[i].x * p[j].y - p[j].x * p[i].y;
    }
    return fabs(a) * 0.5;
}

// area of A intersect B for two rectangles given by their corners in the same (clockwise) order
__device__ __forceinline__ double rect_intersection_area(const P2 (&A)[4], const P2 (&B)[4]) {
    P2 cur[10], nxt[10];
    int n = 4;
    for (int i = 0; i < 4; ++i) cur[i] = A[i];
    for (int e = 0; e < 4 && n > 0; ++e) {
        const P2 p1 = B[e], p2 = B[(e + 1) & 3];
        const double ex = p2.x - p1.x, ey = p2.y - p1.y;
        int k = 0;
        for (int i = 0; i < n; ++i) {
            const P2 s = cur[i], t = cur[i + 1 < n ? i + 1 : 0];
            const double ds = ex * (s.y - p1.y) - ey * (s.x - p1.x);       // <= 0: on the inner side of a clockwise edge
            const double dt = ex * (t.y - p1.y) - ey * (t.x - p1.x);
            if (ds <= 0.0) nxt[k++] = s;
            if ((ds < 0.0 && dt > 0.0) || (ds > 0.0 && dt < 0.0)) {
                const double u = ds / (ds - dt);
                nxt[k].x = s.x + u * (t.x - s.x);
                nxt[k].y = s.y + u * (t.y - s.y);
                ++k;
            }
        }
        n = k;
        for (int i = 0; i < n; ++i) cur[i] = nxt[i];
    }
    return n >= 3 ? poly_area(cur, n) : 0.0;
}

__device__ __forceinline__ void pair_iou(const float *d, const float *g, float *bev, float *iou3d) {
    P2 dc[4], gc[4];
    box_corners(d, dc);
    box_corners(g, gc);
    const double inter = rect_intersection_area(gc, dc);
    const double ad = poly_area(dc, 4), ag = poly_area(gc, 4);
    const double uni = ad + ag - inter;
    *bev = uni > 0.0 ? (float)(inter / uni) : 0.0f;
    const double dt2 = d[1], gt2 = g[1], dh = d[4], gh = g[4];
    const double ymax = fmin(dt2, gt2), ymin = fmax(dt2 - dh, gt2 - gh);
    const double ivol = inter * fmax(0.0, ymax - ymin);
    const double dvol = dh * (double)d[3] * (double)d[5], gvol = gh * (double)g[3] * (double)g[5];
    const double uvol = dvol + gvol - ivol;
    *iou3d = uvol > 0.0 ? (float)(ivol / uvol) : 0.0f;
}

__global__ __launch_bounds__(256) void calc_iou_kernel(long total, int det_num, int gt_num, int matching,
                                                       const float *__restrict__ dets, const float *__restrict__ gts,
                                                       float *__restrict__ iou_bev, float *__restrict__ iou_3d) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long di, gi;
        if (matching) {
            di = gi = e;
        } else {
            const long img = e / ((long)det_num * gt_num);
            const long r = e - img * det_num * gt_num;
            di = img * det_num + r / gt_num;
            gi = img * gt_num + r % gt_num;
        }
        pair_iou(dets + di * 7, gts + gi * 7, iou_bev + e, iou_3d + e);
    }
}

int launch_iou(long total, int det_num, int gt_num, int matching, const float *dets, const float *gts, float *iou_bev,
               float *iou_3d, hipStream_t stream) {
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(calc_iou_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, total, det_num, gt_num, matching, dets,
                       gts, iou_bev, iou_3d);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

}  // namespace

// calc_intersections_cpu(dets, gts, det_num, gt_num, num_images, IoU3D, IoUBeV) -- tf_evaluate.cpp:142: dets [bs,det_num,7],
// gts [bs,gt_num,7] -> iou_bev, iou_3d [bs,det_num,gt_num].  (Argument order here: bev first, like the op's outputs.)
extern "C" int sa_calc_iou(int bs, int det_num, int gt_num, const float *dets, const float *gts, float *iou_bev,
                           float *iou_3d, hipStream_t stream) {
    if (bs <= 0 || det_num <= 0 || gt_num <= 0 || !dets || !gts || !iou_bev || !iou_3d) return SA_ERR_INVALID;
    return launch_iou((long)bs * det_num * gt_num, det_num, gt_num, 0, dets, gts, iou_bev, iou_3d, stream);
}

// calc_intersections_matching_cpu(dets, gts, bs, IoU3D, IoUBeV) -- tf_evaluate.cpp:182: row i of dets against row i of gts.
extern "C" int sa_calc_iou_match(int n, const float *dets, const float *gts, float *iou_bev, float *iou_3d,
                                 hipStream_t stream) {
    if (n <= 0 || !dets || !gts || !iou_bev || !iou_3d) return SA_ERR_INVALID;
    return launch_iou(n, 1, 1, 1, dets, gts, iou_bev, iou_3d, stream);
}
