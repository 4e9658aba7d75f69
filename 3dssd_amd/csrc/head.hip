// The step after the SA backbone: anchor-free box decoding + sigmoid scores
// + BEV boxes, and per-class BEV non-maximum suppression.
//
// Reference: lib/utils/anchor_decoder.py:6-14,86-112 (decode_class2angle, decode_dist_anchor_free),
// lib/modeling/single_stage_detector.py:208-212 (sigmoid), lib/utils/box_3d_utils.py:25-58 +
// lib/utils/anchors_util.py:11-50 (BEV box), lib/builder/postprocessor.py:76-88 (tf.image.non_max_suppression
// per class, max_output_size / iou_threshold from 3dssd.yaml:70-71).  In the reference these are TensorFlow graph
// ops; the arithmetic pinned by oracle/head_oracle.py is reproduced operation by operation (fp32, no contraction;
// |cos|, |sin| and the sigmoid through float64).
#include <math.h>

#include "sa_common.h"

namespace {

__global__ __launch_bounds__(256) void decode_anchor_free_kernel(long total, int A, int C, float interval,
                                                                 const float *__restrict__ xyz,
                                                                 const float *__restrict__ reg,
                                                                 const float *__restrict__ cls,
                                                                 float *__restrict__ boxes,
                                                                 float *__restrict__ scores,
                                                                 float *__restrict__ bev) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float *r = reg + i * (6 + 2 * A);
        int bin = 0;
        float best = r[6];
        for (int a = 1; a < A; ++a) {                 // tf.argmax: first maximum
            const float v = r[6 + a];
            if (v > best) { best = v; bin = a; }
        }
        const float ry = ((float)bin + r[6 + A + bin]) * interval;   // anchor_decoder.py:6-14
        const float cx = xyz[i * 3 + 0] + r[0];
        const float cy = (xyz[i * 3 + 1] + r[1]) + r[4];             // + (0, half_y, 0), :104-107
        const float cz = xyz[i * 3 + 2] + r[2];
        const float l = sa::fmax_nn(r[3] * 2.0f, 0.1f);
        const float h = sa::fmax_nn(r[4] * 2.0f, 0.1f);
        const float w = sa::fmax_nn(r[5] * 2.0f, 0.1f);
        float *o = boxes + i * 7;
        o[0] = cx; o[1] = cy; o[2] = cz; o[3] = l; o[4] = h; o[5] = w; o[6] = ry;
        for (int c = 0; c < C; ++c)
            scores[i * C + c] = (float)(1.0 / (1.0 + exp(-(double)cls[i * C + c])));
        const float cr = (float)fabs(cos((double)ry)), sr = (float)fabs(sin((double)ry));
        const float dimx = l * cr + w * sr;           // box_3d_utils.py:51-53 (no contraction: -ffp-contract=off)
        const float dimz = w * cr + l * sr;
        const float hx = dimx / 2.0f, hz = dimz / 2.0f;
        float *bv = bev + i * 4;
        bv[0] = cx - hx; bv[1] = cz - hz; bv[2] = cx + hx; bv[3] = cz + hz;   // anchors_util.py:37-47
    }
}

__global__ void boxes_to_bev_kernel(long total, const float *__restrict__ boxes, float *__restrict__ bev) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float *o = boxes + i * 7;
        const float cr = (float)fabs(cos((double)o[6])), sr = (float)fabs(sin((double)o[6]));
        const float dimx = o[3] * cr + o[5] * sr;
        const float dimz = o[5] * cr + o[3] * sr;
        const float hx = dimx / 2.0f, hz = dimz / 2.0f;
        float *bv = bev + i * 4;
        bv[0] = o[0] - hx; bv[1] = o[2] - hz; bv[2] = o[0] + hx; bv[3] = o[2] + hz;
    }
}

__device__ __forceinline__ float bev_iou(float4 a, float4 b) {
    const float ymin_i = sa::fmin_nn(a.x, a.z), xmin_i = sa::fmin_nn(a.y, a.w);
    const float ymax_i = sa::fmax_nn(a.x, a.z), xmax_i = sa::fmax_nn(a.y, a.w);
    const float ymin_j = sa::fmin_nn(b.x, b.z), xmin_j = sa::fmin_nn(b.y, b.w);
    const float ymax_j = sa::fmax_nn(b.x, b.z), xmax_j = sa::fmax_nn(b.y, b.w);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.0f || area_j <= 0.0f) return 0.0f;
    const float iy = sa::fmax_nn(sa::fmin_nn(ymax_i, ymax_j) - sa::fmax_nn(ymin_i, ymin_j), 0.0f);
    const float ix = sa::fmax_nn(sa::fmin_nn(xmax_i, xmax_j) - sa::fmax_nn(xmin_i, xmin_j), 0.0f);
    const float inter = iy * ix;
    return inter / ((area_i + area_j) - inter);
}

constexpr int kNmsMaxN = 1024;

// one wave per (frame, class): sort by (score desc, index asc), then the greedy pass with the kept boxes in LDS
__global__ __launch_bounds__(64) void nms_bev_kernel(int n, int C, int max_out, float thr,
                                                     const float *__restrict__ bev,
                                                     const float *__restrict__ scores,
                                                     int *__restrict__ out_idx, int *__restrict__ out_cnt) {
    __shared__ unsigned long long s_key[kNmsMaxN];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    float4 *s_box = (float4 *)dyn;                       // [max_out]
    const int b = blockIdx.y, c = blockIdx.x, lane = threadIdx.x;
    const float *sc = scores + (size_t)b * n * C + c;
    const float4 *bx = (const float4 *)(bev + (size_t)b * n * 4);
    int npad = 64;
    while (npad < n) npad <<= 1;
    for (int i = lane; i < npad; i += 64) {
        unsigned long long k = ~0ull;
        if (i < n) {
            // descending score, ascending index: order-preserving float -> uint map (sign bit set: flip all bits,
            // else flip the sign bit), complemented; any finite score, negative ones included, sorts correctly
            unsigned u = __float_as_uint(sc[(size_t)i * C]);
            u = u == 0x80000000u ? 0u : u;                 // -0.0 == +0.0: the index decides, as in a float compare
            const unsigned asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            k = ((unsigned long long)(~asc) << 32) | (unsigned)i;
        }
        s_key[i] = k;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int kk = 2; kk <= npad; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int q = lane; q < npad / 2; q += 64) {
                const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1));
                const int l = i | j;
                const unsigned long long x = s_key[i], y = s_key[l];
                const bool up = (i & kk) == 0;
                if ((x > y) == up) { s_key[i] = y; s_key[l] = x; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    int nsel = 0;
    int *oi = out_idx + ((size_t)b * C + c) * max_out;
    for (int i = 0; i < n && nsel < max_out; ++i) {
        const int cand = (int)(s_key[i] & 0xFFFFFFFFull);
        const float4 cb = bx[cand];
        bool hit = false;
        for (int t = lane; t < nsel; t += 64) hit = hit || (bev_iou(cb, s_box[t]) > thr);
        if (__ballot(hit) == 0ull) {                    // keep: IoU <= threshold with every kept box
            if (lane == 0) { s_box[nsel] = cb; oi[nsel] = cand; }
            ++nsel;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    for (int t = nsel + lane; t < max_out; t += 64) oi[t] = -1;
    if (lane == 0) out_cnt[(size_t)b * C + c] = nsel;
}

// the kept candidates of every (frame, class) as fixed-size rows: box, score, class id (zeros / -1 behind the count)
__global__ __launch_bounds__(256) void nms_gather_kernel(long total, int n, int C, int K, int kbox, const float *__restrict__ boxes,
                                                         const float *__restrict__ scores, const int *__restrict__ idx,
                                                         float *__restrict__ ob, float *__restrict__ os, int *__restrict__ oc) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int k = (int)(t % K);
        const int c = (int)((t / K) % C);
        const long b = t / ((long)K * C);
        const int i = idx[t];
        const bool ok = i >= 0;
        const int r = c < kbox ? c : kbox - 1;                              // class-aware boxes: reg_i = min(i, k - 1), postprocessor.py:76-80
        const float *bp = boxes + ((b * n + (ok ? i : 0)) * kbox + r) * 7;
#pragma unroll
        for (int e = 0; e < 7; ++e) ob[t * 7 + e] = ok ? bp[e] : 0.0f;
        os[t] = ok ? scores[(b * n + i) * C + c] : 0.0f;
        oc[t] = ok ? c : -1;
    }
}

}  // namespace

// The rows sa_nms_bev kept, as the reference's post-processor returns them (lib/builder/postprocessor.py:90-118: gather of
// boxes / scores per class, class id), fixed-size: out_boxes [b,C*max_out,7], out_scores / out_cls [b,C*max_out]; rows
// behind a class's count are zero with class -1.  boxes [b,n,kbox,7] (kbox = 1: class-agnostic), scores [b,n,C], idx [b,C,max_out].
extern "C" int sa_nms_gather(int b, int n, int C, int max_out, int kbox, const float *boxes, const float *scores, const int *idx,
                             float *out_boxes, float *out_scores, int *out_cls, hipStream_t stream) {
    if (b <= 0 || n <= 0 || C <= 0 || max_out <= 0 || kbox <= 0 || !boxes || !scores || !idx || !out_boxes || !out_scores || !out_cls)
        return SA_ERR_INVALID;
    const long total = (long)b * C * max_out;
    const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(nms_gather_kernel, dim3(grid), dim3(256), 0, stream, total, n, C, max_out, kbox, boxes, scores, idx, out_boxes,
                       out_scores, out_cls);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// xyz [b,n,3], reg [b,n,6+2A] (offsets | angle cls | angle res), cls [b,n,C] -> boxes [b,n,7], scores [b,n,C],
// bev [b,n,4].  Additional to the reference API (there it is TF graph code).
extern "C" int sa_decode_anchor_free(int b, int n, int A, int C, const float *xyz, const float *reg, const float *cls,
                                     float *boxes, float *scores, float *bev, hipStream_t stream) {
    if (b <= 0 || n <= 0 || A <= 0 || C <= 0 || !xyz || !reg || !cls || !boxes || !scores || !bev) return SA_ERR_INVALID;
    const long total = (long)b * n;
    const float interval = (float)(2.0 * 3.14159265358979323846 / (double)A);
    const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(decode_anchor_free_kernel, dim3(grid), dim3(256), 0, stream, total, A, C, interval, xyz, reg,
                       cls, boxes, scores, bev);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// bev [b,n,4], scores [b,n,C] -> idx [b,C,max_out] (kept candidate indices in selection order, padded with -1),
// cnt [b,C].  tf.image.non_max_suppression(boxes, scores, max_output_size, iou_threshold) per frame and class.
extern "C" int sa_nms_bev(int b, int n, int C, int max_out, float iou_threshold, const float *bev, const float *scores,
                          int *idx, int *cnt, hipStream_t stream) {
    if (b <= 0 || n <= 0 || C <= 0 || max_out <= 0 || !bev || !scores || !idx || !cnt) return SA_ERR_INVALID;
    if (n > kNmsMaxN || max_out > 2048) return SA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(nms_bev_kernel, dim3(C, b), dim3(64), (size_t)max_out * sizeof(float4), stream, n, C, max_out,
                       iou_threshold, bev, scores, idx, cnt);
    SA_CHECK_LAUNCH();
    return SA_OK;
}

// boxes [nboxes,7] = [x,y,z,l,h,w,ry] -> bev [nboxes,4] = [x_min,z_min,x_max,z_max]
// (box_3d_to_anchor + project_to_bev, lib/utils/box_3d_utils.py:25-58, lib/utils/anchors_util.py:11-50)
extern "C" int sa_boxes_to_bev(long nboxes, const float *boxes, float *bev, hipStream_t stream) {
    if (nboxes <= 0 || !boxes || !bev) return SA_ERR_INVALID;
    const int grid = (int)((nboxes + 255) / 256 < 1024 ? (nboxes + 255) / 256 : 1024);
    hipLaunchKernelGGL(boxes_to_bev_kernel, dim3(grid), dim3(256), 0, stream, nboxes, boxes, bev);
    SA_CHECK_LAUNCH();
    return SA_OK;
}
