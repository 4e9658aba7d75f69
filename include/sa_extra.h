/* sa_extra.h -- C ABI of lib3dssd_extra.so: operators of the reference that are NOT on the set-abstraction hot path
 * (SURVEY.md section 2 rows 1 / 14 / 17 "OUT OF SCOPE": second-stage points pooling, the evaluation IoU).  Frozen
 * round-1 surface, built by `make extra` into a library of its own so that lib3dssd_sa.so is the section-8 path and
 * nothing else; conventions (status codes, sa_stream_t) as in sa_ops.h. */
#ifndef SA_EXTRA_H
#define SA_EXTRA_H
#include "sa_ops.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- lib/utils/tf_ops/points_pooling (second-stage pooling, lib/builder/points_pooler.py:81) ------------------- */
/* pointsPoolingLauncher(bs,proposal_num,point_num,channel_num,l,h,w,sample_num,pc,box_3d,pc_loc,out_features,out_idx,
 * sampled_num_lists,pillars) -- tf_points_pooling.cpp:24.  The outputs are zeroed first (the op's cudaMemsets);
 * pillars is [bs,proposal_num,l,h,w,3] (the reference's kernel offsets it by l*h*w floats per proposal); l*h*w <= 2048. */
int sa_points_pooling(int bs, int proposal_num, int point_num, int channel_num, int l, int h, int w, int sample_num,
                      const float *pc, const float *box_3d, const float *pc_loc, float *out_features, int *out_idx,
                      int *sampled_num_lists, float *pillars, sa_stream_t stream);
/* pointsPoolingGradLauncher(...) -- tf_points_pooling.cpp:150 without the shape-only `pc` pointer; pc_grad
 * [bs,proposal_num,point_num,channel_num] is zeroed first. */
int sa_points_pooling_grad(int bs, int proposal_num, int point_num, int channel_num, int l, int h, int w,
                           int sample_num, const int *out_idx, const int *sampled_num_lists, const float *features_grad,
                           float *pc_grad, sa_stream_t stream);

/* ---- lib/utils/tf_ops/evaluation: rotated-box IoU (a CPU op on boost::geometry in the reference) -------------------- */
/* calc_intersections_cpu(dets,gts,det_num,gt_num,num_images,IoU3D,IoUBeV) -- tf_evaluate.cpp:142.  dets [bs,det_num,7],
 * gts [bs,gt_num,7] = (x, bottom y, z, l, h, w, ry) -> iou_bev, iou_3d [bs,det_num,gt_num] (agreement to rounding). */
int sa_calc_iou(int bs, int det_num, int gt_num, const float *dets, const float *gts, float *iou_bev, float *iou_3d,
                sa_stream_t stream);
/* calc_intersections_matching_cpu(dets,gts,bs,IoU3D,IoUBeV) -- tf_evaluate.cpp:182: row i against row i, [n] outputs. */
int sa_calc_iou_match(int n, const float *dets, const float *gts, float *iou_bev, float *iou_3d, sa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SA_EXTRA_H */
