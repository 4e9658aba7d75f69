/*
 * sa_ops.h -- C ABI of lib3dssd_sa.so, the gfx950 (MI355X) implementation of the 3DSSD
 * set-abstraction hot path.
 *
 * Drop-in boundary.  The reference binds its CUDA kernels through plain C++ launcher functions that
 * the TensorFlow OpKernels call with raw device pointers (declared in tf_sampling.cpp /
 * tf_grouping.cpp, defined in the *_g.cu files).  Every entry below that cites a reference launcher
 * takes the same scalars and device pointers in the same order, plus an explicit hipStream_t (the
 * reference launches on the legacy default stream, e.g. tf_sampling_g.cu:393) and returns a status
 * instead of void (the reference never checks cudaGetLastError).
 *
 * Conventions: all tensors are dense, row-major, channel-last, device resident; float = fp32,
 * int = int32.  The library never allocates and keeps no state; outputs and scratch are owned by
 * the caller (TF allocate_output / allocate_temp analogue, tf_sampling.cpp:149-155).  Functions are
 * re-entrant.  Return: 0 ok, -1 invalid argument (the reference's OP_REQUIRES conditions), -2 launch
 * failure (hipGetLastError), -3 unsupported size.  hipStream_t is passed as void*.
 */
#ifndef SA_OPS_H
#define SA_OPS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *sa_stream_t; /* hipStream_t */

/* ---- sampling: lib/utils/tf_ops/sampling ------------------------------------------------ */

/* farthestpointsamplingLauncher(b,n,c,m,inp,temp,out)   tf_sampling.cpp:131, tf_sampling_g.cu:123-178,392-394
 * inp [b,n,c], temp [b,n] scratch (only touched when c != 3 or n > 16384), out [b,m] int32. */
int sa_farthest_point_sample(int b, int n, int c, int m, const float *inp, float *temp, int *out,
                             sa_stream_t stream);

/* farthestpointsamplingwithdistLauncher(b,n,m,inp,temp,out)   tf_sampling.cpp:164, tf_sampling_g.cu:180-230,396-398
 * dist [b,n,n], temp [b,n] scratch (only touched when n > 16384), out [b,m] int32. */
int sa_farthest_point_sample_with_distance(int b, int n, int m, const float *dist, float *temp,
                                           int *out, sa_stream_t stream);

/* gatherpointLauncher(b,n,m,c,inp,idx,out)   tf_sampling.cpp:235, tf_sampling_g.cu:320-331,403-407 */
int sa_gather_point(int b, int n, int m, int c, const float *inp, const int *idx, float *out,
                    sa_stream_t stream);

/* ---- grouping: lib/utils/tf_ops/grouping ------------------------------------------------ */

/* queryBallPointLauncher(b,n,m,radius,nsample,xyz1,xyz2,idx,pts_cnt)   tf_grouping.cpp:270, tf_grouping_g.cu:215-255,461-464
 * xyz1 [b,n,3], xyz2 [b,m,3], idx [b,m,nsample], pts_cnt [b,m].  Rows of empty balls are zero-filled
 * (the reference leaves them unwritten). */
int sa_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1,
                        const float *xyz2, int *idx, int *pts_cnt, sa_stream_t stream);

/* queryBallPointDilatedLauncher(b,n,m,min_radius,max_radius,nsample,...)   tf_grouping.cpp:363, tf_grouping_g.cu:308-357,465-468 */
int sa_query_ball_point_dilated(int b, int n, int m, float min_radius, float max_radius, int nsample,
                                const float *xyz1, const float *xyz2, int *idx, int *pts_cnt,
                                sa_stream_t stream);

/* groupPointLauncher(b,n,c,m,nsample,points,idx,out)   tf_grouping.cpp:446, tf_grouping_g.cu:362-379,476-479 */
int sa_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                   float *out, sa_stream_t stream);

/* ---- F-FPS distance matrix: lib/utils/model_util.py:144-160 (calc_square_dist, norm=False) ----
 * In the reference this is TensorFlow graph code (tf.matmul); a [bs,n,c], bb [bs,m,c] -> out [bs,n,m]. */
int sa_calc_square_dist(int b, int n, int m, int c, const float *a, const float *bb, float *out,
                        sa_stream_t stream);

/* ======== additional entry points (no counterpart in the reference API) ==================== */

/* Same kernels with an output row stride and an index offset, so the F-FPS and D-FPS halves of an
 * 'FS' layer land in one [b, npoint_total] tensor (layers_util.py:96-98,108). */
int sa_fps_ex(int b, int n, int c, int m, const float *inp, float *temp, int *out, int out_stride,
              int idx_off, sa_stream_t stream);
int sa_fps_with_distance_ex(int b, int n, int m, const float *dist, float *temp, int *out,
                            int out_stride, int idx_off, sa_stream_t stream);
/* D-FPS (c == 3, n <= 16384) with Morton-bucket culling: same output as sa_fps_ex, which dispatches to it
 * for large frames (3dssd_amd/csrc/fps_bucket.hip). */
int sa_fps_bucket_ex(int b, int n, int m, const float *inp, int *out, int out_stride, int idx_off,
                     sa_stream_t stream);
/* Diagnostic: sa_fps_bucket_ex's picks plus, per frame, stats[2f] = bucket re-evaluations (256 point slots each) and
 * stats[2f+1] = m - 1; the reference evaluates (m - 1) * n pairs.  stats: device, 2*b unsigned 64-bit words. */
int sa_fps_bucket_stats(int b, int n, int m, const float *inp, int *out, unsigned long long *stats, sa_stream_t stream);
/* The samplers with two fused neighbours of theirs in pointnet_sa_module_msg (layers_util.py:84-119), each one launch
 * less per layer: in_bstride = elements between consecutive frames of inp (0 = dense), so a range slice xyz[:, s:e] of
 * a larger tensor is sampled in place (tf.slice, layers_util.py:85-86); ctr != NULL additionally receives the picked
 * points, frame f pick i at ctr + f * ctr_bstride + 3 * i (the gather_point of layers_util.py:116-119; the matrix
 * sampler reads them from xyz, frames xyz_bstride floats apart).  Register-resident kernels only (c == 3 and
 * n <= 16384 / n <= 16384): SA_ERR_UNSUPPORTED otherwise. */
int sa_fps_ex2(int b, int n, int c, int m, const float *inp, long in_bstride, float *temp, int *out, int out_stride,
               int idx_off, float *ctr, long ctr_bstride, sa_stream_t stream);
/* sa_fps_ex2 with flags.  bit 0: frames that need the multi-workgroup sampler (n > 16384 or c != 3: csrc/fps_coop.hip)
 * may launch it PLAINLY on a capturing stream -- the caller guarantees that all such launches of the process are issued on
 * one stream (partner workgroups of two interleaved grids can wait for each other for ever).  Without the bit (and in
 * sa_fps_ex / sa_fps_ex2 / sa_farthest_point_sample) a capturing stream gets the single-workgroup kernels, which are
 * safe on any number of streams; eager calls use a cooperative launch either way. */
int sa_fps_ex3(int b, int n, int c, int m, const float *inp, long in_bstride, float *temp, int *out, int out_stride,
               int idx_off, float *ctr, long ctr_bstride, int flags, sa_stream_t stream);
/* Sticky error word of the multi-workgroup samplers (sa_fps_ex* beyond one workgroup per frame, sa_ffps_fly_ex): 0 =
 * fine; bit 0 / bit 1: a D-FPS / an on-the-fly F-FPS launch gave up waiting for partner workgroups (they were not all
 * resident: the one-stream rule above was violated) -- that launch's outputs are invalid, and every further call of
 * these samplers returns -4 until the word is cleared (reset != 0).  Reads host memory only: no synchronisation. */
int sa_coop_error_state(int reset);
/* Test hook: sa_fps_ex's multi-workgroup launch with the LAST workgroup missing and a short poll bound. */
int sa_debug_fps_coop_orphan(int b, int n, int c, int m, const float *inp, float *temp, int *out, unsigned max_spin,
                             sa_stream_t stream);
int sa_fps_bucket_ex2(int b, int n, int m, const float *inp, long in_bstride, int *out, int out_stride, int idx_off,
                      float *ctr, long ctr_bstride, sa_stream_t stream);
int sa_fps_with_distance_ex2(int b, int n, int m, const float *dist, float *temp, int *out, int out_stride,
                             int idx_off, const float *xyz, long xyz_bstride, float *ctr, long ctr_bstride,
                             sa_stream_t stream);
/* The matrix sampler (F-FPS on dist [b,nf,nf], mf picks, centres read from xyz_f) and the coordinate sampler (D-FPS on
 * inp [b,nd,3], md picks) of one SA layer in ONE launch: two independent serial chains side by side, the launch lasts
 * as long as the longer (layers_util.py:93-106).  Arguments as in the two _ex2 entry points.  SA_ERR_UNSUPPORTED unless
 * both fit the register-resident kernels of the same points-per-thread class (n <= 4096). */
int sa_fps_dual_ex(int b, int nf, int mf, const float *dist, int *out_f, int out_stride_f, int idx_off_f,
                   const float *xyz_f, long xyz_bstride_f, float *ctr_f, long ctr_bstride_f, int nd, int md,
                   const float *inp, long in_bstride, int *out_d, int out_stride_d, int idx_off_d, float *ctr_d,
                   long ctr_bstride_d, sa_stream_t stream);
/* F-FPS without the distance matrix (csrc/ffps_fly.hip): the composition farthest_point_sample_with_distance(m,
 * calc_square_dist(concat(xyz, feat))) of layers_util.py:94-96,102-104 (model_util.py:144-160, tf_sampling_g.cu:180-230)
 * with every needed row of the matrix computed on the fly -- same picks, bit for bit.  xyz [b,.,3] / feat [b,.,c1] point
 * at the start of the sampled range (frames xyz_bstride / feat_bstride floats apart, 0 = dense); workspace = caller-owned
 * device memory of sa_ffps_fly_ws_bytes(b, n) bytes; out / idx_off / ctr as in sa_fps_with_distance_ex2.  n / 1024
 * workgroups share a frame: all calls of a process must be issued on ONE stream at a time (see the file header).
 * SA_ERR_UNSUPPORTED unless c1 == 64 and n is 1024, 2048 or 4096 (the caller then builds the matrix). */
unsigned long sa_ffps_fly_ws_bytes(int b, int n);
int sa_ffps_fly_ex(int b, int n, int c1, int m, const float *xyz, long xyz_bstride, const float *feat, long feat_bstride,
                   void *workspace, int *out, int out_stride, int idx_off, float *ctr, long ctr_bstride, sa_stream_t stream);
/* Up to four strided block copies in one launch (the tf.slice calls of single_stage_detector.py:117-118 and
 * layers_util.py:85-86): jobs = host array of njobs records of 9 longs {src, dst, frames, rows, cols,
 * src_frame_stride, src_row_stride, dst_frame_stride, dst_row_stride}, pointers as integers, strides in floats;
 * dst[f, r, 0:cols] = src[f, r, 0:cols]. */
int sa_copy_blocks(int njobs, const long *jobs, sa_stream_t stream);
/* Up to 32 dense batches of bytes_per_batch bytes each (a multiple of 16; pointers 16-byte aligned) copied back to back
 * into dst in one launch: the executor's package fill (3dssd_amd/pipeline.py). */
int sa_copy_batches(int n, const void *const *srcs, void *dst, long bytes_per_batch, sa_stream_t stream);
/* Forces the global-scratch kernels (mode 0: points [b,n,c], mode 1: matrix [b,n,n]); test hook. */
int sa_fps_generic(int b, int n, int c, int m, const float *inp, float *temp, int *out, int mode,
                   sa_stream_t stream);

/* calc_square_dist on rows given as two pieces [a0 | a1] (concat([xyz, feat]) never materialised,
 * layers_util.py:94,102). */
int sa_calc_square_dist_split(int b, int n, int m, int c0, int c1, const float *a0, const float *a1,
                              const float *b0, const float *b1, float *out, sa_stream_t stream);
/* The same with caller-owned scratch (device memory of sa_calc_square_dist_ws_bytes(b, n, m, c0 + c1, symmetric)
 * bytes, 16-byte aligned; symmetric = a and bb are the same operand): each operand is first packed once into the
 * matrix kernel's LDS image (3dssd_amd/csrc/sqdist.hip, third form).  Bit-identical output; workspace == NULL falls
 * back to the scratch-free form. */
unsigned long sa_calc_square_dist_ws_bytes(int b, int n, int m, int c, int symmetric);
int sa_calc_square_dist_split_ws(int b, int n, int m, int c0, int c1, const float *a0, const float *a1,
                                 const float *b0, const float *b1, float *out, void *workspace, sa_stream_t stream);
/* The symmetric F-FPS case (a == b) on a range slice read in place: frames of a0 / a1 are rs0 / rs1 ROWS apart
 * (0 = n).  Packed form only: SA_ERR_UNSUPPORTED when it cannot run (no workspace, very wide rows). */
int sa_calc_square_dist_self_ws(int b, int n, int c0, int c1, const float *a0, int rs0, const float *a1, int rs1,
                                float *out, void *workspace, sa_stream_t stream);

/* All radius bands of one SA layer in one pass (layers_util.py:134-147).  rmin/rmax/ns: host arrays of
 * nbands entries; idx/cnt: host arrays of nbands device pointers.  dilated=0 ignores rmin. */
int sa_query_ball_point_multi(int b, int n, int m, int nbands, const float *rmin, const float *rmax,
                              const int *ns, int dilated, const float *xyz1, const float *xyz2,
                              int *const *idx, int *const *cnt, sa_stream_t stream);

/* The same through a uniform x-z grid built per frame (3dssd_amd/csrc/ballquery_grid.hip): identical outputs,
 * ~10x fewer distance evaluations on large frames.  workspace: caller-owned, 16-byte aligned device memory of
 * sa_query_ball_point_grid_ws_bytes(b, n, m) bytes (the frames' grids, then -- round 6 -- one 48-byte record per query:
 * the size depends on m); nbands <= 4.  sum(ns) <= 192 runs the sorting form of the query kernel, larger sums the list
 * form, ns[i] > 256 the scan kernels: identical outputs. */
unsigned long sa_query_ball_point_grid_ws_bytes(int b, int n, int m);
int sa_query_ball_point_grid(int b, int n, int m, int nbands, const float *rmin, const float *rmax, const int *ns,
                             int dilated, const float *xyz1, const float *xyz2, int *const *idx, int *const *cnt,
                             void *workspace, sa_stream_t stream);
/* flags bit 0: `workspace` still holds the grid an earlier, stream-ordered call built over the SAME xyz1 contents (same
 * b, n; the workspace sized for an m at least as large): kept if its cells are wide enough for these radii (decided on
 * the device), rebuilt otherwise; the query records are rewritten by every call.  The per-band
 * calls of the reference's stand-alone ops over one point set (tf_grouping.py:53-83) share one grid this way. */
int sa_query_ball_point_grid_ex(int b, int n, int m, int nbands, const float *rmin, const float *rmax, const int *ns,
                                int dilated, const float *xyz1, const float *xyz2, int *const *idx, int *const *cnt,
                                void *workspace, int flags, sa_stream_t stream);

/* One scale of pointnet_sa_module_msg fused: mask, group, concat [features, rel-xyz], nl x
 * (conv1x1 + folded BN + ReLU), max over nsample, empty-ball mask (layers_util.py:157-181).
 * dims[0] = c+3, dims[l+1] = output channels of layer l; wpack[l]/bias[l] device pointers in the
 * layouts documented in 3dssd_amd/csrc/mlp.hip; out[(b*m+j)*out_stride + out_off + ch].
 * Only the DISTINCT rows of a ball are evaluated: the ball query pads a ball of cnt < nsample points with copies of
 * its first hit (tf_grouping_g.cu:245-248), those rows give identical outputs and the max ignores them -- same
 * result bit for bit, a fraction of the work on KITTI-like clouds (3dssd_amd/csrc/mlp_plan.h).  ws: caller-owned
 * device scratch of sa_group_mlp_max_ws_bytes(b, m, ns) bytes (the per-call row plan: header, one entry per 8-row
 * granule of the densest plan, one summary per 4096 balls).  flags bit 0: evaluate all
 * nsample rows of every ball instead (A/B measurements); bit 1: the plan in ws was built by sa_group_mlp_plan (bit 6
 * with it: by sa_group_mlp_plan2 in 4-row granules);
 * bit 2: wpack[] holds single-plane fp16 fragments and the scale runs one fp16 MFMA pass per k-step (fp32
 * accumulate) instead of the three split-bf16 passes -- chosen per scale by the host (utils/weights.py).
 * bit 3 (8): keep the 64-row / streamed fused kernels where the 96-row kernel of csrc/mlp_wide128.hip would be taken
 * (default for the widest fp16 scales: second hidden width x last width >= 512 x 1024); bit 5 (32): take the 96-row kernel
 * for every shape it supports (A/B measurements, tests: the results are bit-identical either way).
 * overflow (device int, may be NULL; fp16 form only): OR-ed with 1 when an input feature or a hidden activation left
 * the fp16 range (|x| > 65504) -- the result of that call is then unspecified; the word is sticky, the caller zeroes
 * and reads it (3dssd_amd/csrc/mlp_act.h).
 * Preconditions of the default (distinct-row) mode: idx rows are in the ball query's output format -- entries
 * cnt .. ns-1 of a row repeat entry 0 (tf_grouping_g.cu:245-248), which sa_query_ball_point* guarantee; a caller that
 * builds idx / cnt itself passes flags bit 0.  ns <= 512 and b*m < 2^24 (row-plan entry fields): SA_ERR_UNSUPPORTED
 * beyond that. */
unsigned long sa_group_mlp_max_ws_bytes(int b, int m, int ns);
/* The same plus, for a scale the GEMM chain of 3dssd_amd/csrc/mlp_gemm.hip can take (three layers, fp16, c a multiple of
 * 8, hidden widths multiples of 32 and >= 128, last width >= 256: layer4 of 3dssd.yaml), room for the packed fp16
 * hidden activations of the densest row plan.  OPT-IN: sa_group_mlp_max / _layer take the chain -- three launches over
 * large tiles, every weight byte fetched once per 128 rows -- only with flags bit 4 (16) set and ws_bytes >= this value;
 * otherwise they run the one-launch fused kernels, which measured faster on the reference shapes (the intermediates
 * round-trip through Infinity Cache / HBM).  Bit-identical results either way. */
unsigned long sa_group_mlp_gemm_ws_bytes(int b, int m, int ns, int c, int nl, const int *dims);
/* The row plans of all scales of a layer in ONE launch: cnt[i] = pts_cnt of scale i, ws[i] = that scale's scratch,
 * out_off[i] / nout[i] = where scale i's channels go in out.  The layer's sa_group_mlp_max calls then pass
 * flags | 2 (plan already built).  flags bit 7 (128): the next-fit packing of rounds 2-5 (a ball of <= 32 rows never crosses
 * a 32-row tile; the open tile is padded instead) for A/B measurements; default since round 6: TIGHT packing -- no padding,
 * a ball that crosses a tile boundary is combined through the atomic max that balls of more than 32 rows always used (its
 * row of `out` is zeroed by the plan launch).  Same results, 5-22 % fewer rows on the wide scales. */
int sa_group_mlp_plan(int b, int m, int nscale, const int *ns, const int *const *cnt, void *const *ws, float *out,
                      int out_stride, const int *out_off, const int *nout, int flags, sa_stream_t stream);
/* The same with a granule size per scale: bit 6 (64) of scale_flags[i] builds scale i's plan in granules of FOUR rows
 * instead of eight (half the padding on balls of 1-4 distinct rows: the inner bands of layer 1 / layer 2 on KITTI-like
 * frames).  Only the row-wave kernels read such plans: ask sa_group_mlp_granule_rows() which scales they take, and pass
 * the same bit 6 in that scale's flags to sa_group_mlp_max / sa_group_mlp_max_layer (a scale whose flags say "4 rows"
 * but which no row-wave kernel can take returns -3).  scale_flags == NULL: eight rows everywhere (= sa_group_mlp_plan).
 * Bit 8 (256) of scale_flags[i]: granules of TWO rows (round 6; tight packing only -- flags bit 7 with it returns -3 --
 * and read by sa_group_mlp_max_layer's one-launch kernels only: the combinations instantiated in mlp_rowwave.hip, any
 * other returns -3; sa_group_mlp_max refuses such a plan). */
int sa_group_mlp_plan2(int b, int m, int nscale, const int *ns, const int *const *cnt, void *const *ws, float *out,
                       int out_stride, const int *out_off, const int *nout, int flags, const int *scale_flags,
                       sa_stream_t stream);
/* 4 or 8: the granule size the plan of this scale should be built with (4 = a row-wave kernel of mlp_rowwave.hip will
 * take it under these flags; nothing is launched).  wpack: the scale's packed layers; ws_bytes: its scratch size. */
int sa_group_mlp_granule_rows(int b, int n, int m, int ns, int c, int nl, const int *dims, const void *const *wpack,
                              unsigned long ws_bytes, int flags);
int sa_group_mlp_max(int b, int n, int m, int ns, int c, const float *xyz, const float *feat,
                     const float *new_xyz, const int *idx, const int *cnt, int nl, const int *dims,
                     const void *const *wpack, const float *const *bias, float *out, int out_stride,
                     int out_off, void *ws, unsigned long ws_bytes, int flags, int *overflow, sa_stream_t stream);

/* All scales of one SA layer in one call: scale i has nsample ns[i], idx[i] / cnt[i], layer widths
 * dims[i*(nl+1) .. (i+1)*(nl+1)), weights wpack[i*nl ..] / bias[i*nl ..], output slice out_off[i], plan scratch ws[i]
 * (ws_bytes[i]) and flags[i] as in sa_group_mlp_max.  Three-scale layers of the reference configuration whose plans
 * come from sa_group_mlp_plan are ONE launch (the scales are independent); anything else is the per-scale loop. */
int sa_group_mlp_max_layer(int nscale, int b, int n, int m, const int *ns, int c, const float *xyz, const float *feat,
                           const float *new_xyz, const int *const *idx, const int *const *cnt, int nl, const int *dims,
                           const void *const *wpack, const float *const *bias, float *out, int out_stride,
                           const int *out_off, void *const *ws, const unsigned long *ws_bytes, const int *flags,
                           int *overflow, sa_stream_t stream);

/* y[rows,N] = act(x[rows,K] W + b): tf_util.conv1d 1x1 + folded BN (tf_util.py:51-124).  Two kernels, the same bits:
 * 32-row workgroups, and 128-row x 128-column blocks when K % 192 == 0, N % 128 == 0 and there are >= 192 blocks. */
int sa_dense(long rows, int K, int N, const float *x, const void *wpack, const float *bias, int relu,
             float *y, sa_stream_t stream);

/* vote_layer tail (layers_util.py:21-23): out = xyz + clip(off, lo, -lo), lo = MAX_TRANSLATE_RANGE. */
int sa_vote_translate(long npoints, const float *xyz, const float *off, float lo_x, float lo_y,
                      float lo_z, float *out, sa_stream_t stream);
/* The whole vote_layer tail in one launch (layers_util.py:17-23): hidden = relu(x W1 + b1) [rows,H] (the layer's
 * feature output), offsets = hidden W2 + b2 [rows,3] (no activation), out = xyz + clip(offsets, lo, -lo).  The same
 * bits as sa_dense, sa_dense, sa_vote_translate.  H <= 128: SA_ERR_UNSUPPORTED otherwise. */
int sa_vote_tail(long rows, int K, int H, const float *x, const void *w1pack, const float *bias1, const void *w2pack,
                 const float *bias2, float *hidden, float *offsets, const float *xyz, float lo_x, float lo_y,
                 float lo_z, float *out, sa_stream_t stream);

/* ---- next row after the backbone: TF graph code in the reference ------------------- */

/* decode_dist_anchor_free + decode_class2angle (lib/utils/anchor_decoder.py:6-14,86-112), sigmoid scores
 * (lib/modeling/single_stage_detector.py:211-212) and the BEV box of every prediction (box_3d_utils.py:25-58,
 * anchors_util.py:11-50).  reg [b,n,6+2A] = offsets | angle cls | angle res; boxes [b,n,7]; bev [b,n,4]. */
int sa_decode_anchor_free(int b, int n, int A, int C, const float *xyz, const float *reg, const float *cls,
                          float *boxes, float *scores, float *bev, sa_stream_t stream);
int sa_boxes_to_bev(long nboxes, const float *boxes, float *bev, sa_stream_t stream);
/* tf.image.non_max_suppression per frame and class (lib/builder/postprocessor.py:76-88): idx [b,C,max_out] kept
 * candidate indices in selection order padded with -1, cnt [b,C].  Equal scores: lower index first. */
int sa_nms_bev(int b, int n, int C, int max_out, float iou_threshold, const float *bev, const float *scores,
               int *idx, int *cnt, sa_stream_t stream);
/* the rows sa_nms_bev kept as fixed-size tensors (lib/builder/postprocessor.py:90-118): out_boxes [b,C*max_out,7],
 * out_scores / out_cls [b,C*max_out], zero rows with class -1 behind a class's count.  boxes [b,n,kbox,7] (kbox = 1:
 * class-agnostic; class i reads box set min(i, kbox-1), :76-80), scores [b,n,C], idx [b,C,max_out]. */
int sa_nms_gather(int b, int n, int C, int max_out, int kbox, const float *boxes, const float *scores, const int *idx,
                  float *out_boxes, float *out_scores, int *out_cls, sa_stream_t stream);

/* ---- lib/utils/tf_ops/interpolation (used by the PointRCNN configurations) ------------ */

/* ThreeNNLauncher(b,n,m,xyz1,xyz2,dist,idx) -- tf_interpolate.cpp:215.  For each unknown point of xyz1 [b,n,3]
 * the three nearest known points of xyz2 [b,m,3]: dist [b,n,3] SQUARED distances ascending, idx [b,n,3]. */
int sa_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx, sa_stream_t stream);
/* ThreeInterpolateLauncher(b,m,c,n,points,idx,weight,out) -- tf_interpolate.cpp:285.  out [b,n,c]. */
int sa_three_interpolate(int b, int m, int c, int n, const float *points, const int *idx, const float *weight,
                         float *out, sa_stream_t stream);
/* KInterpolateLauncher(b,m,c,n,k,points,idx,weight,out) -- tf_interpolate.cpp:407. */
int sa_k_interpolate(int b, int m, int c, int n, int k, const float *points, const int *idx, const float *weight,
                     float *out, sa_stream_t stream);

/* ---- lib/utils/tf_ops/grouping + sampling, remaining operators used by the second stage / training graph --------- */

/* queryBoxes3dPointsLauncher(b,n,m,nsample,xyz,proposals,idx,pts_cnt) -- tf_grouping.cpp:228.  proposals [b,m,7] =
 * (cx, bottom y, cz, l, h, w, ry); idx [b,m,nsample] first points inside each box (short rows repeat the first,
 * empty boxes: zeros), pts_cnt [b,m]. */
int sa_query_boxes_3d_points(int b, int n, int m, int nsample, const float *xyz, const float *proposals, int *idx,
                             int *pts_cnt, sa_stream_t stream);
/* queryBoxes3dMaskLauncher(b,n,m,xyz,boxes_3d,mask) -- tf_grouping.cpp:151.  mask [b,m,n] int. */
int sa_query_boxes_3d_mask(int b, int n, int m, const float *xyz, const float *boxes_3d, int *mask,
                           sa_stream_t stream);
/* queryPointsIouLauncher(b,n,anchors_num,gt_num,xyz,anchors_3d,gt_boxes_3d,iou_matrix,iou_points) --
 * tf_grouping.cpp:182. */
int sa_query_points_iou(int b, int n, int anchors_num, int gt_num, const float *xyz, const float *anchors_3d,
                        const float *gt_boxes_3d, const float *iou_matrix, float *iou_points, sa_stream_t stream);
/* scatteraddpointLauncher(b,n,m,c,out_g,idx,inp_g) -- tf_sampling.cpp:261; inp_g [b,n,c] is zeroed first (the
 * cudaMemset of tf_sampling.cpp:285). */
int sa_gather_point_grad(int b, int n, int m, int c, const float *out_g, const int *idx, float *inp_g,
                         sa_stream_t stream);
/* groupPointGradLauncher(b,n,c,m,nsample,grad_out,idx,grad_points) -- tf_grouping.cpp:479 (+ memset of :509). */
int sa_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                        float *grad_points, sa_stream_t stream);
/* GatherByMaskLauncher(b,n,c,proposal_num,inp,mask,out) -- tf_sampling.cpp:293, plus `sel` [b,proposal_num] int
 * scratch that receives the selected point indices. */
int sa_gather_by_mask(int b, int n, int c, int proposal_num, const float *inp, const float *mask, float *out,
                      int *sel, sa_stream_t stream);

/* queryBallPointWithidxLauncher(b,n,m,radius,nsample,xyz1,xyz2,sort_idx,idx,pts_cnt) -- tf_grouping.cpp:314.
 * sort_idx [b,m,n]: the order in which the dataset points are visited for every query. */
int sa_query_ball_point_withidx(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                                const int *sort_idx, int *idx, int *pts_cnt, sa_stream_t stream);
/* selectionSortLauncher(b,n,m,k,dist,outi,out) -- tf_grouping.cpp:411.  dist/outi/out [b,m,n]; n <= 16384. */
int sa_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out, sa_stream_t stream);
/* The distance matrix knn_point builds with TensorFlow ops (tf_grouping.py:146-150): xyz1 [b,n,c], xyz2 [b,m,c] ->
 * dist [b,m,n] = sum_l (xyz1 - xyz2)^2.  No launcher in the reference (graph ops). */
int sa_pairwise_sqdist(int b, int n, int m, int c, const float *xyz1, const float *xyz2, float *dist,
                       sa_stream_t stream);
/* farthestpointsamplingwithpreidxLauncher(b,n,c,m,m1,inp,preidx,temp,out) -- tf_sampling.cpp:195. */
int sa_farthest_point_sample_with_preidx(int b, int n, int c, int m, int m1, const float *inp, const int *preidx,
                                         float *temp, int *out, sa_stream_t stream);
/* ThreeInterpolateGradLauncher(b,n,c,m,grad_out,idx,weight,grad_points) -- tf_interpolate.cpp:363 and
 * KInterpolateGradLauncher(b,n,c,m,k,...) -- tf_interpolate.cpp:445; grad_points [b,m,c] is zeroed first. */
int sa_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out, const int *idx, const float *weight,
                              float *grad_points, sa_stream_t stream);
int sa_k_interpolate_grad(int b, int n, int c, int m, int k, const float *grad_out, const int *idx,
                          const float *weight, float *grad_points, sa_stream_t stream);

/* ---- host-side helper (no device work) --------------------------------------------------------------------- */
/* CRC-32C of `len` bytes continuing from `crc` (0 to start): the checksum of TensorFlow tensor-bundle checkpoints
 * (tensorflow/core/lib/hash/crc32c.h), used by 3dssd_amd/utils/tf_checkpoint.py when importing the reference's
 * saved weights (lib/core/trainer.py:157-174).  Returns the checksum, not a status. */
unsigned int sa_host_crc32c(const void *data, size_t len, unsigned int crc);

#ifdef __cplusplus
}
#endif
#endif /* SA_OPS_H */
