/*
 * retinanet_mi355x.h -- C ABI of libretinanet_mi355x.so (gfx950 / MI355X).
 *
 * The reference (DerekGloudemans/3D-playground) has no native layer: its hot path is Python on top of
 * torch / torchvision / numpy.  This library is the hand-written HIP layer inserted under the same Python
 * surface.  Every entry point takes raw DEVICE pointers, sizes and a hipStream_t (as void*), launches on
 * that stream without synchronising, allocates nothing, and returns 0 (hipSuccess) or a hipError_t /
 * RN_E* code.  Tensors are dense row-major fp32 unless a comment says otherwise.  No torch types cross
 * this boundary; the ctypes binding a reference maintainer would add is shown in INTEGRATION.md.
 *
 * Each function cites the reference code it replaces.  D/ = pytorch_retinanet_detector_directional/retinanet/,
 * R/ = retinanet/ of the reference checkout.
 */
#ifndef RETINANET_MI355X_H
#define RETINANET_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RN_OK 0
#define RN_EINVAL 10001      /* bad size / unsupported shape */
#define RN_ETOOMANY 10002    /* more ground-truth rows per image than RN_MAX_GT */
#define RN_MAX_GT 256
#define RN_LEVELS 5

const char *rn_version(void);
/* Device the calling thread is bound to must be gfx950; returns 0 when it is. */
int rn_check_device(void);

/* ---------------------------------------------------------------- anchors ---------------------------------
 * Replaces Anchors.forward + generate_anchors + shift (D/anchors.py:21-40, 42-73, 109-129): levels 3..7,
 * 9 boxes per cell (ratio-major, scale-minor), fp64 arithmetic with ONE rounding to fp32, order
 * level -> row -> col -> box.  out: [A,4] fp32 (x1,y1,x2,y2). */
int64_t rn_anchor_count(int height, int width);
void rn_anchor_base_boxes(double out[RN_LEVELS * 9 * 4]);           /* host helper, numpy-identical fp64 */
int rn_anchors_fwd(float *out, int height, int width, void *stream);

/* ---------------------------------------------------------------- IoU -------------------------------------
 * Replaces calc_iou (D/losses.py:5-22): a [A,4] x b [N,4] -> iou [A,N]; no +1, union clamped at 1e-8. */
int rn_pairwise_iou(const float *a, const float *b, float *iou, int64_t A, int N, void *stream);

/* ---------------------------------------------------------------- focal / smooth-L1 / VP loss -------------
 * Replaces FocalLoss.forward, directional (D/losses.py:27-362, label_cols = 27, n_reg = 12) and 2D
 * (R/losses.py:27-177, label_cols = 5, n_reg = 4) in ONE launch each way (forward incl. the batch reduction): corner-envelope /
 * box IoU against every valid label row, max + first-argmax, 0.4/0.5 bands, alpha=.25 gamma=2 focal BCE on
 * clamp(cls,1e-4,1-1e-4), and for positives the smooth-L1 (beta 1/9; 20 values with the 0.5 top-corner
 * weight, or 4 std-scaled deltas) and the vanishing-point cosine term.
 *
 *   cls [B,A,C] post-sigmoid, reg [B,A,n_reg], anchors [A,4], ann [B,N,label_cols] (padding rows: class -1)
 *   workspace: rn_focal_workspace_bytes(B, A) bytes, written by fwd, read by bwd (per-image statistics).  Its first
 *           rn_focal_workspace_zero_bytes(B) bytes hold the forward's completion counters (one per image + one for the
 *           batch) and must be ZERO when rn_focal_loss_fwd is called; the kernel leaves them zero, so a workspace can be
 *           reused call after call without clearing it again (re-zero it after an aborted launch).  One workspace
 *           serves one stream at a time.
 *   losses: 3 floats  (cls, reg, vp; vp = 0 for the 2D variant).  An all-empty batch yields vp = NaN
 *           (the reference raises there, D/losses.py:362; the Python binding raises before launching).
 *   bwd: grad_losses = 3 device floats (dL/dcls_loss, dL/dreg_loss, dL/dvp_loss); writes dense
 *        dcls [B,A,C] and dreg [B,A,n_reg] (zeros where no gradient flows).
 */
int64_t rn_focal_workspace_bytes(int B, int64_t A);
int64_t rn_focal_workspace_zero_bytes(int B);
int rn_focal_loss_fwd(const float *cls, const float *reg, const float *anchors, const float *ann,
                      int B, int64_t A, int C, int N, int directional,
                      void *workspace, float *losses, void *stream);
int rn_focal_loss_bwd(const float *cls, const float *reg, const float *anchors, const float *ann,
                      int B, int64_t A, int C, int N, int directional,
                      const void *workspace, const float *grad_losses,
                      float *dcls, float *dreg, void *stream);
/* Assignment only (test / debugging aid): iou_max [B,A] fp32, argmax [B,A] int32 (index among VALID rows,
 * -1 for an image without labels), state [B,A] int32 in {-1 ignore, 0 negative, 1 positive}. */
int rn_assign(const float *anchors, const float *ann, int B, int64_t A, int N, int directional,
              float *iou_max, int32_t *argmax, int32_t *state, void *stream);

/* ---------------------------------------------------------------- decode / clip ---------------------------
 * rn_decode_dir replaces BBoxTransform.forward (D/utils.py:102-149): reg [B,A,12] -> boxes [B,A,20].
 * rn_decode_2d  replaces BBoxTransform.forward (R/utils.py:102-126) with std (.1,.1,.2,.2) and, when
 * clip != 0, ClipBoxes.forward fused (R/utils.py:134-144): deltas [B,A,4] -> boxes [B,A,4].
 * rn_clip_boxes is the standalone in-place ClipBoxes. */
int rn_decode_dir(const float *anchors, const float *reg, float *boxes, int B, int64_t A, void *stream);
/* Decode of the score filter's survivors only (the eval branches decode all B*A anchors at D/model.py:347 and then keep
 * <= 10 000, :322-328 / :368-374): candidate k (k < min(*count, max_candidates)) is flat anchor sel_idx[k] of [B*A];
 * writes boxes [max_candidates,20] (same per-anchor arithmetic as rn_decode_dir: bit-identical rows), cand_scores[k] =
 * scores[sel_idx[k] * score_stride] and, when cand_image != NULL, the image index sel_idx[k] / A (D/model.py:314-316).
 * count stays on the device (no host sync). */
int rn_decode_dir_select(const float *anchors, const float *reg, int64_t A, const float *scores, int64_t score_stride,
                         const int32_t *sel_idx, const int32_t *count, int max_candidates, float *boxes,
                         float *cand_scores, int32_t *cand_image, void *stream);
int rn_decode_2d(const float *anchors, const float *deltas, float *boxes, int B, int64_t A,
                 int clip, float width, float height, void *stream);
int rn_clip_boxes(float *boxes, int64_t n, float width, float height, void *stream);

/* ---------------------------------------------------------------- score filter + NMS ----------------------
 * The eval branch of ResNet.forward (D/model.py:311-397, R/model.py:283-311).
 *
 * rn_rowmax: scores[n] = max_c cls[n,c], classes[n] = first argmax (int64)          (D/model.py:320)
 * rn_threshold_select: the adaptive loop "t = start; repeat { mask = s > t; cnt = sum(mask); t *= 10**0.2 }
 *   until cnt <= keep" (D/model.py:322-328, 368-374) evaluated as one histogram pass; fixed_threshold >= 0
 *   selects the plain "s > thr" of the 2D path instead (R/model.py:289).  scores has `stride` floats between
 *   consecutive elements (class column of a [A,C] matrix).  Outputs, all device: count[0] = number selected,
 *   sel_idx[0..count) = indices in increasing order (the order a boolean mask gives).
 * rn_nms: greedy NMS (torchvision.ops.nms contract: score-descending, suppress IoU > thr; ties by lower
 *   index) over `count[0]` candidates: boxes are rows box_idx[i] of `boxes` (row stride box_stride floats,
 *   4 coords at column box_col); category[i] (may be NULL) reproduces batched_nms' fp32 offset trick
 *   (D/model.py:47-57).  keep[0..keep_count[0]) = positions into the candidate list, score-descending.
 * Workspace sizes are returned by rn_post_workspace_bytes(max_candidates).
 */
int64_t rn_post_workspace_bytes(int64_t n_scores, int64_t max_candidates);
int rn_rowmax(const float *cls, int64_t n, int C, float *scores, int64_t *classes, void *stream);
int rn_threshold_select(const float *scores, int64_t n, int64_t stride, double start, int keep,
                        double fixed_threshold, void *workspace, int32_t *count, int32_t *sel_idx, void *stream);
int rn_nms(const float *boxes, int64_t box_stride, int box_col, const float *scores, int64_t score_stride,
           const int32_t *cand_idx, const int32_t *category, const int32_t *count, int max_candidates,
           float iou_thr, void *workspace, int32_t *keep, int32_t *keep_count, void *stream);

/* ---------------------------------------------------------------- homography ------------------------------
 * Replaces Homography.i24_state_to_space + space_to_im = state_to_im (homography.py:305-320, 438-488) and
 * im_to_space + i24_space_to_state = im_to_state (homography.py:388-435, 274-303, 491-500).
 *   state [d,6] fp32 (x_rear, y_ctr, l, w, h, dir); image points [d,8,2] fp64; heights [d] fp32.
 *   P: 3x4 fp64 row-major, H: 3x3 fp64 row-major.  mat_index (may be NULL) picks a matrix per object
 *   (the reference's name = list of cameras); NULL uses matrix 0 for all (name = str).
 *   P2/H2 non-NULL = Homography_Wrapper (homography.py:840-862): objects whose corner-0 space y > 60 use
 *   the second set. */
int rn_state_to_space(const float *state, float *space, int64_t d, void *stream);
int rn_space_to_state(const double *space, float *state, int64_t d, void *stream);
int rn_state_to_im(const float *state, const double *P, const double *P2, const int32_t *mat_index,
                   double *im, int64_t d, void *stream);
int rn_space_to_im(const float *space, const double *P, const double *P2, const int32_t *mat_index,
                   double *im, int64_t d, void *stream);
int rn_im_to_space(const double *im, const float *heights, const double *H, const double *H2,
                   const int32_t *mat_index, double *space, int64_t d, void *stream);
int rn_im_to_state(const double *im, const float *heights, const double *H, const double *H2,
                   const int32_t *mat_index, float *state, int64_t d, void *stream);

/* ---------------------------------------------------------------- tracker: detection parsing ---------------
 * Replaces MC_Crop_Tracker.parse_detections with im_nms / space_nms (MC3D_crop_tracker.py:319-383, 592-636), the
 * step right after the MULTI_FRAME detector, without leaving the device or synchronising:
 *   keep scores > sigma_d (input order) -> NMS(phi_nms_im) on the image envelopes of the 8 corners, every box
 *   shifted by the constant 10 000 exactly as the reference does (its per-camera offset is computed and dropped,
 *   :610-613) -> image -> state through the camera's H of both wrapper homographies (switch at y > 60); with
 *   refine_height, state -> image through P, height_from_template, image -> state again (:366-370)
 *   -> NMS(phi_nms_space) on the road-plane footprints -> survivors in NMS order.
 *   scores [d] f32, labels [d] i64, boxes20 [d,20] f32 (16 corner coords + 2D box), camera_idxs [d] i64;
 *   H1,H2 [n_cam,3,3] / P1,P2 [n_cam,3,4] fp64 row-major, index = camera index (H2/P2 may be NULL: one homography;
 *   P1 is only read with refine_height); heights [d] f32 per input detection or NULL = 5 ft, which is what
 *   guess_heights returns for the integer labels the tracker passes (homography.py:502-517).
 *   Outputs sized for d rows; out_count[0] = rows valid.  d <= RN_PARSE_MAX.  perform_nms: bit 0 = image NMS, bit 1 =
 *   space NMS (3 = the reference's perform_nms=True, 0 = filter + transforms only, 1 = stop before the space NMS, where
 *   the tracker's estimate_ts_bias (tracker state, :373-374; not part of this call) looks at the boxes).
 * rn_md_iou: MC_Crop_Tracker.md_iou (:1030-1049), element-wise fp64 IoU of a[n,4], b[n,4] (union not clamped). */
#define RN_PARSE_MAX 16384
int64_t rn_parse_workspace_bytes(int64_t d);
int rn_parse_detections(const float *scores, const int64_t *labels, const float *boxes20, const int64_t *camera_idxs,
                        int64_t d, const double *H1, const double *H2, const double *P1, const double *P2, int n_cam,
                        const float *heights, float sigma_d, float phi_nms_im, float phi_nms_space,
                        int perform_nms, int refine_height, void *workspace,
                        float *out_state, int64_t *out_labels, float *out_scores, int64_t *out_cams,
                        int32_t *out_count, void *stream);
int rn_md_iou(const double *a, const double *b, double *out, int64_t n, void *stream);

/* ---------------------------------------------------------------- tracker: crop refinement ----------------
 * The block of MC_Crop_Tracker.track around the LOCALIZE detector (MC3D_crop_tracker.py:1172-1226), device-resident.
 * rn_crop_boxes: get_crop_boxes (:920-944): im_objs [n,8,2] fp64 (state_to_im of the priors) -> crop_boxes [n,4] fp64
 *   (x1,y1,x2,y2), squares of side max(w,h)*b; rois (may be NULL) [n,5] fp32 = (camera index, box), the argument of
 *   roi_align (:1183-1185).
 * rn_roi_align: torchvision.ops.roi_align(frames [N,C,H,W] fp32, rois, (out_h,out_w)) with its defaults
 *   (spatial_scale 1, sampling_ratio -1, aligned False), fp32, torchvision's operation order (third-party: restated,
 *   "parity unpinned"); nhwc4 != 0 writes [n,out_h,out_w,4] (C <= 4, rest 0), the stem convolution's layout.
 * rn_crop_select: everything after the detector (:1192-1226) for n objects: reg_boxes [n,A,20] / cls [n,A,C] (the
 *   LOCALIZE outputs), crop_boxes, cam_idxs [n] i64, pre_loc [n,6] priors -> per object the best of the cd_max most
 *   confident detections by (1-W)*IoU(footprint, prior footprint) + W*conf, as state [n,6] with the height refinement,
 *   class [n] i64, confidence [n].  A <= 4096, cd_max <= 256.  Heights start at 5 ft ("other"): what guess_heights
 *   returns for the integer classes the tracker passes. */
int rn_crop_boxes(const double *im_objs, const int64_t *cam_idxs, int n, double b, double *crop_boxes, float *rois,
                  void *stream);
int rn_roi_align(const float *frames, int N, int C, int H, int W, const float *rois, int n, int out_h, int out_w,
                 float *out, int nhwc4, void *stream);
int rn_crop_select(const float *reg_boxes, const float *cls, const double *crop_boxes, const int64_t *cam_idxs,
                   const float *pre_loc, const double *H1, const double *H2, const double *P1, const double *P2,
                   int n_cam, int n, int A, int C, double cs, int cd_max, float W, float *out_state, int64_t *out_cls,
                   float *out_conf, void *stream);

/* ---------------------------------------------------------------- tracker: Kalman filter -------------------
 * The tensor algebra of Torch_KF (util_track/kf.py:264-403), one lane per object.  X [n,6] fp32 (x,y,l,w,h,v),
 * P [n,6,6] fp32, D [n] fp32 travel direction, T [n] fp64 time stamps, F [6,6], Q [6,6], H [5,6], R [5,5], mu_R [5]
 * fp32.  dt: one fp64 value (dt_is_tensor 0: the reference's Python-float path, all fp32) or [n] fp64 values
 * (dt_is_tensor 1: Q*dt/dt_default is formed in fp64, kf.py:321-326).  F[0][5] is replaced by D*dt per object.
 * rn_kf_view (dt may be NULL = no prediction): out [n,6], or [n,7] with the direction inserted before the speed.
 * rn_kf_predict updates X, P, T in place.  rn_kf_update applies measurements z [m,5] fp64 to the objects rows[m]
 * (distinct rows), innovation in fp64, S^-1 by fp32 Gauss-Jordan with partial pivoting. */
int rn_kf_view(const float *X, const float *D, const float *F, const double *dt, int dt_is_tensor, int with_direction,
               float *out, int n, void *stream);
int rn_kf_predict(float *X, float *P, const float *D, double *T, const float *F, const float *Q, const double *dt,
                  int dt_is_tensor, double dt_default, int n, void *stream);
int rn_kf_update(float *X, float *P, const int32_t *rows, const double *z, const float *H, const float *R,
                 const float *mu_R, int m, void *stream);

/* ---------------------------------------------------------------- frame ingest ----------------------------
 * Replaces F.to_tensor + F.normalize of the reference's loaders (util_track/mp_loader.py:239-243,
 * perform_3D_detection_on_video_sequences.py:51-58) on device: frames uint8 [B,H,W,3] (as the decoder / cv2.resize
 * leaves them) -> fp32 (u8/255 - mean[c]) / std[c], three separate fp32 operations as torchvision performs them
 * (bit-identical to the CPU).  swap_rb != 0 applies the second caller's BGR->RGB swap.
 * layout 0: out = NCHW [B,3,H,W], the tensor the reference hands to the model;
 * layout 1: out = NHWC4 [B,H,W,4] (4th channel 0), the stem convolution's input layout. */
int rn_frame_ingest(const uint8_t *frames, int B, int H, int W, int swap_rb, float mean0, float mean1, float mean2,
                    float std0, float std1, float std2, int layout, float *out, void *stream);

/* ---------------------------------------------------------------- convolution engine ----------------------
 * fp32 implicit-GEMM convolutions on the matrix cores (v_mfma_f32_32x32x2_f32).  Replaces nn.Conv2d +
 * BatchNorm2d(eval) + ReLU + residual add (D/utils.py:25-43, 60-80), PyramidFeatures (D/model.py:84-117) and
 * the head towers (D/model.py:139-157, 182-205), forward and backward (the reference gets the latter from
 * torch autograd / cuDNN).
 *
 * Activations are NHWC fp32.  Weights are re-packed once per optimizer step by rn_pack_weights into
 * [rows][Kpad] with K = kh*kw*Cin contiguous (Kpad = K rounded up to 32, zero filled).
 *
 * rn_conv_igemm computes, for every output pixel (n, oh, ow) and output channel c:
 *     acc = sum_{r<kh, s<kw, ci<Cin} x[n, ih, iw, ci] * w[c][r][s][ci]
 *     with  ih = (oh*a + p + r*b) >> div_shift  (contributes only if the numerator is >= 0, divisible by
 *     1 << div_shift and ih < Hi; iw likewise from ow, s and p_w),
 *     v = scale[c]*acc + shift[c];  mask_mode 1: v = mask[...] > 0 ? v : 0;  v += add[...];  v = act(v);
 *     mask_mode 2: v = mask[...] > 0 ? v : 0;  y = v.   (mask has the geometry of y: ReLU backward of the
 *     tensor the gradient flows into, applied before or after other gradient contributions are added.)
 *     in_relu != 0 applies max(x, 0) to the input as it is loaded (P7 = conv(ReLU(P6)), D/model.py:114-115).
 *   forward conv (stride st, padding pd):    a = st, b = +1, p = p_w = -pd, div_shift = 0
 *   data gradient of that conv:              a = 1,  b = -1, p = p_w = +pd, div_shift = log2(st), x = dY, and
 *                                            w packed as [Cin][kh][kw][Cout] (rn_pack_weights mode 1)
 * add_mode 1: add has the geometry of y (residual / gradient accumulation); 2: add is [N,Ha,Wa,Cout] read at
 * (oh>>1, ow>>1) -- the FPN nearest-upsample + add, cropped to the output (D/model.py:88-108).
 * batch strides are in floats; y_batch_stride lets a head write straight into its slice of the concatenated
 * [B, A, n] tensor (the reference's permute+view+cat, D/model.py:155-157, 302-304).
 *
 * Size limits (RN_EINVAL otherwise): the operands are read through 32-bit buffer offsets, so one input image
 * (Hi*Wi*Cin floats) plus the images a 256-pixel run of outputs can span ((255 / (Ho*Wo) + 1) * x_batch_stride) must
 * stay below 2 GiB, likewise the packed weights, and N*Ho*Wo below 2^31.  x, w_packed need 4-byte alignment only.
 */
typedef struct rn_conv_desc {
    int N, Hi, Wi, Cin;            /* input  [N,Hi,Wi,Cin]; Cin % 4 == 0 */
    int Ho, Wo, Cout;              /* output [N,Ho,Wo,Cout] */
    int kh, kw;
    int a, b, p, p_w, div_shift;   /* p applies to rows, p_w to columns */
    int act;                       /* 0 none, 1 ReLU, 2 sigmoid */
    int add_mode;                  /* 0 none, 1 same geometry, 2 nearest-upsample x2 */
    int Ha, Wa;
    int mask_mode;                 /* 0 none, 1 before the add, 2 after the activation; | RN_MASK_BITS (4): `mask` points to the
                                      masking tensor's SIGN BITS (see sign_out below) instead of the tensor */
    int in_relu;                   /* ReLU applied to x on load */
    int os, oo_h, oo_w, Hy, Wy;    /* output pixel (oh,ow) is stored at (oh*os+oo_h, ow*os+oo_w) of a [N,Hy,Wy,Cout]
                                      tensor (os = 1, offsets 0, Hy = Ho, Wy = Wo: dense).  add (mode 1) and mask
                                      are read at the same place.  Used by the stride-2 data gradient, which is
                                      computed per output-parity class. */
    int add2_mode;                 /* 0 none, 3: add2 is [N,Ha2,Wa2,Cout], added where the stored position has
                                      even row and column, read at (row/2, col/2): gradient of a 1x1 stride-2
                                      shortcut (D/model.py:265-270) without materialising its zeros */
    int Ha2, Wa2;
    int64_t add2_batch_stride;
    int64_t x_batch_stride, y_batch_stride, add_batch_stride;
    int64_t w_batch_stride;        /* 0: one weight tensor.  != 0 (floats): image n uses w_packed + n * w_batch_stride -- a batch of
                                      independent GEMMs in one launch (the 36 positions of the Winograd path); Ho*Wo must then
                                      be a multiple of 256 so that no tile spans two images */
    int w_format;                  /* 0: w_packed is the fp32 tensor of rn_pack_weights; 1: its pre-split form (rn_split_weights),
                                      accepted in RN_FP32_SPLIT mode only (RN_EINVAL otherwise; not by the split-K form);
                                      2 (round 4): the pre-split form, and the products are formed from the operands' FIRST bf16
                                      terms only (one MFMA instead of six) -- bf16 arithmetic on fp32 tensors, for the fp32 stem of
                                      the bf16 / fp8 engines; layers with at most 64 output channels, single launches */
    void *sign_out;                /* NULL, or uint32 words that receive the SIGN BITS of the result: bit (e & 31) of word (e >> 5)
                                      = (y[e] > 0) for every stored element at float offset e -- what the backward pass needs of a ReLU
                                      output (D/utils.py:60-80), at 1/32 of the bytes.  Needs Cout % 32 == 0 and y_batch_stride % 32
                                      == 0.  mask_mode | RN_MASK_BITS (4): `mask` points to such words (the consumer's side). */
    const void *x_amax;            /* RN_FP32_SPLIT3 (round 5): the AMAX TABLES of x: per image 256 bytes, byte e != 0 iff some element of the
                                      image has fp32 exponent field e (what a producer's y_amax left, or rn_amax) -- the kernel takes the
                                      largest exponent present as the image's power-of-two scale for its fp16 split, row by row: GEMM
                                      row (image n, pixel r) uses table n * x_amax_img_stride when x_amax_row_stride == 0 (the usual
                                      form, img stride 1: an image's result then does not depend on what else is in the batch), or the
                                      plain uint32 WORD x_amax[r * x_amax_row_stride] (an fp32 bit pattern whose exponent field bounds
                                      the row) when it is not: the Winograd-stage GEMM, whose rows are tiles (rn_wino_input_group
                                      writes the words).  Required when w_format == 3. */
    void *y_amax;                  /* NULL, or the amax tables of the RESULT, one per image [N][256]: every kernel that finishes elements
                                      sets the byte of the largest exponent it stored in image n (plain stores, idempotent: several
                                      launches may fill one tensor); the caller zeroes the tables before the first.  Any product mode;
                                      not by the raw Winograd-stage GEMM (its result feeds a transform, not a convolution). */
    const float *w_unscale;        /* w_format == 3: per weight row the inverse 2^-s of the power-of-two scale its fp16 terms were
                                      written with (rn_split_weights_f16); [batch * Cout] when w_batch_stride != 0 */
    int x_amax_img_stride, x_amax_row_stride;
} rn_conv_desc;
#define RN_MASK_BITS 4

/* How the fp32 convolution kernels (rn_conv_igemm*, rn_conv_wgrad*, and through them the Winograd GEMMs) form their
 * products.  Operands, accumulation, epilogue and results are fp32 either way.
 *   RN_FP32_NATIVE  v_mfma_f32_32x32x2_f32 (157 TF peak).
 *   RN_FP32_SPLIT   each fp32 operand is split in registers into three bf16 terms h + m + l (exactly equal to it) and a
 *                   product is the sum of the six largest of the nine term products on v_mfma_f32_32x32x16_bf16 with the
 *                   fp32 accumulator; the three dropped terms are at most 2^-23 of the product (2^-25 rms), the size of
 *                   one fp32 rounding (csrc/mfma_split.h; DESIGN.md 4.6 has the measured errors of both modes against fp64).
 *   RN_FP32_SPLIT3  (round 5) each operand, scaled by a power of two so that its tensor's largest magnitude lands in [2^14, 2^15), is
 *                   split into TWO fp16 terms hi + lo (22 significand bits + a sign: |error| <= 2^-22 |x| for elements down to 2^-18
 *                   of the tensor's maximum, an absolute 2^-40 of the maximum below that) and a product is hi*hi + hi*lo + lo*hi
 *                   on v_mfma_f32_16x16x32_f16 / 32x32x16_f16: THREE MFMAs instead of six (csrc/mfma_split.h, second half).  Weight
 *                   scales per output channel (rn_split_weights_f16), activation / gradient scales per tensor from amax words
 *                   (rn_conv_desc.x_amax).  (The split-K form stays on the fp32 MFMA in every mode.)
 * Process-wide; initial value from the environment variable RN_FP32_MFMA = native | split | split3, else RN_FP32_DEFAULT. */
#define RN_FP32_NATIVE 0
#define RN_FP32_SPLIT 1
#define RN_FP32_SPLIT3 2
#define RN_FP32_DEFAULT RN_FP32_SPLIT3
int rn_get_fp32_mfma(void);
int rn_set_fp32_mfma(int mode);
/* Process-wide run-time options; initial value from the environment variable named (read ONCE, at the option's first use --
 * never on a launch path), else the default.  rn_set_option returns RN_EINVAL for a value outside the option's range.
 *   RN_OPT_SPLITK         (RN_SPLITK, 0/1, default 1)  rn_conv_splitk_workspace_bytes may choose the split-K form.  Off, a
 *                         convolution's result does not depend on how many images share the launch: every output element
 *                         is one K loop in a fixed order (the batch-invariance the batch-8 parity test relies on).
 *   RN_OPT_DETERMINISTIC  (RN_DETERMINISTIC, 0/1, default 0)  weight gradients are reduced in a fixed order: every K slice of
 *                         rn_conv_wgrad* stores its partial tile in a slab of the workspace and an ordered pass adds the
 *                         slabs (and the column sums) -- two runs give bit-identical gradients, as the reference's CPU path
 *                         does; costs the slab traffic (rn_conv_wgrad_det_workspace_bytes).  Off: fp32 atomics, fastest,
 *                         last bits depend on arrival order.
 * Kernel selectors of the split-operand family (A/B switches; the defaults are the measured best):
 *   RN_OPT_MF16           (RN_MF16, 0/1, default 1)  wide layers with Cin % 32 == 0 take csrc/conv_igemm_mf16.hip (16x16x32 MFMA).
 *   RN_OPT_MF16_MIN       (RN_MF16_MIN, >= 0, default 1)  fewest tiles a launch needs to take it.
 *   RN_OPT_WGRAD_ONCE     (RN_WGRAD_ONCE, 0/1, default 1)  weight gradient's 128 x 128 tile splits its operands once per workgroup.
 *   RN_OPT_BF16_P8        (RN_BF16_P8, 0..2, default 1)  the bf16 engine's stride-1 same-size convolutions on the eight-wave 256 x 256 x 64
 *                         tile with the phased K loop (csrc/conv_bf16_p8.hip): 0 never, 1 where it measured faster, 2 wherever legal.
 *   RN_OPT_FP8_P8         (RN_FP8_P8, 0..2, default 1)  the same for the fp8 inference engine (csrc/conv_fp8_p8.hip, 256 x 256 x 128). */
#define RN_OPT_SPLITK 0
#define RN_OPT_DETERMINISTIC 1
#define RN_OPT_MF16 2
#define RN_OPT_MF16_MIN 3
#define RN_OPT_WGRAD_ONCE 4
#define RN_OPT_BF16_P8 5
#define RN_OPT_FP8_P8 6
#define RN_OPT_COUNT 7
int rn_get_option(int option);
int rn_set_option(int option, int value);
/* RN_FP32_SPLIT applies to rn_conv_igemm / _grouped launches with kh*kw*Cin >= this (64; environment RN_FP32_SPLIT_MIN_K);
 * shorter reductions keep the fp32 MFMA kernel.  (A pre-split weight operand, w_format 1, is always
 * taken by the split kernels: prepare it for the long reductions only.) */
int rn_fp32_split_min_k(void);
/* RN_FP32_SPLIT: the weights' three bf16 terms can be prepared once per optimizer step instead of in every workgroup:
 * w_split [rows][Kpad/16][3][16] bf16 (6 bytes per element of w_packed [rows][Kpad], Kpad % 16 == 0; for a batch of GEMMs
 * rows = batch * rows).  Pass it as w_packed with rn_conv_desc.w_format = 1.  (rn_prep_batched: job kind 4.) */
int rn_split_weights(const float *w_packed, void *w_split, int64_t rows, int Kpad, void *stream);
/* RN_FP32_SPLIT3's weight operand: [rows][Kpad/16] records of 64 bytes = the hi and lo fp16 terms (2 x 16) of the row's 16 values of
 * a K-step, written with the row's own power-of-two scale (largest |value| of the row -> [2^14, 2^15)); row_unscale[rows] receives
 * the inverse scales.  Pass as w_packed with rn_conv_desc.w_format = 3 and w_unscale = row_unscale.  (rn_prep_batched: job kind 5.) */
int rn_split_weights_f16(const float *w_packed, void *w_split, float *row_unscale, int64_t rows, int Kpad, void *stream);
/* The amax tables of a tensor (rn_conv_desc.x_amax) for tensors no producer left them for: x = n_images images of per_image floats,
 * amax = [n_images][256] bytes, byte e of table i set when an element of image i has exponent field e (only the largest matters; the
 * kernel sets the largest per thread); zero the tables first.  One pass over x. */
int rn_amax(const float *x, int64_t per_image, int n_images, void *amax, void *stream);
/* 1 when rn_conv_igemm would run this problem on an fp16-split kernel (so: wants w_format 3, x_amax, w_unscale): in RN_FP32_SPLIT3 mode,
 * every problem (the 16x16x32 kernel for the wide layers with Cin % 32 == 0, the two-term form of the 128 x 128 / 256 x 64 tile for the
 * rest); 0 in the other modes. */
int rn_conv_igemm_wants_f16(const rn_conv_desc *d);

int rn_conv_igemm(const rn_conv_desc *d, const float *x, const float *w_packed, float *y,
                  const float *scale, const float *shift, const float *add, const float *mask, const float *add2,
                  void *stream);

/* Split-K form for problems with few output tiles and a long K loop (the P6 / P7 pyramid levels, the deep layers of
 * the 112x112 crop detector): rn_conv_splitk_workspace_bytes returns 0 when splitting is not worthwhile, otherwise the
 * scratch size; rn_conv_igemm_splitk then slices K over grid.y, each slice storing its raw partial tile in its own slab
 * (no atomics: results are reproducible), and a finish kernel adds the slabs in order and applies the same epilogue. */
int64_t rn_conv_splitk_workspace_bytes(const rn_conv_desc *d);
int rn_conv_igemm_splitk(const rn_conv_desc *d, const float *x, const float *w_packed, float *y, const float *scale,
                         const float *shift, const float *add, const float *mask, const float *add2, void *workspace,
                         void *stream);

/* Grouped launch: up to RN_MAX_GROUP problems that share the weights and every scalar of the descriptor except the
 * geometry (N, Hi, Wi, Ho, Wo, output map, batch strides) run as ONE grid -- the five pyramid levels of a head tower
 * (D/model.py:302-304 loops over them): the small levels no longer occupy a fraction of the GPU for a full
 * workgroup round each.  tile_end[i] = exclusive prefix sum of the problems' tile counts (tile = 128x128 outputs,
 * 256x64 when Cout <= 64); add2 and in_relu are not available in grouped launches (RN_EINVAL). */
#define RN_MAX_GROUP 5
typedef struct rn_conv_group {
    int n;
    int tile_end[RN_MAX_GROUP];
    rn_conv_desc d[RN_MAX_GROUP];
    const float *x[RN_MAX_GROUP];
    float *y[RN_MAX_GROUP];
    const float *add[RN_MAX_GROUP];
    const float *mask[RN_MAX_GROUP];
} rn_conv_group;
int rn_conv_igemm_grouped(const rn_conv_group *g, const float *w_packed, const float *scale, const float *shift,
                          void *stream);

/* Weight gradient: dw[co][r][s][ci] += sum_{n,oh,ow} dy[n,oh,ow,co] * x[n, oh*st + r - pd, ow*st + s - pd, ci]
 * (fp32 atomics into a zeroed or previously accumulated [Cout][Kpad] buffer, same layout as the packed forward
 * weights; heads accumulate their five pyramid levels into one buffer).  dy: [N,Ho,Wo,Cout] with channel
 * stride ldy >= Cout (a padded copy is allowed), x: [N,Hi,Wi,Cin]; in_relu applies max(x,0) on load.
 * colsum (may be NULL): colsum[co] += sum over pixels of dy[.,co] -- the bias / batch-norm-beta gradient, fused
 * because this kernel streams dy anyway.
 * Size limit: the K range is cut into slices whose dy bytes and span of input images each stay below 2 GiB (the
 * slices are made finer if need be); RN_EINVAL only if a single input image exceeds that. */
int rn_conv_wgrad(const float *dy, int ldy, const float *x, float *dw, float *colsum, int N, int Hi, int Wi,
                  int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int in_relu, void *stream);
/* nbatch independent problems of identical shape in one launch: operands and result of entry b at dy + b*dy_bstride,
 * x + b*x_bstride, dw + b*dw_bstride (floats); colsum (if given) is taken from entry colsum_batch only. */
int rn_conv_wgrad_batched(const float *dy, int ldy, const float *x, float *dw, float *colsum, int nbatch,
                          int64_t dy_bstride, int64_t x_bstride, int64_t dw_bstride, int colsum_batch, int N, int Hi, int Wi,
                          int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int in_relu,
                          const void *dy_amax, int dy_amax_n, const void *x_amax, int x_amax_n, void *stream);
/* dy_amax / x_amax (round 5): the amax of the two operands: count > 0 = amax tables (rn_conv_desc.x_amax) of that many images;
 * count -1 = ONE plain uint32 word (an fp32 bit pattern whose exponent field bounds the tensor: a Winograd-domain tensor's, whatever the
 * batch) -- with both given, RN_FP32_SPLIT3 mode runs the fp16 two-term kernels, each operand scaled by the power of two of its LARGEST
 * exponent (the reduction runs over all images); NULL: the three-term kernels. */
/* The same reduction in a FIXED order (RN_OPT_DETERMINISTIC; the host logic selects it when the option is on): every K slice
 * stores its partial result into its own slab of `workspace` (plain stores) and one ordered pass adds slabs 0, 1, 2 ... and the
 * slices' column sums into dw / colsum: bit-identical from run to run, as the reference's CPU autograd is.  Costs one write and
 * one read of slices x the result size (measured per training step: DESIGN.md 4.2). */
int64_t rn_conv_wgrad_det_workspace_bytes(int ldy, int nbatch, int64_t dw_bstride, int N, int Hi, int Wi, int Cin, int Ho, int Wo,
                                          int Cout, int kh, int kw);
int rn_conv_wgrad_batched_det(const float *dy, int ldy, const float *x, float *dw, float *colsum, int nbatch,
                              int64_t dy_bstride, int64_t x_bstride, int64_t dw_bstride, int colsum_batch, int N, int Hi, int Wi,
                              int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int in_relu,
                              const void *dy_amax, int dy_amax_n, const void *x_amax, int x_amax_n, void *workspace, int64_t workspace_bytes,
                              void *stream);

/* ---------------------------------------------------------------- bf16 convolution engine -----------------
 * The reduced-precision form of rn_conv_igemm / rn_conv_wgrad for BASELINE configs[2] (bf16 MFMA,
 * v_mfma_f32_32x32x16_bf16): activations, addend, mask and packed weights are bf16 (IEEE bfloat16, 2 bytes; NHWC /
 * [Cout][kh][kw][Cin] rows padded to a multiple of 32 ELEMENTS -- rn_pack_weights' fp32 output cast by rn_f32_to_bf16),
 * accumulation and the epilogue (scale, shift, mask, addend, activation: as rn_conv_igemm) are fp32, the result is rounded
 * once: to bf16, or stored as fp32 when y_is_f32 != 0 (head outputs that feed the loss).  Same rn_conv_desc, with strides
 * counted in elements; Cin % 8 == 0, Cout % 8 == 0 for a bf16 result (% 4 for an fp32 one); in_relu, add2 and w_batch_stride
 * are not available (RN_EINVAL).  x, w_packed, y 16-byte aligned; add, mask 16-byte aligned (8 beside an fp32 result).  Replaces the same nn.Conv2d call sites as rn_conv_igemm
 * (D/model.py:59-205, D/utils.py:12-80) when the caller opts into bf16 storage; the reference itself is fp32. */
int rn_f32_to_bf16(const float *src, void *dst, int64_t n, void *stream);   /* round-to-nearest-even, NaN stays NaN */
int rn_bf16_to_f32(const void *src, float *dst, int64_t n, void *stream);
int rn_conv_igemm_bf16(const rn_conv_desc *d, const void *x, const void *w_packed, void *y, int y_is_f32,
                       const float *scale, const float *shift, const void *add, const void *mask, void *stream);
/* Grouped form (see rn_conv_igemm_grouped): the pointers of rn_conv_group are bf16 (x, add, mask) / bf16 or fp32 (y). */
int rn_conv_igemm_bf16_grouped(const rn_conv_group *g, const void *w_packed, int y_is_f32, const float *scale,
                               const float *shift, void *stream);
/* Tile shape the grouped bf16 launcher uses for this group, as rows * 1000 + cols (128128, 256128 or 256256) --
 * tile_end[i] = running sum of ceil(N*Ho*Wo / rows) * ceil(Cout / cols).  (tile_end is ignored on input by this query;
 * d[], the pointers and n must be filled.) */
int rn_conv_igemm_bf16_tile_rows(const rn_conv_group *g, int y_is_f32);
/* ---------------------------------------------------------------- fp8 forward (BASELINE configs[4], first cut) -----------
 * The convolutions of the detector's forward pass with e4m3fn (OCP) weights and activations on v_mfma_scale_f32_32x32x64_f8f6f4
 * (all block scales 2^0; fp32 accumulation): every nn.Conv2d behind the fp32 stem with its fused batch-norm / bias / residual /
 * ReLU / sigmoid / FPN upsample-add epilogue (D/model.py:59-205, D/utils.py:12-80).  Inference only: no data or weight gradient.
 *   rn_fp8_quantize / _dequantize   fp32 <-> e4m3 with one scale per tensor: q = fp8(x * inv_scale) (saturating at +-448), x = q * scale;
 *   rn_fp8_quantize_rows            packed fp32 weight rows [rows][Kpad] -> e4m3 rows [rows][round64(Kpad)] + row_scale[rows]
 *                                   (= max|row| / 448): the per-output-channel weight scale;
 *   rn_conv_igemm_fp8               rn_conv_desc geometry as rn_conv_igemm (no mask, no input ReLU, no add2, Cin % 16 == 0):
 *                                   y = act(acc * scale[c] + shift[c] + add_q * add_scale), the caller folding
 *                                   x_scale * row_scale[c] * bn_scale[c] into scale[c]; stored as e4m3(y * out_inv_scale)
 *                                   (Cout % 16 == 0) or as fp32 (y_is_f32: head outputs). */
int rn_fp8_quantize(const float *src, void *dst, int64_t n, float inv_scale, void *stream);
/* The 3x3 / stride 2 / pad 1 max-pool of an fp32 NHWC tensor written as e4m3 with one scale (y = fp8(max * inv_scale)): max-pool followed by
 * rn_fp8_quantize, bit for bit, in one pass (the fp8 engine's stem boundary). */
int rn_maxpool_fwd_fp8out(const float *x, void *y, int N, int H, int W, int C, int Ho, int Wo, float inv_scale, void *stream);
int rn_fp8_dequantize(const void *src, float *dst, int64_t n, float scale, void *stream);
/* e4m3 -> bf16: dst[i] = bf16(q[i] * scale), n a multiple of 16, both pointers 16-byte aligned.  The fp8-forward training step (round 5:
 * BASELINE configs[4] as a TRAINING configuration) saves its activations as e4m3 and runs its data / weight gradients on the bf16
 * kernels (rn_conv_igemm_bf16, rn_conv_wgrad_bf16): this is the hand-over. */
int rn_fp8_to_bf16(const void *src, void *dst, int64_t n, float scale, void *stream);
/* bf16 -> e4m3 with one scale: dst[i] = fp8(src[i] * inv_scale), saturating at +-448; n a multiple of 16, pointers 16-byte aligned.  The
 * fp8 engine's bf16 residual stream (the last convolution of every bottleneck, the shortcuts and the FPN run in bf16 by default since
 * round 5: profiles/r05_fp8_error_budget.txt) enters the next e4m3 convolution through this. */
int rn_bf16_to_fp8(const void *src, void *dst, int64_t n, float inv_scale, void *stream);
int rn_fp8_quantize_rows(const float *w_packed, void *w_q, float *row_scale, int64_t rows, int Kpad, void *stream);
int rn_conv_igemm_fp8(const rn_conv_desc *d, const void *x_q, const void *w_q, void *y, int y_is_f32, const float *scale,
                      const float *shift, const void *add_q, float add_scale, float out_inv_scale, void *stream);
/* The pyramid levels of a head layer as one launch (rn_conv_igemm_grouped's form; the group's x / y / add are e4m3 -- y: or fp32 --
 * behind the float-typed fields, one input scale for the whole group folded into scale[c]).  tile_end[i] = running sum of
 * ceil(N*Ho*Wo / rows) * ceil(Cout / cols) with the tile rn_conv_igemm_fp8_tile_rows reports for the group (rows * 1000 + cols:
 * 128128, or 256256 where RN_OPT_FP8_P8 selects csrc/conv_fp8_p8.hip). */
int rn_conv_igemm_fp8_tile_rows(const rn_conv_group *g, int y_is_f32);
/* The tile a SINGLE launch takes for this problem (profiling and tests): rows * 1000 + cols; rn_conv_igemm_bf16_tile adds 1 000 000 when
 * the eight-wave phased kernel (csrc/conv_bf16_p8.hip) runs; for fp8, 256256 is csrc/conv_fp8_p8.hip. */
int rn_conv_igemm_fp8_tile(const rn_conv_desc *d, int y_is_f32);
int rn_conv_igemm_bf16_tile(const rn_conv_desc *d, int y_is_f32);
int rn_conv_igemm_fp8_grouped(const rn_conv_group *g, const void *w_q, int y_is_f32, const float *scale, const float *shift,
                              float add_scale, float out_inv_scale, void *stream);

/* dw[co][r][s][ci] (fp32, packed [Cout][Kpad] like rn_conv_wgrad, atomically accumulated) from bf16 dy [N,Ho,Wo,ldy>=Cout]
 * and bf16 x [N,Hi,Wi,Cin]; colsum (may be NULL) += column sums of dy.  Cin % 8 == 0, ldy % 8 == 0. */
int rn_conv_wgrad_bf16(const void *dy, int ldy, const void *x, float *dw, float *colsum, int N, int Hi, int Wi, int Cin,
                       int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, void *stream);
/* The same for n <= RN_MAX_GROUP problems that share ONE weight tensor -- the pyramid levels of a head layer (D/model.py:110-205), whose
 * weight gradient is the sum over the levels -- as one launch: dy[i] [N, Ho_i, Wo_i, ldy], x[i] [N, Hi[i], Wi[i], Cin]; dw / colsum
 * accumulate the sum.  The small levels, launches of tens of microseconds on their own, ride along with the big ones. */
int rn_conv_wgrad_bf16_grouped(int n, const void *const *dy, int ldy, const void *const *x, float *dw, float *colsum, int N,
                               const int *Hi, const int *Wi, int Cin, int Cout, int kh, int kw, int stride, int pad, void *stream);

/* Non-convolution steps of the schedule with bf16 activations (elementwise_bf16.hip): the same operations as
 * rn_maxpool_fwd / rn_maxpool_bwd / rn_upsample_add_bwd / rn_sigmoid_bwd_pad with bf16 storage on the activation side.
 * The stem stays fp32, so pooling is the boundary: forward reads the fp32 stem output and writes bf16 (+ uint8 argmax),
 * backward reads the bf16 gradient and writes the fp32 gradient of the stem output (masked by the stem's ReLU). */
int rn_maxpool_fwd_bf16out(const float *x, void *y, uint8_t *argmax, int N, int H, int W, int C, int Ho, int Wo, void *stream);
int rn_maxpool_bwd_bf16in(const float *x, const void *dy, const uint8_t *argmax, float *dx, int N, int H, int W, int C, int Ho,
                          int Wo, int relu_mask, void *stream);
int rn_upsample_add_bwd_bf16(const void *src, void *dst, int N, int Hs, int Ws, int Hd, int Wd, int C, void *stream);
int rn_sigmoid_bwd_pad_bf16(const float *dy, const float *s, void *out, int B, int64_t rows_per_image, int C, int ld,
                            int64_t src_batch_stride, void *stream);
int rn_relu_bf16(const void *src, void *dst, int64_t n, void *stream);      /* n % 4 == 0 */

/* Winograd F(4x4,3x3) stages for 3x3 / stride 1 / padding 1 convolutions (the head towers, D/model.py:120-205), fp32:
 *   rn_wino_input   x [N,H,W,C] -> V [36][Tpad][C]: B^T d B of every 6x6 patch; the problem's tiles (N * ceil(H/4) *
 *                   ceil(W/4), image-major) are written from row tile_offset on, so several problems (pyramid levels)
 *                   share one V;
 *   (GEMM)          rn_conv_igemm as a 1x1 convolution over 36 images of Tpad pixels with w_batch_stride = rows * Kpad:
 *                   M [36][Tpad][Cout] = V_p x U_p^T;
 *   rn_wino_output  M -> y [N,H,W,Cout]: A^T m A, then v = scale*v + shift; mask_mode 1: v = mask > 0 ? v : 0;
 *                   v += add; act 1: ReLU, 2: sigmoid; mask_mode 2: v = mask > 0 ? v : 0  (mask / add: dense geometry of
 *                   y, may be NULL); y_batch_stride (floats, 0 = dense) lets a head write into its slice of [B, A, n];
 *   rn_wino_weights U [36][rows][Kpad] from the OIHW parameter: mode 0 forward (rows = Cout, K = Cin), mode 1 data
 *                   gradient (rows = Cin, K = Cout, filter rotated by 180 degrees, times scale[co]); Kpad = K rounded up to 32.
 * Accuracy: ~1e-5 of the output's max magnitude (the direct kernel: ~3e-7). */
/* Grouped forms: up to RN_MAX_GROUP problems (the pyramid levels of a head layer) in one launch; their tiles are
 * concatenated in V / M / Z in table order starting at row tile_offset.  src: x or dy per problem (input transforms,
 * dy_form = 1 for A dy A^T); dst / add / mask: per problem for the output transform (add, mask may be NULL). */
typedef struct rn_wino_group {
    int n;
    int N[RN_MAX_GROUP], H[RN_MAX_GROUP], W[RN_MAX_GROUP];
    const float *src[RN_MAX_GROUP];
    float *dst[RN_MAX_GROUP];
    const float *add[RN_MAX_GROUP];
    const float *mask[RN_MAX_GROUP];       /* rn_wino_output_group: fp32, or sign-bit words with mask_mode | RN_MASK_BITS */
    void *sign[RN_MAX_GROUP];              /* rn_wino_output_group: NULL or the words that receive the result's sign bits (rn_conv_desc.sign_out) */
    void *amax[RN_MAX_GROUP];              /* RN_FP32_SPLIT3: NULL, or amax words, one per image [N] -- of src for the input transforms (read:
                                              rn_conv_desc.x_amax of the untransformed tensor), of dst for rn_wino_output_group (raised:
                                              rn_conv_desc.y_amax) */
} rn_wino_group;
/* row_amax / tensor_amax (round 5, RN_FP32_SPLIT3; NULL otherwise): the amax words of the TRANSFORMED tensor, derived from the sources'
 * per-image words by the transform's gain bound (2^7 for B^T d B, 2^8 for A dy A^T; csrc/conv_wino.hip) without a reduction:
 * row_amax[tile_offset + t] = the word of tile row t (its image's) -- the Winograd-stage GEMM's rn_conv_desc.x_amax with strides (0, 1);
 * *tensor_amax = max(*tensor_amax, the largest of them) -- the weight gradient's single word (zero it before the first launch). */
int rn_wino_input_group(const rn_wino_group *g, float *V, int C, int64_t tile_offset, int64_t Tpad, int dy_form,
                        void *row_amax, void *tensor_amax, void *stream);
/* Both input-side transforms of an output gradient in one pass over it: V = B^T dy B (rn_wino_input_group, dy_form 0: what the
 * data gradient's GEMM reads) and Z = A dy A^T (dy_form 1: what the weight gradient's reads). */
int rn_wino_input_both_group(const rn_wino_group *g, float *V, float *Z, int C, int64_t tile_offset, int64_t Tpad,
                             void *v_row_amax, void *z_tensor_amax, void *stream);
int rn_wino_output_group(const rn_wino_group *g, const float *M, int Cout, int64_t tile_offset, int64_t Tpad,
                         const float *scale, const float *shift, int mask_mode, int act, int64_t y_batch_stride, void *stream);
int rn_wino_input(const float *x, float *V, int N, int H, int W, int C, int64_t tile_offset, int64_t Tpad, void *stream);
int rn_wino_output(const float *M, float *y, int N, int H, int W, int Cout, int64_t tile_offset, int64_t Tpad,
                   const float *scale, const float *shift, const float *add, const float *mask, int mask_mode, int act,
                   int64_t y_batch_stride, void *stream);
int rn_wino_weights(const float *w, float *U, int Cout, int Cin, int mode, const float *scale, void *stream);
/* Weight gradient of the same convolutions:  dL/dg = G^T [ sum over tiles (A dy A^T) .* (B^T d B) ] G.
 *   rn_wino_dy   dy [N,H,W,C] -> Z [36][Tpad][C] (A dy A^T of every 4x4 tile; same tile order as rn_wino_input);
 *   (GEMM)       rn_conv_wgrad_batched over the 36 positions: dU[p] += Z_p^T V_p, a 1x1 problem with Tpad-strided
 *                operands and T "pixels"; colsum of batch entry 7 (position (1,1)) is the bias gradient;
 *   rn_wino_dw   dw[co][(r*3+s)*Cin + ci] += (G^T dU G)[r][s]  (the packed layout rn_conv_wgrad accumulates into,
 *                dU [36][Cout][Cin rounded up to 32]). */
int rn_wino_dy(const float *dy, float *Z, int N, int H, int W, int C, int64_t tile_offset, int64_t Tpad, void *stream);
int rn_wino_dw(const float *dU, float *dw, int Cout, int Cin, void *stream);

/* Batched per-step preparation: batch-norm folding (kind 0: bn_scale = gamma / sqrt(var + eps), bn_shift = beta - mean *
 * bn_scale, bn_rstd), weight packing (kind 1: the arguments of rn_pack_weights, rows / Kpad as it derives them) and
 * Winograd weight transforms (kind 2: the arguments of rn_wino_weights, dst [36][rows][Kpad]) and bf16 copies of packed
 * buffers (kind 3: rows * Kpad floats at src -> bf16 at dst; a launch AFTER the one that packs src), the pre-split forms of packed buffers
 * (kind 4: rn_split_weights, 8 values per thread; kind 5: rn_split_weights_f16, chunk = 4 rows, inverse row scales to bn_scale; both AFTER
 * the launch that packs src) for many layers in one launch.  jobs_dev: device array of rn_prep_job; chunks_dev: device array of nchunks (job index,
 * 256-element block inside the job) pairs.  Jobs of one launch must not depend on each other: the data-gradient packs
 * (which read bn_scale) go in a second launch. */
typedef struct rn_prep_job {
    int kind;
    int Cout, Cin, kh, kw, kw_pad, c_pad, mode, r0, nr, s0, ns, rows, Kpad;
    float eps;
    const float *src;
    float *dst;
    const float *scale;
    const float *gamma, *beta, *mean, *var;
    float *bn_scale, *bn_shift, *bn_rstd;
} rn_prep_job;
int rn_prep_batched(const rn_prep_job *jobs_dev, const int32_t *chunks_dev, int nchunks, void *stream);

/* Weight packing.  src is the reference's OIHW parameter [Cout][Cin][kh][kw] (state_dict layout).
 *   mode 0 (forward):  dst[co][r][s][ci]          = src[co][ci][r][s]
 *   mode 1 (dgrad):    dst[ci][r][s][co]          = src[co][ci][r][s] * scale[co]   (scale NULL = 1)
 *   mode 2 (dgrad, tap subset): as mode 1 but only taps r = r0 + 2*i (i < nr), s = s0 + 2*j (j < ns) are
 *                      packed, as an nr x ns filter: one output-parity class of a stride-2 data gradient
 * kw_pad >= kw and cin_pad >= Cin (forward) / cout_pad >= Cout (dgrad) give zero-filled padding of the tap and
 * channel dimensions (stem: 7x7x3 -> 7x8x4); rows are Kpad = roundup(kh*kw_pad*c_pad, 32) floats.
 * rn_unpack_wgrad converts an accumulated dw[Cout][Kpad] back to OIHW and applies the folded batch-norm:
 *   dweight[co][ci][r][s] = scale[co] * dw[co][r][s][ci]
 *   dgamma[co] = (sum_k w[co][k]*dw[co][k] - mean[co]*colsum[co]) * rstd[co],  dbeta[co] = colsum[co]
 * (any of dgamma/dbeta/scale may be NULL). */
int rn_pack_weights(const float *src, float *dst, int Cout, int Cin, int kh, int kw, int kw_pad, int c_pad,
                    int mode, const float *scale, int r0, int nr, int s0, int ns, void *stream);
int rn_unpack_wgrad(const float *dw, const float *w_packed, float *dweight, int Cout, int Cin, int kh, int kw,
                    int kw_pad, int c_pad, const float *scale, const float *mean, const float *rstd,
                    const float *colsum, float *dgamma, float *dbeta, void *stream);

/* rn_unpack_wgrad for many layers in one launch.  jobs_dev: device array of rn_unpack_job (the arguments of rn_unpack_wgrad; Kpad
 * = roundup(kh*kw_pad*c_pad, 32)); chunks_dev: device array of nchunks (job index, output channel) pairs -- every output
 * channel of every job exactly once.  The engine issues one per all-reduce bucket (or one per backward pass). */
typedef struct rn_unpack_job {
    const float *dw, *w_packed;
    float *dweight;
    int Cout, Cin, kh, kw, kw_pad, c_pad, Kpad;
    const float *scale, *mean, *rstd, *colsum;
    float *dgamma, *dbeta;
} rn_unpack_job;
int rn_unpack_batched(const rn_unpack_job *jobs_dev, const int32_t *chunks_dev, int nchunks, void *stream);

/* Frozen batch-norm folding (eval-mode BatchNorm2d, eps 1e-5, D/model.py:278-282):
 *   scale = gamma / sqrt(var + eps), shift = beta - mean*scale, rstd = 1/sqrt(var + eps). */
int rn_bn_fold(const float *gamma, const float *beta, const float *mean, const float *var, float eps, int C,
               float *scale, float *shift, float *rstd, void *stream);

/* Elementwise / data-movement kernels around the convolutions (all NHWC fp32):
 *   rn_nchw_to_nhwc4:   image [N,3,H,W] -> [N,H,W,4] (4th channel 0) for the stem's 16-byte loads
 *   rn_maxpool_fwd/bwd: MaxPool2d(3, stride 2, pad 1) (D/model.py:216); fwd also records (argmax may be NULL) the
 *                       window position 3*r+s of the FIRST maximum, bwd routes dy there (torch semantics) and
 *                       applies the stem ReLU mask (x > 0)
 *   rn_colsum:          out[c] (+)= sum over rows of g[rows, ld] (bias / beta gradients), deterministic two-pass;
 *                       accumulate != 0 adds to out (shared heads sum their five pyramid levels)
 *   rn_upsample_add_bwd: dst[n,h,w,c] += sum_{dy,dx<2} src[n,2h+dy,2w+dx,c] within src bounds (FPN top-down bwd)
 *   rn_relu_mask:       g = (z > 0) ? g : 0 in place
 *   rn_maxpool_bwd:     relu_mask 1: the gradient is also masked by x > 0 (x = the pooled tensor, a ReLU output); 2: the same with
 *                       x pointing to that tensor's SIGN BITS (rn_conv_desc.sign_out) instead of the tensor (C % 32 == 0)
 *   rn_sigmoid_bwd_pad: out[b][p][c<C] = dy[b][p][c] * s[b][p][c]*(1-s[b][p][c]) (s = sigmoid output, NULL =
 *                       identity), out[..][C..ld) = 0: a head-output gradient slice (rows_per_image rows per
 *                       image, images src_batch_stride floats apart in dy and s) copied to a dense, channel-
 *                       padded [B*rows_per_image, ld] matrix the GEMMs accept
 *   rn_add_inplace:     dst += src
 */
int rn_nchw_to_nhwc4(const float *src, float *dst, int N, int H, int W, void *amax, void *stream);
int rn_maxpool_fwd(const float *x, float *y, uint8_t *argmax, int N, int H, int W, int C, int Ho, int Wo, void *stream);
int rn_maxpool_bwd(const float *x, const float *dy, const uint8_t *argmax, float *dx, int N, int H, int W, int C,
                   int Ho, int Wo, int relu_mask, void *amax, void *stream);
int rn_colsum(const float *g, int64_t rows, int C, int ld, float *out, int accumulate, void *workspace, void *stream);
int64_t rn_colsum_workspace_bytes(int64_t rows, int C);
int rn_upsample_add_bwd(const float *src, float *dst, int N, int Hs, int Ws, int Hd, int Wd, int C, void *amax, void *stream);
int rn_relu_mask(float *g, const float *z, int64_t n, void *stream);
int rn_sigmoid_bwd_pad(const float *dy, const float *s, float *out, int B, int64_t rows_per_image, int C, int ld,
                       int64_t src_batch_stride, void *amax, void *stream);
int rn_add_inplace(float *dst, const float *src, int64_t n, int64_t per_image, void *amax, void *stream);
/* `amax` of rn_nchw_to_nhwc4 / rn_maxpool_bwd / rn_upsample_add_bwd / rn_sigmoid_bwd_pad / rn_add_inplace (round 5): NULL, or the amax
 * words of the tensor the call produces, one per image (rn_conv_desc.y_amax: raised by atomic max; the caller zeroes them first).
 * rn_add_inplace: image of element i = i / per_image. */

/* ---------------------------------------------------------------- optimizer step ---------------------------
 * clip_grad_norm_(params, max_norm) + Adam(lr, betas, eps).step() of the reference trainer
 * (train_detector_3D_angle.py:337, 385-387) over every parameter tensor in two launches, no host sync.
 *   tensor_table: device array of n_tensors entries {float *p, *g, *m, *v; int64_t n}
 *   chunk_table:  device array of n_chunks entries {int tensor; int chunk_in_tensor}, 4096 elements per chunk
 *   step: 1-based Adam step count (bias correction); max_norm <= 0 disables clipping
 *   total_norm: device float, receives the pre-clip global L2 norm; write_clipped != 0 stores g*coef back
 */
int64_t rn_opt_workspace_bytes(int n_chunks);
int rn_opt_clip_adam(const void *tensor_table, const void *chunk_table, int n_chunks, float max_norm, float lr,
                     float beta1, float beta2, float eps, int step, int write_clipped, void *workspace,
                     float *total_norm, void *stream);
/* The same step with the step number on the device: *step_dev (int32, start at 0) is incremented, then used for the bias
 * corrections.  No argument changes between steps: the launch sequence can be captured into a hipGraph and replayed. */
int rn_opt_clip_adam_dev(const void *tensor_table, const void *chunk_table, int n_chunks, float max_norm, float lr,
                         float beta1, float beta2, float eps, int *step_dev, int write_clipped, void *workspace,
                         float *total_norm, void *stream);
/* The same step with the hyper-parameters on the device as well: hp = six floats [lr, max_norm, beta1, beta2, eps, grad_scale].
 * A scheduler's new lr (ReduceLROnPlateau, train_detector_3D_angle.py:338, 412) reaches a replayed hipGraph because the caller
 * rewrites hp outside the graph.  grad_scale multiplies every gradient before the norm and the update (1/world for a
 * data-parallel gradient SUM, so no separate pass scales the gradient buffer); max_norm <= 0 disables the clip. */
int rn_opt_clip_adam_hp(const void *tensor_table, const void *chunk_table, int n_chunks, const float *hp, int *step_dev,
                        int write_clipped, void *workspace, float *total_norm, void *stream);

#ifdef __cplusplus
}
#endif
#endif
