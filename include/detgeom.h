/* detgeom.h — flat C ABI of libdetgeom.so, the MI355X (gfx950) box-geometry library.
 *
 * The reference (crapine1119/vision) is pure Python and has NO FFI / plugin boundary
 * (SURVEY.md §8(b)); its boundary is a set of Python callables.  Each entry point below
 * names the reference callable (file:line, relative to the reference repo) whose
 * arithmetic it replaces; the modules of vision_amd/ re-export those callables with the
 * reference's signatures and reaches these symbols through ctypes (vision_amd/_abi.py).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller unless marked "host";
 *  - sizes are int64_t, the last argument is the HIP stream (hipStream_t as void*);
 *  - calls are asynchronous on that stream and allocate nothing: scratch memory comes in
 *    through (ws, ws_bytes), sized by the matching dg_*_workspace_bytes();
 *  - return value: DG_OK or a negative DG_E* code; dg_last_error() gives the text
 *    (thread-local).  No exceptions and no torch types cross this boundary.
 */
#ifndef DETGEOM_H
#define DETGEOM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DG_OK 0
#define DG_EINVAL (-1)    /* bad argument (null pointer, negative size, unsupported shape) */
#define DG_EEMPTYROI (-2) /* reported through the status word of dg_roi_windows */
#define DG_ENOMEM (-3)    /* workspace too small */
#define DG_EHIP (-4)      /* a HIP runtime call / launch failed */
#define DG_ERCCL (-5)     /* RCCL could not be bound (librccl.so absent) or one of its calls failed */

#define DG_POOL_AVG 0 /* rcnn/fast_rcnn.py:354 as written (AdaptiveAvgPool2d) */
#define DG_POOL_MAX 1 /* north_star: max + argmax */

typedef void *dg_stream_t;

const char *dg_version(void);
/* copies the calling thread's last error text into buf (host); returns its length */
int dg_last_error(char *buf, size_t n);

/* ---------------------------------------------------------------- IoU
 * I1 get_iou(bb1,bb2): yolo/yolo_v1/modules.py:73-98 (eps 1e-7), copy yolo/yolo_v2/utils.py:238-263;
 * I2 Loss.get_iou: yolo/yolo_v1/modules.py:208-228 (eps 1e-6).  bb*: (n,4) xyxy, out: (n). */
int dg_iou_elementwise_f32(const float *bb1, const float *bb2, int64_t n, float eps, float *out, dg_stream_t s);
int dg_iou_elementwise_f64(const double *bb1, const double *bb2, int64_t n, double eps, double *out, dg_stream_t s);
/* N x M matrix of I1: row i is what yolo/yolo_v1/yolo.py:200 evaluates for head box i.
 * a: (n,4), b: (m,4), out: (n,m) row-major. */
int dg_iou_pairwise_f32(const float *a, int64_t n, const float *b, int64_t m, float eps, float *out, dg_stream_t s);
/* I3 get_iou(roi, grt) scalar, float64, no eps: rcnn/utils.py:50-67 (dup rcnn/fast_rcnn.py:62-79),
 * evaluated for n rois against one ground-truth box (the loop of rcnn/rcnn.py:35-40).
 * rois: (n,4) float64 xyxy, grt: (4) float64; out: (n) float64; flags: (n) int32, 1 where the
 * reference's asserts (:51-54, :65-66) would fire, else 0. */
int dg_iou_rois_vs_gt_f64(const double *rois, const double *grt, int64_t n, double *out, int32_t *flags, dg_stream_t s);

/* ---------------------------------------------------------------- NMS
 * N1 nms(out_box,out_score,threshold): yolo/yolo_v1/yolo.py:188-203, batched over images.
 * boxes: (batch,nmax,4) fp32 xyxy; scores: (batch,nmax); counts: (batch) int32 candidates per image
 * or NULL (= nmax each); a count is held to [0, nmax] on the device (it never becomes a read outside the image's boxes).  keep: (batch,nmax) int64 indices into the image's unsorted input in
 * descending-score order (ties: lower index first); keep_count: (batch) int32.
 * quirk_drop_last=1 reproduces the reference's `while numel()>1` (the candidate left alone is dropped).
 * The compare is `iou < (float)threshold` exactly as torch evaluates :201. */
size_t dg_nms_workspace_bytes(int64_t batch, int64_t nmax);
int dg_nms_batched_f32(const float *boxes, const float *scores, const int32_t *counts, int64_t batch, int64_t nmax,
                       float threshold, int quirk_drop_last, int64_t *keep, int32_t *keep_count,
                       void *ws, size_t ws_bytes, dg_stream_t s);
/* The same NMS for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  The workspace is a file-scope device array inside the library
 * (DG_NMS_NOALLOC_SCRATCH_BYTES per GPU, held until the process ends); the call allocates nothing.  Sizes whose
 * dg_nms_workspace_bytes(batch, nmax) exceeds that (12 images x 2000 boxes or one image of 9,000 boxes still fit) are refused
 * with DG_ENOMEM and a message that names both figures.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (different streams / threads) race. */
#define DG_NMS_NOALLOC_SCRATCH_BYTES (16u << 20)
size_t dg_nms_noalloc_max_bytes(void);
int dg_nms_batched_noalloc_f32(const float *boxes, const float *scores, const int32_t *counts, int64_t batch, int64_t nmax,
                               float threshold, int quirk_drop_last, int64_t *keep, int32_t *keep_count, dg_stream_t s);
/* Per-class NMS (BASELINE.json configs[1] "per-class NMS").  NO COUNTERPART IN THE REFERENCE, whose nms is class-agnostic
 * (yolo/yolo_v1/yolo.py:188-203; evaluate() splits the survivors by class afterwards, :282-298): this is that nms applied
 * to each class's boxes of an image separately (quirk and tie rule as above, IoU bits unchanged -- no coordinate
 * offsets).  labels: (batch,nmax) int64 in [0,num_classes), num_classes <= 256, batch*num_classes <= 65535.
 * keep: (batch,nmax) int64 = the classes' keep lists (image indices, descending score inside a class) concatenated in
 * class order; keep_count: (batch) totals; class_count: (batch,num_classes) int32 kept per class (may be NULL).
 * *status (device int32, zeroed by the caller): bit 0 = a label outside [0,num_classes) was met (that box is ignored). */
size_t dg_nms_per_class_workspace_bytes(int64_t batch, int64_t nmax, int64_t num_classes);
int dg_nms_batched_per_class_f32(const float *boxes, const float *scores, const int64_t *labels, const int32_t *counts,
                                 int64_t batch, int64_t nmax, int64_t num_classes, float threshold, int quirk_drop_last,
                                 int64_t *keep, int32_t *keep_count, int32_t *class_count, int32_t *status, void *ws,
                                 size_t ws_bytes, dg_stream_t s);
/* phases of dg_nms_batched_f32, exposed for measurement (bench.py) and tests; same workspace */
int dg_nms_phase_sort(const float *boxes, const float *scores, const int32_t *counts, int64_t batch, int64_t nmax,
                      void *ws, size_t ws_bytes, dg_stream_t s);
int dg_nms_phase_mask(const int32_t *counts, int64_t batch, int64_t nmax, float threshold,
                      void *ws, size_t ws_bytes, dg_stream_t s);
int dg_nms_phase_reduce(const int32_t *counts, int64_t batch, int64_t nmax, int quirk_drop_last,
                        int64_t *keep, int32_t *keep_count, void *ws, size_t ws_bytes, dg_stream_t s);

/* diagnostic (synchronous): which path image `img` of the LAST call on this workspace took.  out6 (host, 6 ints):
 * [0] 0 = pruned pair search (x-bucket band, no N x N mask), 1 = dense mask; [1] 1 = the fixed-point keep-list kernel
 * finished the image, 0 = a serial reduce did; [2] rounds of the fixed point (most list passes of one workgroup; -1: its workgroups never all arrived); [3] suppressing
 * pairs found; [4] candidate pairs per box as estimated before the search; [5] suppressing pairs per box as estimated from
 * 1024 sampled boxes.  No counterpart in the reference. */
int dg_nms_path_info(const void *ws, size_t ws_bytes, int64_t batch, int64_t nmax, int64_t img, int32_t *out6_host);

/* N2 nms(rects,scores): rcnn/fast_rcnn.py:590-610 -- no suppression there: the rects are ordered by
 * np.argsort(scores)[::-1] (:601) and element 0 is returned (:610).  out_index (device int64) = index of the maximal
 * score on the float64 scores as given; a NaN ranks above every number (argsort puts NaNs last), equal maxima resolve to
 * the highest index.  n >= 1. */
int dg_top_score_f64(const double *scores, int64_t n, int64_t *out_index, dg_stream_t s);

/* ---------------------------------------------------------------- YOLOv1 decode
 * D1 decode(output,target,conf_threshold,prob_threshold): yolo/yolo_v1/yolo.py:115-186, batched.
 * output/target: (batch,7,7,10+C) fp32 (target may be NULL).  Per image: out_box (98,4) fp32,
 * out_label (98) int64, out_conf (98), out_score (98), out_count int32 (M <= 98: slot-1 hits in
 * row-major cell order then slot-2 hits); tgt_box (4) float64, tgt_label int64 (1-based). */
int dg_yolo1_decode_f32(const float *output, const float *target, int64_t batch, int64_t num_classes,
                        float conf_threshold, float prob_threshold,
                        float *out_box, int64_t *out_label, float *out_conf, float *out_score, int32_t *out_count,
                        double *tgt_box, int64_t *tgt_label, dg_stream_t s);

/* decode -> nms as evaluate() chains them (yolo/yolo_v1/yolo.py:281-282), one launch for the whole batch: the candidates
 * of an image stay in LDS between the two.  Outputs of dg_yolo1_decode_f32 as above, plus keep (batch,98) int64 / keep_count
 * (batch) exactly as dg_nms_batched_f32 (per_class == 0: the reference's class-agnostic nms, :188-203) or
 * dg_nms_batched_per_class_f32 (per_class != 0: BASELINE configs[1], no reference counterpart; class_count
 * (batch,num_classes) int32 may be NULL) would give for (out_box, out_score[, out_label], out_count) at nms_threshold. */
int dg_yolo1_decode_nms_f32(const float *output, const float *target, int64_t batch, int64_t num_classes,
                            float conf_threshold, float prob_threshold, float nms_threshold, int per_class, int quirk_drop_last,
                            float *out_box, int64_t *out_label, float *out_conf, float *out_score, int32_t *out_count,
                            int64_t *keep, int32_t *keep_count, int32_t *class_count, double *tgt_box, int64_t *tgt_label,
                            dg_stream_t s);

/* ---------------------------------------------------------------- YOLOv2
 * get_point(output_size,bbox): yolo/yolo_v2/utils.py:228-234 (encoder, 13x13 hard-coded).
 * bbox: (n,4) xywh fp32 -> points (n,2) int64, enc (n,4) fp32 = [normed xy, wh]. */
int dg_yolo2_get_point_f32(const float *bbox, int64_t n, int64_t *points, float *enc, dg_stream_t s);
/* decode (NOT in the reference; build-defined inverse of get_point): xyxy (n,4). */
int dg_yolo2_decode_cells_f32(const int64_t *points, const float *enc, int64_t n, float *xyxy, dg_stream_t s);
/* standard anchor decode (NOT in the reference): pred (G,G,A,5+C), anchors (A,2) -> xyxy (G*G*A,4), conf. */
int dg_yolo2_decode_anchor_f32(const float *pred, int64_t G, int64_t A, int64_t C, const float *anchors,
                               float *xyxy, float *conf, dg_stream_t s);

/* ---------------------------------------------------------------- RoI pooling
 * R1 slowroipool.foward(img,roi_ary,roi_idx): rcnn/fast_rcnn.py:351-377.
 * dg_roi_windows does :366-369 and the slice normalisation of the crop at :374: rois (n,4) float64 normalised xyxy,
 * roi_idx (n) int64 -> win (n,5) int32 = {x1,y1,x2,y2,image}.  A negative bound counts from the end (start AND stop, as
 * python slices do), bounds are clamped to [0, size]; a NaN, an infinity or a product outside int64 gives bound 0 (numpy's
 * astype(int) makes INT64_MIN of them on x86-64); a negative image index counts from the last image.  *status (device
 * int32, zeroed by the caller) is OR-ed with 1 if some crop is empty (the reference raises RuntimeError there) and with 2
 * if an image index is out of range (IndexError there). */
int dg_roi_windows(const double *rois, const int64_t *roi_idx, int64_t n, int64_t B, int64_t H, int64_t W,
                   int32_t *win, int32_t *status, dg_stream_t s);
/* (B,C,H,W) -> (B,H,W,C) staging copy used by the forward kernel (lanes = channels).  nan_flag (device
 * int32, zeroed by the caller, may be NULL) is OR-ed with 1 if the map holds a NaN. */
int dg_nchw_to_nhwc_f32(const float *src, int64_t B, int64_t C, int64_t H, int64_t W, float *dst, int32_t *nan_flag,
                        dg_stream_t s);
/* forward: img_nhwc (B,H,W,C), win from dg_roi_windows; out (n,C,P,P) fp32; argmax (n,C,P,P) int32
 * = flat index y*W+x in the feature-map plane (DG_POOL_MAX only, may be NULL).  P = 7 on a map of at most 65535 cells
 * runs the tuned kernel (VGG16 conv5, the reference's call site rcnn/fast_rcnn.py:381); any other P >= 1 or map size runs
 * the general one (dg_roipool_fwd_any_f32 also takes non-square output sizes).
 * nan_flag: the word dg_nchw_to_nhwc_f32 wrote (0 = no NaN in the map: the kernel skips ATen's "NaN always
 * wins" test), or NULL = unknown (NaN-exact path). */
int dg_roipool_fwd_f32(const float *img_nhwc, int64_t B, int64_t C, int64_t H, int64_t W,
                       const int32_t *win, int64_t n, int64_t P, int mode, const int32_t *nan_flag,
                       float *out, int32_t *argmax, dg_stream_t s);
/* slowroipool(output_size) hands output_size -- an int or a (PH, PW) pair -- straight to the adaptive pool
 * (rcnn/fast_rcnn.py:352-355): out / argmax are (n,C,PH,PW). */
int dg_roipool_fwd_any_f32(const float *img_nhwc, int64_t B, int64_t C, int64_t H, int64_t W,
                           const int32_t *win, int64_t n, int64_t PH, int64_t PW, int mode, const int32_t *nan_flag,
                           float *out, int32_t *argmax, dg_stream_t s);
/* The same forward with the argmax written as uint16 (H*W <= 65535 is a precondition of the forward anyway): half the
 * index bytes to write here and to read in the backward.  The reference has no argmax tensor (autograd keeps its own
 * indices), so its width is this build's choice; the int32 entry points above stay for callers that want one. */
int dg_roipool_fwd_a16_f32(const float *img_nhwc, int64_t B, int64_t C, int64_t H, int64_t W,
                           const int32_t *win, int64_t n, int64_t P, int mode, const int32_t *nan_flag,
                           float *out, uint16_t *argmax16, dg_stream_t s);
/* R2 backward (autograd in the reference).  grad_out (n,C,P,P), argmax from the forward (max mode);
 * grad_in (B,C,H,W) is overwritten (not accumulated).  Workspace holds the per-image RoI lists. */
size_t dg_roipool_bwd_workspace_bytes(int64_t B, int64_t n);
int dg_roipool_bwd_f32(const float *grad_out, const int32_t *argmax, const int32_t *win, int64_t n,
                       int64_t B, int64_t C, int64_t H, int64_t W, int64_t P, int mode, float *grad_in,
                       void *ws, size_t ws_bytes, dg_stream_t s);
/* any output size PH x PW and any map size (7 x 7 on planes that fit the LDS accumulator -- H*W <= 32768 in max mode,
 * 14336 in avg mode -- runs the tuned kernel, everything else global float atomics into a zeroed grad_in) */
int dg_roipool_bwd_any_f32(const float *grad_out, const int32_t *argmax, const int32_t *win, int64_t n,
                           int64_t B, int64_t C, int64_t H, int64_t W, int64_t PH, int64_t PW, int mode, float *grad_in,
                           void *ws, size_t ws_bytes, dg_stream_t s);
/* backward from a uint16 argmax (dg_roipool_fwd_a16_f32; maps of at most 65535 cells); otherwise as dg_roipool_bwd_f32 */
int dg_roipool_bwd_a16_f32(const float *grad_out, const uint16_t *argmax16, const int32_t *win, int64_t n,
                           int64_t B, int64_t C, int64_t H, int64_t W, int64_t P, int mode, float *grad_in,
                           void *ws, size_t ws_bytes, dg_stream_t s);

/* ---------------------------------------------------------------- anchor k-means (1-IoU)
 * K1: the reference only fixes the sklearn-style API (clustering/cluster_prac.py:38-40); the 1-IoU
 * Lloyd step is build-defined (spec: oracle/oracle.c orc_kmeans_step / SURVEY.md §8(a) K1).
 * wh (n,2) fp32, 8-byte aligned; cent (k,2) fp32; labels (n) uint8 in/out; k <= 16; |w|, |h| < 2^31 (the fixed-point
 * sums hold x * 2^32 in int64).  acc (3k+2) int64 = {sum_w[k], sum_h[k], count[k], changed, inertia_fx}.
 *
 * dg_kmeans_iou_step: one step with the centroids as given; acc is OVERWRITTEN with this shard's sums.  ws 16-byte
 * aligned, dg_kmeans_workspace_bytes(n,k) bytes; calls that share a workspace must be stream-ordered. */
size_t dg_kmeans_workspace_bytes(int64_t n, int64_t k);
int dg_kmeans_iou_step(const float *wh, int64_t n, const float *cent, int64_t k, uint8_t *labels, int64_t *acc,
                       void *ws, size_t ws_bytes, dg_stream_t s);
/* The SHARDED Lloyd loop: one launch + one all-reduce per iteration, nothing else.  A launch adds its workgroups' rows
 * into a "partial sums" buffer `part` of dg_kmeans_part_elems(k) int64 (8 replica rows of {the acc columns, padding}:
 * workgroups sharing an XCD share a row) with 64-bit global atomics -- no workgroup waits for another.  The buffer is
 * folded, and the centroids moved, by the PROLOGUE of the next launch, after the caller has all-reduced the whole buffer
 * across ranks (integer sums: any rank count gives the same bits).
 *   part_prev: the previous iteration's buffer (all-reduced), or NULL for the first iteration (centroids as given);
 *              with it, cent is first moved (empty cluster keeps its centroid), written back, and history[slot]
 *              (may be NULL) receives the previous iteration's changed count;
 *   part_out:  this launch adds into it; it must arrive ZEROED;
 *   part_zero: a third buffer this launch zeroes for the next launch to add into (NULL: none).
 * Rotate three buffers: launch t uses (prev, out, zero) = (t-1, t, t+1) mod 3.  want_inertia == 0 leaves the inertia
 * column 0 (a Lloyd iteration does not use it).  After the last iteration all-reduce its buffer and call
 * dg_kmeans_finish: acc_out[3k+2] (may be NULL) = the folded sums, cent (may be NULL) moved by them,
 * history[slot] (may be NULL) = changed count. */
size_t dg_kmeans_part_elems(int64_t k);
int dg_kmeans_iou_iter(const float *wh, int64_t n, float *cent, int64_t k, uint8_t *labels, const int64_t *part_prev,
                       int64_t *part_out, int64_t *part_zero, int want_inertia, int64_t *history, int64_t slot,
                       dg_stream_t s);
int dg_kmeans_finish(const int64_t *part, int64_t k, float *cent, int64_t *acc_out, int64_t *history, int64_t slot,
                     dg_stream_t s);
/* `iters` iterations back to back on ONE device (no all-reduce), one launch per iteration (+ one memset before and the
 * finish after): history[slot0+t] = changed count of iteration t; acc = the sums of the LAST iteration, the only one
 * that forms the inertia sum acc[3k+1]. */
int dg_kmeans_iou_run(const float *wh, int64_t n, float *cent, int64_t k, uint8_t *labels, int64_t *acc,
                      int64_t *history, int64_t slot0, int64_t iters, void *ws, size_t ws_bytes, dg_stream_t s);
int dg_kmeans_labels_i32(const uint8_t *labels, int64_t n, int32_t *out, dg_stream_t s);
/* The PREPARED fit (NO COUNTERPART IN THE REFERENCE, like the 1-IoU k-means itself): dg_kmeans_prepare groups the boxes once
 * by a cell of a log-spaced (w, h) grid (32 cells per octave over [2^-5, 1), everything else in catch-all cells) into the
 * workspace, in pieces of at most 256 boxes with their fixed-point sums; dg_kmeans_iou_run_prepared then gives exactly
 * what dg_kmeans_iou_run gives (cent, labels, acc, history: bit for bit) but decides whole pieces at once wherever the
 * cell's bounds on the IoUs leave no doubt about the label, and walks only the other pieces box by box.  The workspace is
 * bound to the CONTENTS of wh and to the labels as they were at prepare time: prepare again when either changes behind the
 * library's back (labels written by dg_kmeans_iou_run_prepared itself are tracked).  labels == NULL says that no box has a
 * label yet (what a fit starts from: every box then counts as changed by the first iteration, exactly as an array of 255s
 * would make it) and saves moving n label bytes one by one.  dg_kmeans_iou_run_prepared with labels == NULL runs the
 * iterations over cells only: centroids and history as always, but no final pass over the boxes in their own order, so no
 * labels, no inertia, no acc (which may then be NULL as well) -- for the intermediate chunks of a fit that runs until no label
 * changes; once an iteration has changed nothing, one dg_kmeans_iou_step at the centroids as they stand gives the labels,
 * the inertia and acc that the full call would have given.  ws 256-byte aligned. */
size_t dg_kmeans_prepared_workspace_bytes(int64_t n, int64_t k);
int dg_kmeans_prepare(const float *wh, int64_t n, int64_t k, const uint8_t *labels, void *ws, size_t ws_bytes, dg_stream_t s);
int dg_kmeans_iou_run_prepared(const float *wh, int64_t n, float *cent, int64_t k, uint8_t *labels, int64_t *acc,
                               int64_t *history, int64_t slot0, int64_t iters, void *ws, size_t ws_bytes, dg_stream_t s);

/* ---------------------------------------------------------------- the exchange step of the sharded k-means
 * NO COUNTERPART IN THE REFERENCE (its KMeans is single-process sklearn, clustering/cluster_prac.py:38-40).  One process per
 * GPU; the library issues RCCL's all-reduce itself, on the caller's stream, between its own launches.  RCCL is bound at
 * run time (dlopen of librccl.so, the copy already in the process if there is one); without it these return DG_ERCCL and
 * every other entry point is unaffected.
 *  dg_comm_unique_id: rank 0 draws the 128-byte id (host memory) and hands it to the other ranks by any channel it has
 *                     (vision_amd.cluster broadcasts it through torch.distributed);
 *  dg_comm_init:      collective over the `nranks` callers, each with ITS device current; *out = the handle (host object,
 *                     freed by dg_comm_destroy);
 *  dg_comm_allreduce_i64: in-place SUM of buf[count] (device) over the ranks, asynchronous on stream s.
 * A communicator is driven by one host thread at a time (like an ncclComm_t), and every rank issues the same sequence of
 * calls on it. */
#define DG_COMM_ID_BYTES 128
typedef struct dg_comm dg_comm_t;
int dg_comm_unique_id(void *id_host);
int dg_comm_init(dg_comm_t **out, const void *id_host, int rank, int nranks);
int dg_comm_rank(const dg_comm_t *c);
int dg_comm_size(const dg_comm_t *c);
int dg_comm_allreduce_i64(dg_comm_t *c, int64_t *buf, int64_t count, dg_stream_t s);
int dg_comm_destroy(dg_comm_t *c);
/* dg_kmeans_iou_run over a SHARD of the boxes: `iters` iterations of [dg_kmeans_iou_iter; all-reduce of that launch's
 * partial-sums buffer] and the finish, all issued by this one call on stream s (no host work between iterations).  Every
 * rank calls it with its shard, the same cent / k / iters, and its own labels / acc / history / ws; results are
 * bit-identical for every rank count (integer sums).  acc = the GLOBAL sums of the last iteration.
 * use_graph != 0: the second call with the same arguments captures the chunk (launches + all-reduces) in a HIP graph
 * (thread-local capture mode), later calls replay it -- one host launch per chunk; a capture failure falls back to
 * eager issue for good (dg_last_error says why, the call still succeeds). */
int dg_kmeans_iou_run_sharded(dg_comm_t *c, const float *wh, int64_t n, float *cent, int64_t k, uint8_t *labels,
                              int64_t *acc, int64_t *history, int64_t slot0, int64_t iters, void *ws, size_t ws_bytes,
                              int use_graph, dg_stream_t s);
/* what use_graph has come to on this communicator: 0 nothing yet, 1 armed (the next call with the same arguments
 * captures), 2 replaying a captured chunk, -1 capture failed (eager from then on) */
int dg_comm_graph_state(const dg_comm_t *c);

/* ---------------------------------------------------------------- R-CNN box regression ("next" rows)
 * calcul(positive,grt): rcnn/utils.py:153-171; inverse: rcnn/rcnn.py:287-310 (variant 0) and
 * rcnn/fast_rcnn.py:543-560 (variant 1). */
int dg_rcnn_calcul_f32(const float *positive, int64_t n, const float *grt, float *t, dg_stream_t s);
int dg_rcnn_decode_f32(const int32_t *p_rect, const float *tset, int64_t n, int variant, float *g, dg_stream_t s);

/* The quota walk of do_ss ("next" row 4): rcnn/rcnn.py:35-65 (pos_thr 0.1, neg_thr 0.1, quota_neg = 2*min_num) and
 * rcnn/fast_rcnn.py:123-155 (0.5, 0.1, quota_neg = min_num).  iou / flags: what dg_iou_rois_vs_gt_f64 wrote for the image's
 * proposals, in proposal order.  sel / lab (2n int32 each): the proposals taken, in the order the reference appends them
 * (a positive before a negative of the same proposal), with `label` for positives and 0 for negatives.
 * out[0] = how many; out[1] = status: bit 0 = a visited proposal carries an assert flag (the reference raises
 * AssertionError there), bit 1 / bit 2 = the positive / negative quota was found full (the reference's fflag / bflag). */
int dg_rcnn_label_quota_f64(const double *iou, const int32_t *flags, int64_t n, int64_t min_num, int64_t quota_neg,
                            double pos_thr, double neg_thr, int32_t label, int32_t *sel, int32_t *lab, int32_t *out,
                            dg_stream_t s);

/* ---------------------------------------------------------------- YOLOv1 loss, box geometry fused ("next" row 1)
 * Loss.forward(modely, targety): yolo/yolo_v1/modules.py:234-306 -- in-loss decode :262-268, IoU :271-272,
 * responsible-slot select :275-276, terms :283-302, combination :304-305.  modely/targety: (n,7,7,10+num_classes)
 * fp32 contiguous.  quirk_slot2 != 0 keeps the reference's :264 (slot 2 takes its half-size from columns 5:7).
 * loss6[0] = loss, loss6[1..5] = the xy, wh, obj, noobj, label sums.  grad (may be NULL): d loss / d modely as the
 * reference's autograd gives it (the IoU target is not detached there), every element written. */
size_t dg_yolo1_loss_workspace_bytes(int64_t n);
int dg_yolo1_loss_f32(const float *modely, const float *targety, int64_t n, int64_t num_classes, int quirk_slot2,
                      float *loss6, float *grad, void *ws, size_t ws_bytes, dg_stream_t s);

/* YOLOv1 target encoder, batched ("next" row 4).  encode(bb,label,num_classes): yolo/yolo_v1/modules.py:151-174.
 * bb (n,4) float64 {left, top, w, h}; label (n) int64 in 1..num_classes; target (n,7,7,10+num_classes) float64, fully
 * written.  *status (caller zeroes it): bit 0 = cell index outside [-7,7) (NumPy IndexError), bit 1 = bad label. */
int dg_yolo1_encode_f64(const double *bb, const int64_t *label, int64_t n, int64_t num_classes, double *target,
                        int32_t *status, dg_stream_t s);

/* VOC-style average precision ("next" row 3).  ap(detects, targets, iou_threds): yolo/yolo_v1/yolo.py:306-367.
 * Detections: det_img (n_det) int32 image index, det_box (n_det,4) fp32 xyxy; `order` (n_det) int32 lists them by
 * class ascending and, inside a class, by confidence descending with ties in list order (the reference's stable
 * `sorted(..., reverse=True)`, :325); cls_off (n_classes+1) int64 bounds each class's run inside `order`.
 * Targets: tgt_img / tgt_cls (n_tgt) int32, tgt_box (n_tgt,4) fp32; the last row listed for a (class, image) is its
 * ground truth (:321).  ap_out (n_classes) float64 = the value the reference appends per class (NaN for a class
 * with detections but no targets, as numpy's 0/0 gives); n_grt (n_classes) int32 = len(targets[c]) (:317), which the
 * caller needs for the extra 0.0 the reference appends for an empty class (:314-315).  IoU >= iou_thr is a float32
 * comparison, as torch makes it (:341).  One workgroup per class; float64 sums in the reference's order. */
size_t dg_voc_ap_workspace_bytes(int64_t n_det, int64_t n_classes, int64_t n_images);
int dg_voc_ap_f32(const int32_t *det_img, const float *det_box, const int32_t *order, const int64_t *cls_off,
                  int64_t n_det, const int32_t *tgt_img, const int32_t *tgt_cls, const float *tgt_box, int64_t n_tgt,
                  int64_t n_classes, int64_t n_images, float iou_thr, double *ap_out, int32_t *n_grt, void *ws,
                  size_t ws_bytes, dg_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* DETGEOM_H */
