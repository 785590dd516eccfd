/* oracle.c — CPU restatement of the reference's box-geometry hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product (vision_amd/) never does.
 *
 * Plain C, scalar, single-threaded, compiled with -O2 -ffp-contract=off so every
 * float op is one IEEE-754 operation in the order the reference's torch/numpy
 * expression evaluates it.  Each function cites the reference lines it follows
 * (paths relative to the reference repo).  Pinned against the .npz files of tests/golden,
 * which were produced by executing the reference's own function bodies
 * (oracle/gen_golden.py).  The 1-IoU k-means has NO reference counterpart
 * ("parity unpinned"): its spec is stated at orc_kmeans_*.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_EINVAL (-1)
#define ORC_EEMPTYROI (-2)
#define ORC_EASSERT (-3)

/* torch.max / torch.min (elementwise, two tensors) propagate NaN. */
static inline float tmaxf(float a, float b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }
static inline float tminf(float a, float b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }
static inline double tmaxd(double a, double b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }
static inline double tmind(double a, double b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }

/* ---- I1 / I2: yolo/yolo_v1/modules.py:73-98 (eps 1e-7), :208-228 (eps 1e-6);
 *      byte-identical copy yolo/yolo_v2/utils.py:238-263 ------------------------- */
static inline float iou1_f32(const float *p, const float *q, float eps)
{
    float x_left = tmaxf(p[0], q[0]);            /* :84 */
    float y_top = tmaxf(p[1], q[1]);             /* :85 */
    float x_right = tminf(p[2], q[2]);           /* :86 */
    float y_bottom = tminf(p[3], q[3]);          /* :87 */
    float ix = x_right - x_left;                 /* :89 */
    float iy = y_bottom - y_top;                 /* :90 */
    if (ix < 0) ix = 0;                          /* :91 masked assign; NaN stays */
    if (iy < 0) iy = 0;                          /* :92 */
    float inter = ix * iy;                       /* :94 */
    float a1 = fabsf((p[2] - p[0]) * (p[3] - p[1]));   /* :95 */
    float a2 = fabsf((q[2] - q[0]) * (q[3] - q[1]));   /* :96 */
    return inter / (((a1 + a2) - inter) + eps);  /* :97, left-to-right */
}

void orc_iou_elementwise_f32(const float *bb1, const float *bb2, int64_t n, float eps, float *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = iou1_f32(bb1 + 4 * i, bb2 + 4 * i, eps);
}

/* same expression when the caller passes float64 tensors (dtype follows the input) */
void orc_iou_elementwise_f64(const double *bb1, const double *bb2, int64_t n, double eps, double *out)
{
    for (int64_t i = 0; i < n; ++i) {
        const double *p = bb1 + 4 * i, *q = bb2 + 4 * i;
        double ix = tmind(p[2], q[2]) - tmaxd(p[0], q[0]);
        double iy = tmind(p[3], q[3]) - tmaxd(p[1], q[1]);
        if (ix < 0) ix = 0;
        if (iy < 0) iy = 0;
        double inter = ix * iy;
        double a1 = fabs((p[2] - p[0]) * (p[3] - p[1]));
        double a2 = fabs((q[2] - q[0]) * (q[3] - q[1]));
        out[i] = inter / (((a1 + a2) - inter) + eps);
    }
}

/* N x M matrix of I1 (never materialised by the reference; row i is what
 * yolo/yolo_v1/yolo.py:200 computes with expand_as for head box i). */
void orc_iou_pairwise_f32(const float *a, int64_t n, const float *b, int64_t m, float eps, float *out)
{
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < m; ++j) out[i * m + j] = iou1_f32(a + 4 * i, b + 4 * j, eps);
}

/* ---- I3: rcnn/utils.py:50-67 (dup rcnn/fast_rcnn.py:62-79), float64, asserts ----- */
int orc_iou_scalar_f64(const double *bb1, const double *bb2, double *out)
{
    if (!(bb1[0] < bb1[2]) || !(bb1[1] < bb1[3]) || !(bb2[0] < bb2[2]) || !(bb2[1] < bb2[3]))
        return ORC_EASSERT;                                    /* :51-54 */
    double x_left = bb1[0] > bb2[0] ? bb1[0] : bb2[0];         /* python max */
    double y_top = bb1[1] > bb2[1] ? bb1[1] : bb2[1];
    double x_right = bb1[2] < bb2[2] ? bb1[2] : bb2[2];
    double y_bottom = bb1[3] < bb2[3] ? bb1[3] : bb2[3];
    if (x_right < x_left || y_bottom < y_top) { *out = 0.0; return ORC_OK; }  /* :59-60 */
    double inter = (x_right - x_left) * (y_bottom - y_top);    /* :61 */
    double a1 = (bb1[2] - bb1[0]) * (bb1[3] - bb1[1]);
    double a2 = (bb2[2] - bb2[0]) * (bb2[3] - bb2[1]);
    double iou = inter / ((a1 + a2) - inter);                  /* :64, no eps */
    if (!(iou >= 0.0) || !(iou <= 1.0)) return ORC_EASSERT;    /* :65-66 */
    *out = iou;
    return ORC_OK;
}

/* ---- N1: yolo/yolo_v1/yolo.py:188-203 -------------------------------------------
 * Literal restatement: sort ids by score descending, then
 *   while numel(ids) > 1: pop head -> keep; ids = ids[iou(head, ids) < thr]
 * The candidate left alone at the end is NOT kept (the numel()>1 quirk).
 * torch casts the python threshold to fp32 for the compare (verified in
 * gen_golden.py's probe), and NaN < thr is false (suppressed).
 * Sort: descending score, ties by ascending index (the reference sort is
 * unstable, so only tie-free inputs are comparable). */
typedef struct { float s; int64_t i; } orc_si;
static int cmp_desc(const void *a, const void *b)
{
    const orc_si *x = a, *y = b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

int orc_nms_f32(const float *boxes, const float *scores, int64_t n, double threshold,
                int quirk_drop_last, int64_t *keep, int64_t *n_keep)
{
    *n_keep = 0;
    if (n <= 0) return ORC_OK;                                  /* :189,203 */
    float thr = (float)threshold;
    orc_si *ord = malloc(sizeof(orc_si) * (size_t)n);
    int64_t *ids = malloc(sizeof(int64_t) * (size_t)n);
    if (!ord || !ids) { free(ord); free(ids); return ORC_EINVAL; }
    for (int64_t i = 0; i < n; ++i) { ord[i].s = scores[i]; ord[i].i = i; }
    qsort(ord, (size_t)n, sizeof(orc_si), cmp_desc);            /* :194 */
    for (int64_t i = 0; i < n; ++i) ids[i] = ord[i].i;
    int64_t cnt = n, k = 0;
    int64_t *cur = ids;
    while (cnt > (quirk_drop_last ? 1 : 0)) {                   /* :196 */
        int64_t head = cur[0];                                  /* :197 */
        keep[k++] = head;                                       /* :198 */
        cur += 1; cnt -= 1;                                     /* :199 */
        int64_t w = 0;
        for (int64_t j = 0; j < cnt; ++j) {                     /* :200-201 */
            float iou = iou1_f32(boxes + 4 * head, boxes + 4 * cur[j], 1e-7f);
            if (iou < thr) cur[w++] = cur[j];
        }
        cnt = w;
    }
    *n_keep = k;
    free(ord); free(ids);
    return ORC_OK;
}

/* ---- D1: yolo/yolo_v1/yolo.py:115-186, one image ----------------------------------
 * output/target: (7,7,10+C) fp32.  Boxes are computed in float64 (fp32 tensor +
 * float64 numpy grid promotes, :137) and rounded to fp32 once (:180).
 * Output order: slot-1 hits in row-major cell order, then slot-2 hits (:164-178).
 * Returns the number of candidates M (0 => the reference returns empty lists). */
int64_t orc_yolo1_decode(const float *output, const float *target, int64_t C,
                         double conf_threshold, double prob_threshold,
                         float *out_box, int64_t *out_label, float *out_conf, float *out_score,
                         double *bbox_target, int64_t *label_target)
{
    const int S = 7;
    const int64_t N = 10 + C;
    const float ct = (float)conf_threshold, pt = (float)prob_threshold;
    int64_t m = 0;
    for (int slot = 0; slot < 2; ++slot) {
        for (int cy = 0; cy < S; ++cy)
            for (int cx = 0; cx < S; ++cx) {
                const float *o = output + ((int64_t)cy * S + cx) * N;
                /* :124 torch.max over the class channels: value and index of the max.
                 * (first maximal index; NaN would win — not exercised) */
                float cs = o[10]; int64_t cl = 0;
                for (int64_t c = 1; c < C; ++c) if (o[10 + c] > cs) { cs = o[10 + c]; cl = c; }
                float conf = o[4 + 5 * slot];                     /* :125 */
                float prob = conf * cs;                           /* :126 fp32 */
                if (!((prob >= pt) && (cs > ct))) continue;       /* :161-162 */
                const float *bx = o + 5 * slot;
                /* :129-131 lt_norm = cell * 1 / 7 in float64 (numpy int*1/7) */
                double ltx = (double)(cx * 1) / 7.0, lty = (double)(cy * 1) / 7.0;
                /* :137  boxes * 1 / 7 : fp32 tensor ops, then + float64 */
                double x = (double)((bx[0] * 1.0f) / 7.0f) + ltx;
                double y = (double)((bx[1] * 1.0f) / 7.0f) + lty;
                double hw = (double)(bx[2] / 2.0f), hh = (double)(bx[3] / 2.0f);   /* :144 wh/2 in fp32 */
                out_box[4 * m + 0] = (float)(x - hw);             /* :144, :180 */
                out_box[4 * m + 1] = (float)(y - hh);
                out_box[4 * m + 2] = (float)(x + hw);             /* :145 */
                out_box[4 * m + 3] = (float)(y + hh);
                out_label[m] = cl; out_conf[m] = conf; out_score[m] = cs;
                ++m;
            }
    }
    if (target) {
        /* :121 point_ind = target.sum(-1) != 0 ; the reference assumes one such cell */
        for (int cy = 0; cy < S; ++cy)
            for (int cx = 0; cx < S; ++cx) {
                const float *t = target + ((int64_t)cy * S + cx) * N;
                float sum = 0.f;
                for (int64_t c = 0; c < N; ++c) sum += t[c];
                if (sum == 0.f) continue;
                /* :134 argsort(target[10:])[-1] + 1 : index of the largest class entry, 1-based */
                int64_t best = 0;
                for (int64_t c = 1; c < C; ++c) if (t[10 + c] >= t[10 + best]) best = c;
                *label_target = best + 1;
                double ltx = (double)(cx * 1) / 7.0, lty = (double)(cy * 1) / 7.0;
                double x = (double)((t[0] * 1.0f) / 7.0f) + ltx;  /* :140 */
                double y = (double)((t[1] * 1.0f) / 7.0f) + lty;
                double hw = (double)(t[2] / 2.0f), hh = (double)(t[3] / 2.0f);     /* :146 */
                bbox_target[0] = x - hw; bbox_target[1] = y - hh;
                bbox_target[2] = x + hw; bbox_target[3] = y + hh;  /* :152 float64 */
                return m;
            }
    }
    return m;
}

/* ---- R1: rcnn/fast_rcnn.py:351-377 --------------------------------------------------
 * roi window: x1=floor(x1*W) x2=ceil(x2*W) y1=floor(y1*H) y2=ceil(y2*H) in float64 (:366-369)
 * then torch adaptive pooling of the crop to PxP: bin i over length L is
 * [floor(i*L/P), ceil((i+1)*L/P)) (ATen, not in the reference tree).
 * mode 0 = AdaptiveAvgPool2d (as written, :354), mode 1 = AdaptiveMaxPool2d with
 * argmax (north_star).  ATen max: first maximum in row-major scan, NaN wins.
 * argmax is the flat index y*W+x in the feature-map plane (build-defined). */
static inline int bin_start(int i, int L, int P) { return (i * L) / P; }
static inline int bin_end(int i, int L, int P) { return ((i + 1) * L + P - 1) / P; }

/* One bound of a python slice, as rcnn/fast_rcnn.py:366-374 makes it: np.floor / np.ceil of coordinate x size, `.astype(int)`
 * (int64: a NaN, an infinity or a value outside int64 becomes INT64_MIN on x86-64, with a warning), then used as a slice
 * bound on a dimension of `size` cells: a negative bound counts from the end, and the result is clamped to [0, size].  BOTH
 * bounds wrap (a stop of -1 is the last column), not only the start.  No conversion of an out-of-range double (undefined in
 * C): everything outside [-size, size] is decided in double. */
static int py_slice_bound(double v, int size)
{
    if (!(v >= -9223372036854775808.0 && v < 9223372036854775808.0)) return 0;   /* INT64_MIN + size < 0 -> 0 */
    if (v < -(double)size) return 0;
    if (v > (double)size) return size;
    int i = (int)v;                                 /* |v| <= size: exact (v is an integer: floor / ceil made it) */
    return i < 0 ? i + size : i;
}
int orc_roi_window(const double *roi, int H, int W, int *x1, int *y1, int *x2, int *y2)
{
    *x1 = py_slice_bound(floor(roi[0] * (double)W), W);         /* :366 */
    *x2 = py_slice_bound(ceil(roi[2] * (double)W), W);          /* :367 */
    *y1 = py_slice_bound(floor(roi[1] * (double)H), H);         /* :368 */
    *y2 = py_slice_bound(ceil(roi[3] * (double)H), H);          /* :369 */
    if (*x2 <= *x1 || *y2 <= *y1) return ORC_EEMPTYROI;         /* an empty crop makes the adaptive pool raise */
    return ORC_OK;
}

/* output_size may be an int or a (PH, PW) pair (rcnn/fast_rcnn.py:352-355 hands it to the adaptive pool) */
int orc_roipool_fwd2_f32(const float *img, int64_t B, int64_t C, int H, int W,
                         const double *rois, const int64_t *roi_idx, int64_t n, int PH, int PW, int mode,
                         float *out, int32_t *argmax)
{
    for (int64_t r = 0; r < n; ++r) {
        int x1, y1, x2, y2;
        int st = orc_roi_window(rois + 4 * r, H, W, &x1, &y1, &x2, &y2);
        if (st) return st;
        int64_t b = roi_idx[r];
        if (b < 0) b += B;                          /* :374 img[roi_idx[i], ...]: a negative index counts from the end */
        if (b < 0 || b >= B) return ORC_EINVAL;     /* (torch raises IndexError) */
        int Lh = y2 - y1, Lw = x2 - x1;
        for (int64_t c = 0; c < C; ++c) {
            const float *plane = img + (b * C + c) * (int64_t)H * W;
            for (int ph = 0; ph < PH; ++ph) {
                int hs = y1 + bin_start(ph, Lh, PH), he = y1 + bin_end(ph, Lh, PH);
                for (int pw = 0; pw < PW; ++pw) {
                    int ws = x1 + bin_start(pw, Lw, PW), we = x1 + bin_end(pw, Lw, PW);
                    int64_t o = ((r * C + c) * PH + ph) * PW + pw;
                    if (mode == 1) {
                        float best = -INFINITY; int32_t bi = hs * W + ws;
                        for (int y = hs; y < he; ++y)
                            for (int x = ws; x < we; ++x) {
                                float v = plane[y * W + x];
                                if ((v > best) || (v != v)) { best = v; bi = y * W + x; }
                            }
                        out[o] = best;
                        if (argmax) argmax[o] = bi;
                    } else {
                        float sum = 0.f;
                        for (int y = hs; y < he; ++y)
                            for (int x = ws; x < we; ++x) sum += plane[y * W + x];
                        out[o] = sum / (float)(he - hs) / (float)(we - ws);
                    }
                }
            }
        }
    }
    return ORC_OK;
}

int orc_roipool_fwd_f32(const float *img, int64_t B, int64_t C, int H, int W,
                        const double *rois, const int64_t *roi_idx, int64_t n, int P, int mode,
                        float *out, int32_t *argmax)
{
    return orc_roipool_fwd2_f32(img, B, C, H, W, rois, roi_idx, n, P, P, mode, out, argmax);
}

/* ---- R2: backward of R1 (autograd in the reference; no explicit code).
 * max: grad_in[b,c,argmax] += grad_out ; avg: grad_out / kh / kw spread over the bin.
 * Sequential accumulation in RoI order (fp32). grad_in must be zeroed by the caller. */
int orc_roipool_bwd2_f32(const float *grad_out, const int32_t *argmax, int64_t B, int64_t C, int H, int W,
                         const double *rois, const int64_t *roi_idx, int64_t n, int PH, int PW, int mode,
                         float *grad_in)
{
    /* the caller sized (and zeroed) grad_in */
    for (int64_t r = 0; r < n; ++r) {
        int x1, y1, x2, y2;
        int st = orc_roi_window(rois + 4 * r, H, W, &x1, &y1, &x2, &y2);
        if (st) return st;
        int64_t b = roi_idx[r];
        if (b < 0) b += B;                          /* as the forward */
        if (b < 0 || b >= B) return ORC_EINVAL;
        int Lh = y2 - y1, Lw = x2 - x1;
        for (int64_t c = 0; c < C; ++c) {
            float *plane = grad_in + (b * C + c) * (int64_t)H * W;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    int64_t o = ((r * C + c) * PH + ph) * PW + pw;
                    if (mode == 1) {
                        plane[argmax[o]] += grad_out[o];
                    } else {
                        int hs = y1 + bin_start(ph, Lh, PH), he = y1 + bin_end(ph, Lh, PH);
                        int ws = x1 + bin_start(pw, Lw, PW), we = x1 + bin_end(pw, Lw, PW);
                        float g = grad_out[o] / (float)(he - hs) / (float)(we - ws);
                        for (int y = hs; y < he; ++y)
                            for (int x = ws; x < we; ++x) plane[y * W + x] += g;
                    }
                }
        }
    }
    return ORC_OK;
}

int orc_roipool_bwd_f32(const float *grad_out, const int32_t *argmax, int64_t B, int64_t C, int H, int W,
                        const double *rois, const int64_t *roi_idx, int64_t n, int P, int mode,
                        float *grad_in)
{
    return orc_roipool_bwd2_f32(grad_out, argmax, B, C, H, W, rois, roi_idx, n, P, P, mode, grad_in);
}

/* ---- D3: yolo/yolo_v2/utils.py:228-234 get_point (encoder) and its inverse ---------
 * center = xy + wh/2 ; points = (center // (1/13)).long() ; normed = (center % (1/13)) / (1/13)
 * all in fp32 torch arithmetic; 1/13 is a python double cast to fp32 by torch.
 * torch floor_divide / remainder on floats: remainder(a,b) = a - b*floor(a/b);
 * floor_divide uses the fmod-based "div_floor" form. */
static float t_remainder(float a, float b)
{
    float m = fmodf(a, b);
    if (m != 0.f && ((b < 0.f) != (m < 0.f))) m += b;
    return m;
}
static float t_floordiv(float a, float b)
{
    float m = fmodf(a, b);
    float d = (a - m) / b;
    if (m != 0.f && ((b < 0.f) != (m < 0.f))) d -= 1.f;
    float f;
    if (d != 0.f) { f = floorf(d); if (d - f > 0.5f) f += 1.f; }
    else f = copysignf(0.f, a / b);
    return f;
}
void orc_yolo2_get_point(const float *bbox, int64_t n, int64_t *points, float *enc)
{
    const float cell = (float)(1.0 / 13.0);
    for (int64_t i = 0; i < n; ++i) {
        const float *b = bbox + 4 * i;
        for (int d = 0; d < 2; ++d) {
            float center = b[d] + b[2 + d] / 2.0f;                 /* :229 */
            points[2 * i + d] = (int64_t)t_floordiv(center, cell); /* :231 */
            float residual = t_remainder(center, cell);            /* :232 */
            enc[4 * i + d] = residual / cell;                      /* :233 */
            enc[4 * i + 2 + d] = b[2 + d];                         /* :234 */
        }
    }
}
/* Inverse (build-defined; the reference has no YOLOv2 decode):
 * center = (points + normed)/13 ; xyxy = center -/+ wh/2, fp32. */
void orc_yolo2_decode_cells(const int64_t *points, const float *enc, int64_t n, float *xyxy)
{
    for (int64_t i = 0; i < n; ++i)
        for (int d = 0; d < 2; ++d) {
            float center = ((float)points[2 * i + d] + enc[4 * i + d]) / 13.0f;
            float half = enc[4 * i + 2 + d] / 2.0f;
            xyxy[4 * i + d] = center - half;
            xyxy[4 * i + 2 + d] = center + half;
        }
}
/* Standard anchor form (Redmon & Farhadi 2016, eq. in §2 "Direct location prediction"):
 * bx = (sigmoid(tx)+cx)/G, by likewise, bw = pw*exp(tw), bh = ph*exp(th); conf = sigmoid(to).
 * pred: (G,G,A,5+C) fp32, anchors (A,2) in grid-normalised units (fraction of image).
 * Parity unpinned (no reference code); computed in double, rounded once. */
void orc_yolo2_decode_anchor(const float *pred, int64_t G, int64_t A, int64_t C, const float *anchors,
                             float *xyxy, float *conf)
{
    int64_t N = 5 + C;
    for (int64_t gy = 0; gy < G; ++gy)
        for (int64_t gx = 0; gx < G; ++gx)
            for (int64_t a = 0; a < A; ++a) {
                int64_t k = (gy * G + gx) * A + a;
                const float *p = pred + k * N;
                double sx = 1.0 / (1.0 + exp(-(double)p[0])), sy = 1.0 / (1.0 + exp(-(double)p[1]));
                double bx = (sx + (double)gx) / (double)G, by = (sy + (double)gy) / (double)G;
                double bw = (double)anchors[2 * a] * exp((double)p[2]);
                double bh = (double)anchors[2 * a + 1] * exp((double)p[3]);
                xyxy[4 * k + 0] = (float)(bx - bw / 2); xyxy[4 * k + 1] = (float)(by - bh / 2);
                xyxy[4 * k + 2] = (float)(bx + bw / 2); xyxy[4 * k + 3] = (float)(by + bh / 2);
                conf[k] = (float)(1.0 / (1.0 + exp(-(double)p[4])));
            }
}

/* ---- next(2): rcnn/utils.py:153-171 calcul, and its inverse rcnn/rcnn.py:287-310 -- */
void orc_rcnn_calcul_f32(const float *positive, int64_t n, const float *grt, float *t)
{
    float g_w = grt[2] - grt[0], g_h = grt[3] - grt[1];
    float g_x = grt[0] + g_w / 2, g_y = grt[1] + g_h / 2;
    for (int64_t i = 0; i < n; ++i) {
        const float *p = positive + 4 * i;
        float p_w = p[2] - p[0], p_h = p[3] - p[1];
        float p_x = p[0] + p_w / 2, p_y = p[1] + p_h / 2;
        t[4 * i + 0] = (g_x - p_x) / p_w;
        t[4 * i + 1] = (g_y - p_y) / p_h;
        t[4 * i + 2] = logf(g_w / p_w);
        t[4 * i + 3] = logf(g_h / p_h);
    }
}
/* variant 0: rcnn/rcnn.py:292-310 (px = xmin + pw/2, int rects: pw/2 is true division -> fp32)
 * variant 1: rcnn/fast_rcnn.py:543-560 (px = (xmax + xmin)/2) */
void orc_rcnn_decode_f32(const int32_t *p_rect, const float *tset, int64_t n, int variant, float *g)
{
    for (int64_t i = 0; i < n; ++i) {
        const int32_t *p = p_rect + 4 * i; const float *t = tset + 4 * i;
        float pw = (float)(p[2] - p[0]), ph = (float)(p[3] - p[1]);
        float px, py;
        if (variant == 0) { px = (float)p[0] + pw / 2; py = (float)p[1] + ph / 2; }
        else { px = (float)(p[2] + p[0]) / 2; py = (float)(p[3] + p[1]) / 2; }
        float gx = pw * t[0] + px, gy = ph * t[1] + py;
        float gw = pw * expf(t[2]), gh = ph * expf(t[3]);
        g[4 * i + 0] = gx - gw / 2; g[4 * i + 1] = gy - gh / 2;
        g[4 * i + 2] = gx + gw / 2; g[4 * i + 3] = gy + gh / 2;
    }
}

/* ---- L1: YOLOv1 loss, box geometry fused.  yolo/yolo_v1/modules.py:234-306.
 * y, t: (n,7,7,10+C) fp32.  Object cells: t[...,4] > 0 (:243); background: == 0 (:244).
 *   in-loss decode :262-268   lt = cell/7 in float64 (:204-206), x1 = fp32(lt + x*(1/7) - w/2), x2 = w + x1;
 *                             slot 2 takes its half-size from cols 5:7 (:264, kept when quirk != 0)
 *   IoU :271-272 (module-level get_iou, eps 1e-7), responsible slot: 1 if iou1 >= iou2 else 2 (:275-276)
 *   terms :283-302, loss = (5*(xy+wh) + obj + .5*noobj + label)/n (:304-305)
 * comp[5] = {xy, wh, obj, noobj, label} sums in double; *loss = the combination, rounded to fp32.
 * grad (nullable): d loss / d y as autograd gives it -- the IoU target is NOT detached in the reference, so
 * the responsible slot's box also receives d(best IoU); min/max ties split the gradient evenly (ATen). */
static inline float sgnf(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }
static void l1_box(double lt, float c, float half_src, float ext, float *lo, float *hi)
{
    float a = c * (float)(1.0 / 7.0);
    float hw = half_src / 2.0f;
    *lo = (float)((lt + (double)a) - (double)hw);
    *hi = ext + *lo;
}
/* IoU of pred box b against target u, and d iou / d b */
static float l1_iou_grad(const float b[4], const float u[4], float g[4])
{
    float xl = fmaxf(b[0], u[0]), yt = fmaxf(b[1], u[1]), xr = fminf(b[2], u[2]), yb = fminf(b[3], u[3]);
    float ixr = xr - xl, iyr = yb - yt;
    float ix = ixr < 0.f ? 0.f : ixr, iy = iyr < 0.f ? 0.f : iyr;
    float I = ix * iy;
    float w = b[2] - b[0], h = b[3] - b[1];
    float A1 = fabsf(w * h), A2 = fabsf((u[2] - u[0]) * (u[3] - u[1]));
    float U = ((A1 + A2) - I) + 1e-7f;
    float iou = I / U;
    float gI = 1.0f / U + I / (U * U);
    float gA = -I / (U * U);
    float gix = (ixr < 0.f) ? 0.f : gI * iy, giy = (iyr < 0.f) ? 0.f : gI * ix;
    float gwh = gA * sgnf(w * h);
    float gw = gwh * h, gh = gwh * w;
    /* xr = min(b2,u2): to b2 fully if smaller, half on a tie; xl = max(b0,u0) likewise */
    float fx2 = (b[2] < u[2]) ? 1.f : ((b[2] == u[2]) ? 0.5f : 0.f), fx1 = (b[0] > u[0]) ? 1.f : ((b[0] == u[0]) ? 0.5f : 0.f);
    float fy2 = (b[3] < u[3]) ? 1.f : ((b[3] == u[3]) ? 0.5f : 0.f), fy1 = (b[1] > u[1]) ? 1.f : ((b[1] == u[1]) ? 0.5f : 0.f);
    g[0] = -gix * fx1 - gw; g[2] = gix * fx2 + gw;
    g[1] = -giy * fy1 - gh; g[3] = giy * fy2 + gh;
    return iou;
}
int orc_yolo1_loss(const float *y, const float *t, int64_t n, int C, int quirk, double *comp, float *loss, float *grad)
{
    const int D = 10 + C;
    double Lxy = 0, Lwh = 0, Lobj = 0, Lno = 0, Llab = 0;
    const float gN = 1.0f / (float)n;
    if (grad) memset(grad, 0, sizeof(float) * (size_t)n * 49 * D);
    for (int64_t b = 0; b < n; ++b)
        for (int i = 0; i < 7; ++i)
            for (int j = 0; j < 7; ++j) {
                const int64_t o = ((b * 7 + i) * 7 + j) * D;
                const float *p = y + o, *q = t + o;
                float *gp = grad ? grad + o : NULL;
                if (q[4] > 0.f) {
                    const double ltx = (double)j * (1.0 / 7.0), lty = (double)i * (1.0 / 7.0);
                    float b1[4], b2[4], u[4];
                    l1_box(ltx, p[0], p[2], p[2], &b1[0], &b1[2]);
                    l1_box(lty, p[1], p[3], p[3], &b1[1], &b1[3]);
                    l1_box(ltx, p[5], quirk ? p[5] : p[7], p[7], &b2[0], &b2[2]);
                    l1_box(lty, p[6], quirk ? p[6] : p[8], p[8], &b2[1], &b2[3]);
                    l1_box(ltx, q[0], q[2], q[2], &u[0], &u[2]);
                    l1_box(lty, q[1], q[3], q[3], &u[1], &u[3]);
                    float g1[4], g2[4];
                    float iou1 = l1_iou_grad(b1, u, g1), iou2 = l1_iou_grad(b2, u, g2);
                    const int r = (iou1 >= iou2) ? 0 : 1;
                    const float best = r ? iou2 : iou1;
                    const float *pr = p + 5 * r;
                    float dx = pr[0] - q[0], dy = pr[1] - q[1];
                    Lxy += (double)(dx * dx) + (double)(dy * dy);
                    float sw = sgnf(pr[2]) * sqrtf(fabsf(pr[2])), sh = sgnf(pr[3]) * sqrtf(fabsf(pr[3]));
                    float dw = sw - sqrtf(q[2]), dh = sh - sqrtf(q[3]);
                    Lwh += (double)(dw * dw) + (double)(dh * dh);
                    float dob = best - pr[4];
                    Lobj += (double)(dob * dob);
                    for (int c = 10; c < D; ++c) { float d = p[c] - q[c]; Llab += (double)(d * d); }
                    if (gp) {
                        float *gr = gp + 5 * r;
                        gr[0] += 10.f * dx * gN; gr[1] += 10.f * dy * gN;
                        gr[2] += 10.f * dw * gN * (sgnf(pr[2]) * (0.5f / sqrtf(fabsf(pr[2]))) * sgnf(pr[2]));
                        gr[3] += 10.f * dh * gN * (sgnf(pr[3]) * (0.5f / sqrtf(fabsf(pr[3]))) * sgnf(pr[3]));
                        gr[4] += -2.f * dob * gN;
                        const float gb = 2.f * dob * gN;            /* d loss / d best iou */
                        const float *g = r ? g2 : g1;
                        const float inv7 = (float)(1.0 / 7.0);
                        for (int a = 0; a < 2; ++a) {               /* a = 0: x, 1: y */
                            float glo = gb * g[a], ghi = gb * g[a + 2], gt = glo + ghi;
                            if (r == 1 && quirk) { gr[a] += gt * inv7 - gt * 0.5f; gr[a + 2] += ghi; }
                            else { gr[a] += gt * inv7; gr[a + 2] += ghi - gt * 0.5f; }
                        }
                        for (int c = 10; c < D; ++c) gp[c] += 2.f * (p[c] - q[c]) * gN;
                    }
                } else if (q[4] == 0.f) {
                    float d4 = p[4] - q[4], d9 = p[9] - q[9];
                    Lno += (double)(d4 * d4) + (double)(d9 * d9);
                    if (gp) { gp[4] += d4 * gN; gp[9] += d9 * gN; }
                }
            }
    if (comp) { comp[0] = Lxy; comp[1] = Lwh; comp[2] = Lobj; comp[3] = Lno; comp[4] = Llab; }
    if (loss) *loss = (float)((5.0 * (Lxy + Lwh) + Lobj + 0.5 * Lno + Llab) / (double)n);
    return ORC_OK;
}

/* ---- K1: 1-IoU anchor k-means.  NOT IN THE REFERENCE (clustering/cluster_prac.py:38-40
 * only fixes the sklearn-style API).  Spec, parity unpinned:
 *   d(b,c) = 1 - inter/(w*h + cw*ch - inter), inter = min(w,cw)*min(h,ch), fp32, I1 op order, no eps
 *   assign = argmin_k d, ties -> lowest index
 *   update = (sum_fx / count) with sum_fx = sum llrint(w * 2^32) as int64; centroid =
 *            (float)((double)sum_fx / (double)count / 2^32); empty cluster keeps its centroid
 *   inertia_fx = sum llrint(d * 2^32)  (int64)
 * One Lloyd step: labels (in/out, uint8), acc[3k+2] = {sumw[k], sumh[k], cnt[k], changed, inertia_fx}
 * (acc is ADDED to, so shards can accumulate into one buffer). */
static inline float kdist(float w, float h, float cw, float ch)
{
    float inter = fminf(w, cw) * fminf(h, ch);
    float uni = ((w * h) + (cw * ch)) - inter;
    return 1.0f - inter / uni;
}
void orc_kmeans_step(const float *wh, int64_t n, const float *cent, int k, uint8_t *labels, int64_t *acc)
{
    for (int64_t i = 0; i < n; ++i) {
        float w = wh[2 * i], h = wh[2 * i + 1];
        int best = 0; float bd = kdist(w, h, cent[0], cent[1]);
        for (int c = 1; c < k; ++c) {
            float d = kdist(w, h, cent[2 * c], cent[2 * c + 1]);
            if (d < bd) { bd = d; best = c; }
        }
        acc[best] += llrint((double)w * 4294967296.0);
        acc[k + best] += llrint((double)h * 4294967296.0);
        acc[2 * k + best] += 1;
        acc[3 * k] += (labels[i] != (uint8_t)best);
        acc[3 * k + 1] += llrint((double)bd * 4294967296.0);
        labels[i] = (uint8_t)best;
    }
}
void orc_kmeans_update(const int64_t *acc, int k, float *cent)
{
    for (int c = 0; c < k; ++c) {
        int64_t cnt = acc[2 * k + c];
        if (cnt > 0) {
            cent[2 * c] = (float)((double)acc[c] / (double)cnt / 4294967296.0);
            cent[2 * c + 1] = (float)((double)acc[k + c] / (double)cnt / 4294967296.0);
        }
    }
}
