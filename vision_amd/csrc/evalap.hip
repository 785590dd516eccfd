// evalap.hip — VOC-style average precision (yolo/yolo_v1/yolo.py:306-367) on the device (gfx950).
// The reference walks the detections of a class in descending-confidence order, takes the IoU (I1, eps 1e-7) against
// the one ground truth its image holds for that class (:321 keeps the last one listed), and counts the first
// detection per image that reaches the threshold as TP, everything else as FP (:341-348); then cumsum, recall /
// precision, the precision envelope and the area sum (:349-364), all in float64.
// Here: one workgroup per class.  "First per image" is an atomicMin of the sorted rank; the cumulative TP count is a
// block scan; precision tp/(rank+1) and the recall step k/n - (k-1)/n are the same float64 divisions numpy does; the
// envelope is a reverse max scan (exact); only TP ranks contribute a non-zero area term, and those are summed by one
// lane in rank order, so the float64 sum has the reference's rounding.  The IoU comparison is done in float32 like
// torch's tensor-vs-Python-scalar comparison (:341).
#include "common.hpp"
#include <math.h>

#define AP_BLOCK 1024
#define AP_WAVES (AP_BLOCK / 64)

// gt_idx[c*I + img] = index of the last target of class c listed for that image (-1: none); n_grt[c] counts all rows
__global__ __launch_bounds__(256) void k_ap_gt_table(const int32_t *__restrict__ timg, const int32_t *__restrict__ tcls, int64_t T,
                                                     int64_t C, int64_t I, int32_t *gt_idx, int32_t *n_grt)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int c = tcls[t], img = timg[t];
    if ((unsigned)c >= (unsigned)C) return;
    atomicAdd(&n_grt[c], 1);
    if ((unsigned)img < (unsigned)I) atomicMax(&gt_idx[(int64_t)c * I + img], (int)t);
}

__device__ __forceinline__ int ap_scan_incl(int v, int *lds, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < AP_WAVES; ++w) {
        const int x = lds[w];
        before += (w < wave) ? x : 0;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return v + before;
}

// inclusive max over this thread and all higher-numbered threads of the workgroup
__device__ __forceinline__ double ap_rscan_max(double v, double *lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_down(v, o, 64);
        if (lane + o < 64) v = fmax(v, t);
    }
    if (lane == 0) lds[wave] = v;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < AP_WAVES; ++w)
        if (w > wave) v = fmax(v, lds[w]);
    __syncthreads();
    return v;
}

__global__ __launch_bounds__(AP_BLOCK) void k_voc_ap(const int32_t *__restrict__ det_img, const float4 *__restrict__ det_box,
                                                     const int32_t *__restrict__ order, const int64_t *__restrict__ cls_off,
                                                     const float4 *__restrict__ tgt_box, const int32_t *__restrict__ gt_idx,
                                                     const int32_t *__restrict__ n_grt, int64_t I, float thr, unsigned *first,
                                                     int32_t *tpc, double *terms, double *__restrict__ ap_out)
{
    __shared__ int s_i[AP_WAVES];
    __shared__ double s_d[AP_WAVES];
    __shared__ double s_term[AP_BLOCK];
    __shared__ double s_carry;
    const int c = blockIdx.x, tid = threadIdx.x;
    const int64_t off = cls_off[c];
    const int D = (int)(cls_off[c + 1] - off);
    const int n = n_grt[c];
    const int64_t row = (int64_t)c * I;

    // pass 0: which detections reach the threshold, and the first such rank per image
    for (int r = tid; r < D; r += AP_BLOCK) {
        const int d = order[off + r];
        const int img = det_img[d];
        int q = 0;
        if ((unsigned)img < (unsigned)I) {
            const int g = gt_idx[row + img];
            if (g >= 0) {
                const float4 p = det_box[d], t = tgt_box[g];
                const float iou = dg_iou(p.x, p.y, p.z, p.w, dg_area(p.x, p.y, p.z, p.w), t.x, t.y, t.z, t.w,
                                         dg_area(t.x, t.y, t.z, t.w), 1e-7f);
                if (iou >= thr) {
                    q = 1;
                    atomicMin(&first[row + img], (unsigned)r);
                }
            }
        }
        tpc[off + r] = q;
    }
    __syncthreads();

    // pass 1: TP flags and their running count (bit 31 = this rank is a TP)
    int carry = 0;
    for (int base = 0; base < D; base += AP_BLOCK) {
        const int r = base + tid;
        int tp = 0;
        if (r < D && tpc[off + r]) {
            const int img = det_img[order[off + r]];
            tp = __hip_atomic_load(&first[row + img], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)r;
        }
        int total;
        const int incl = ap_scan_incl(tp, s_i, total);
        if (r < D) tpc[off + r] = (carry + incl) | (tp << 31);
        carry += total;
    }
    const int K = carry;
    __syncthreads();

    // pass 2: precision envelope from the back; every TP rank leaves its area term at terms[k-1]
    double later = 0.0;                                     // the reference appends precision 0.0 (:357)
    for (int base = ((D - 1) / AP_BLOCK) * AP_BLOCK; base >= 0 && D > 0; base -= AP_BLOCK) {
        const int r = base + tid;
        int v = 0, k = 0;
        double p = 0.0;
        if (r < D) {
            v = tpc[off + r];
            k = v & 0x7fffffff;
            p = (double)k / (double)(r + 1);                // tp_sum / (tp_sum + fp_sum)  (:353)
        }
        const double env = fmax(ap_rscan_max(p, s_d), later);
        if (r < D && v < 0) terms[off + k - 1] = ((double)k / (double)n - (double)(k - 1) / (double)n) * env;
        if (tid == 0) s_carry = env;
        __syncthreads();
        later = s_carry;
        __syncthreads();
    }

    // pass 3: the area sum in rank order (zero terms of the reference's loop leave a float64 sum unchanged)
    double ap = 0.0;
    for (int base = 0; base < K; base += AP_BLOCK) {
        if (base + tid < K) s_term[tid] = terms[off + base + tid];
        __syncthreads();
        if (tid == 0) {
            const int m = (K - base) < AP_BLOCK ? (K - base) : AP_BLOCK;
            for (int j = 0; j < m; ++j) ap += s_term[j];
        }
        __syncthreads();
    }
    if (tid == 0) ap_out[c] = (n == 0 && D > 0) ? (double)NAN : ap;   // 0/0 recall (:352) poisons the sum
}

static inline size_t ap_align(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t dg_voc_ap_workspace_bytes(int64_t n_det, int64_t n_classes, int64_t n_images)
{
    if (n_det < 0 || n_classes <= 0 || n_images < 0) return 0;
    const size_t table = ap_align((size_t)n_classes * (size_t)n_images * 4);
    return 2 * table + ap_align((size_t)n_det * 4) + ap_align((size_t)n_det * 8) + 256;
}

extern "C" int dg_voc_ap_f32(const int32_t *det_img, const float *det_box, const int32_t *order, const int64_t *cls_off,
                             int64_t n_det, const int32_t *tgt_img, const int32_t *tgt_cls, const float *tgt_box, int64_t n_tgt,
                             int64_t n_classes, int64_t n_images, float iou_thr, double *ap_out, int32_t *n_grt, void *ws,
                             size_t ws_bytes, dg_stream_t s)
{
    DG_REQUIRE(n_det >= 0 && n_tgt >= 0 && n_classes >= 1 && n_images >= 0, "voc_ap: negative size");
    DG_REQUIRE(n_det < (1ll << 31) && n_classes * n_images < (1ll << 40), "voc_ap: too large");
    DG_REQUIRE(cls_off && ap_out && n_grt && ws, "voc_ap: null pointer");
    DG_REQUIRE(n_det == 0 || (det_img && det_box && order), "voc_ap: null pointer");
    DG_REQUIRE(n_tgt == 0 || (tgt_img && tgt_cls && tgt_box), "voc_ap: null pointer");
    DG_REQUIRE((((uintptr_t)det_box) & 15) == 0 && (((uintptr_t)tgt_box) & 15) == 0 && (((uintptr_t)ws) & 255) == 0,
               "voc_ap: misaligned pointer");
    if (ws_bytes < dg_voc_ap_workspace_bytes(n_det, n_classes, n_images)) { dg_set_error("voc_ap: workspace too small"); return DG_ENOMEM; }
    hipStream_t st = (hipStream_t)s;
    const size_t table = ap_align((size_t)n_classes * (size_t)n_images * 4);
    char *w = (char *)ws;
    int32_t *gt_idx = (int32_t *)w;
    unsigned *first = (unsigned *)(w + table);
    int32_t *tpc = (int32_t *)(w + 2 * table);
    double *terms = (double *)(w + 2 * table + ap_align((size_t)n_det * 4));
    // gt_idx = -1 and first = 0xffffffff in one fill; the per-class counts start at zero
    if (hipMemsetAsync(w, 0xff, 2 * table, st) != hipSuccess || hipMemsetAsync(n_grt, 0, (size_t)n_classes * 4, st) != hipSuccess) {
        dg_set_error("voc_ap: memset failed");
        return DG_EHIP;
    }
    if (n_tgt > 0)
        hipLaunchKernelGGL(k_ap_gt_table, dim3((unsigned)dg_cdiv(n_tgt, 256)), dim3(256), 0, st, tgt_img, tgt_cls, n_tgt, n_classes,
                           n_images, gt_idx, n_grt);
    hipLaunchKernelGGL(k_voc_ap, dim3((unsigned)n_classes), dim3(AP_BLOCK), 0, st, det_img, (const float4 *)det_box, order, cls_off,
                       (const float4 *)tgt_box, gt_idx, n_grt, n_images, iou_thr, first, tpc, terms, ap_out);
    DG_CHECK_LAUNCH("k_voc_ap");
    return DG_OK;
}
