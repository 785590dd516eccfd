// decode.hip — YOLOv1 decode (batched), YOLOv2 cell encode/decode, R-CNN box regression (gfx950).
#include "common.hpp"

// ---------------------------------------------------------------- D1: yolo/yolo_v1/yolo.py:115-186
// One wave per image: the (7,7,10+C) prediction is staged in LDS with coalesced loads, lane = cell.
// Box corners are computed in float64 and rounded once, as the reference's fp32-tensor + float64-grid
// arithmetic does (:129-131,:137,:144-145,:180).  Ordered compaction (slot-1 hits in row-major cell
// order, then slot-2 hits, :164-178) uses wave ballots + popcounts.
#define Y1_S 7
#define Y1_CELLS 49
#define Y1_MAXCH 96

__device__ __forceinline__ void y1_box(const float *o, int cx, int cy, double *b)
{
    double ltx = (double)(cx * 1) / 7.0, lty = (double)(cy * 1) / 7.0;
    double x = (double)((o[0] * 1.0f) / 7.0f) + ltx;
    double y = (double)((o[1] * 1.0f) / 7.0f) + lty;
    double hw = (double)(o[2] / 2.0f), hh = (double)(o[3] / 2.0f);
    b[0] = x - hw; b[1] = y - hh; b[2] = x + hw; b[3] = y + hh;
}

__global__ __launch_bounds__(64) void k_yolo1_decode(const float *__restrict__ output, const float *__restrict__ target,
                                                     int C, float ct, float pt,
                                                     float *__restrict__ out_box, int64_t *__restrict__ out_label,
                                                     float *__restrict__ out_conf, float *__restrict__ out_score,
                                                     int32_t *__restrict__ out_count,
                                                     double *__restrict__ tgt_box, int64_t *__restrict__ tgt_label)
{
    __shared__ float so[Y1_CELLS * Y1_MAXCH];
    const int img = blockIdx.x, lane = threadIdx.x;
    const int N = 10 + C, total = Y1_CELLS * N;
    const float *g = output + (int64_t)img * total;
    for (int t = lane; t < total; t += 64) so[t] = g[t];
    __syncthreads();
    const bool active = lane < Y1_CELLS;
    const int cell = active ? lane : 0;
    const int cy = cell / Y1_S, cx = cell % Y1_S;
    const float *o = so + cell * N;
    float cs = o[10];
    int cl = 0;
    for (int c = 1; c < C; ++c) {          // torch.max over classes: first maximal index (:124)
        float v = o[10 + c];
        if (v > cs) { cs = v; cl = c; }
    }
    const float conf1 = o[4], conf2 = o[9];
    const bool p1 = active && (conf1 * cs >= pt) && (cs > ct);   // :161
    const bool p2 = active && (conf2 * cs >= pt) && (cs > ct);   // :162
    const u64 b1 = __ballot(p1), b2 = __ballot(p2);
    const u64 lt = (1ull << lane) - 1ull;
    const int c1 = __builtin_popcountll(b1);
    const int64_t ob = (int64_t)img * 2 * Y1_CELLS;
    if (p1 || p2) {
        double bx[4];
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {
            if (slot == 0 ? p1 : p2) {
                int pos = slot == 0 ? __builtin_popcountll(b1 & lt) : c1 + __builtin_popcountll(b2 & lt);
                y1_box(o + 5 * slot, cx, cy, bx);
                float4 f = make_float4((float)bx[0], (float)bx[1], (float)bx[2], (float)bx[3]);
                reinterpret_cast<float4 *>(out_box)[ob + pos] = f;
                out_label[ob + pos] = cl;
                out_conf[ob + pos] = slot == 0 ? conf1 : conf2;
                out_score[ob + pos] = cs;
            }
        }
    }
    if (lane == 0) out_count[img] = c1 + __builtin_popcountll(b2);

    if (target) {                                    // :121,:133-134,:140-152
        __syncthreads();
        const float *tg = target + (int64_t)img * total;
        for (int t = lane; t < total; t += 64) so[t] = tg[t];
        __syncthreads();
        float sum = 0.f;
        for (int c = 0; c < N; ++c) sum += o[c];
        const u64 objm = __ballot(active && sum != 0.f);
        if (objm == 0) {
            if (lane == 0) {
                tgt_label[img] = 0;
                for (int k = 0; k < 4; ++k) tgt_box[4 * img + k] = __longlong_as_double(0x7ff8000000000000LL);
            }
        } else if (lane == __builtin_ctzll(objm)) {
            int best = 0;
            for (int c = 1; c < C; ++c) if (o[10 + c] >= o[10 + best]) best = c;
            tgt_label[img] = best + 1;
            double bx[4];
            y1_box(o, cx, cy, bx);
            for (int k = 0; k < 4; ++k) tgt_box[4 * img + k] = bx[k];
        }
    }
}

extern "C" int dg_yolo1_decode_f32(const float *output, const float *target, int64_t batch, int64_t num_classes,
                                   float conf_threshold, float prob_threshold,
                                   float *out_box, int64_t *out_label, float *out_conf, float *out_score, int32_t *out_count,
                                   double *tgt_box, int64_t *tgt_label, dg_stream_t s)
{
    DG_REQUIRE(batch >= 0, "yolo1_decode: batch < 0");
    if (batch == 0) return DG_OK;
    DG_REQUIRE(num_classes >= 1 && 10 + num_classes <= Y1_MAXCH, "yolo1_decode: num_classes must be in [1,%d]", Y1_MAXCH - 10);
    DG_REQUIRE(output && out_box && out_label && out_conf && out_score && out_count, "yolo1_decode: null pointer");
    DG_REQUIRE(!target || (tgt_box && tgt_label), "yolo1_decode: target given without tgt outputs");
    DG_REQUIRE((((uintptr_t)out_box) & 15) == 0, "yolo1_decode: out_box must be 16-byte aligned");
    hipLaunchKernelGGL(k_yolo1_decode, dim3((unsigned)batch), dim3(64), 0, (hipStream_t)s, output, target, (int)num_classes,
                       conf_threshold, prob_threshold, out_box, out_label, out_conf, out_score, out_count, tgt_box, tgt_label);
    DG_CHECK_LAUNCH("k_yolo1_decode");
    return DG_OK;
}

// ---------------------------------------------------------------- YOLOv2: yolo/yolo_v2/utils.py:228-234
// torch's float floor_divide / remainder (fmod based), restated for fp32.
__device__ __forceinline__ float t_remainder(float a, float b)
{
    float m = fmodf(a, b);
    if (m != 0.f && ((b < 0.f) != (m < 0.f))) m += b;
    return m;
}
__device__ __forceinline__ float t_floordiv(float a, float b)
{
    float m = fmodf(a, b);
    float d = (a - m) / b;
    if (m != 0.f && ((b < 0.f) != (m < 0.f))) d -= 1.f;
    float f;
    if (d != 0.f) { f = floorf(d); if (d - f > 0.5f) f += 1.f; }
    else f = copysignf(0.f, a / b);
    return f;
}

__global__ __launch_bounds__(256) void k_yolo2_get_point(const float4 *__restrict__ bbox, int64_t n,
                                                         int64_t *__restrict__ points, float4 *__restrict__ enc)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float cell = (float)(1.0 / 13.0);
    float4 b = bbox[i];
    float cx = b.x + b.z / 2.0f, cy = b.y + b.w / 2.0f;
    points[2 * i] = (int64_t)t_floordiv(cx, cell);
    points[2 * i + 1] = (int64_t)t_floordiv(cy, cell);
    enc[i] = make_float4(t_remainder(cx, cell) / cell, t_remainder(cy, cell) / cell, b.z, b.w);
}

__global__ __launch_bounds__(256) void k_yolo2_decode_cells(const int64_t *__restrict__ points, const float4 *__restrict__ enc,
                                                            int64_t n, float4 *__restrict__ xyxy)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 e = enc[i];
    float cx = ((float)points[2 * i] + e.x) / 13.0f, cy = ((float)points[2 * i + 1] + e.y) / 13.0f;
    float hw = e.z / 2.0f, hh = e.w / 2.0f;
    xyxy[i] = make_float4(cx - hw, cy - hh, cx + hw, cy + hh);
}

__global__ __launch_bounds__(256) void k_yolo2_decode_anchor(const float *__restrict__ pred, int64_t G, int64_t A, int64_t C,
                                                             const float *__restrict__ anchors, float4 *__restrict__ xyxy,
                                                             float *__restrict__ conf)
{
    int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= G * G * A) return;
    const int64_t a = k % A, gx = (k / A) % G, gy = k / (A * G);
    const float *p = pred + k * (5 + C);
    double sx = 1.0 / (1.0 + exp(-(double)p[0])), sy = 1.0 / (1.0 + exp(-(double)p[1]));
    double bx = (sx + (double)gx) / (double)G, by = (sy + (double)gy) / (double)G;
    double bw = (double)anchors[2 * a] * exp((double)p[2]), bh = (double)anchors[2 * a + 1] * exp((double)p[3]);
    xyxy[k] = make_float4((float)(bx - bw / 2), (float)(by - bh / 2), (float)(bx + bw / 2), (float)(by + bh / 2));
    conf[k] = (float)(1.0 / (1.0 + exp(-(double)p[4])));
}

extern "C" int dg_yolo2_get_point_f32(const float *bbox, int64_t n, int64_t *points, float *enc, dg_stream_t s)
{
    DG_REQUIRE(n >= 0, "yolo2_get_point: n < 0");
    if (n == 0) return DG_OK;
    DG_REQUIRE(bbox && points && enc, "yolo2_get_point: null pointer");
    DG_REQUIRE((((uintptr_t)bbox | (uintptr_t)enc) & 15) == 0, "yolo2_get_point: 16-byte alignment required");
    hipLaunchKernelGGL(k_yolo2_get_point, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, (const float4 *)bbox, n,
                       points, (float4 *)enc);
    DG_CHECK_LAUNCH("k_yolo2_get_point");
    return DG_OK;
}

extern "C" int dg_yolo2_decode_cells_f32(const int64_t *points, const float *enc, int64_t n, float *xyxy, dg_stream_t s)
{
    DG_REQUIRE(n >= 0, "yolo2_decode_cells: n < 0");
    if (n == 0) return DG_OK;
    DG_REQUIRE(points && enc && xyxy, "yolo2_decode_cells: null pointer");
    DG_REQUIRE((((uintptr_t)xyxy | (uintptr_t)enc) & 15) == 0, "yolo2_decode_cells: 16-byte alignment required");
    hipLaunchKernelGGL(k_yolo2_decode_cells, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, points,
                       (const float4 *)enc, n, (float4 *)xyxy);
    DG_CHECK_LAUNCH("k_yolo2_decode_cells");
    return DG_OK;
}

extern "C" int dg_yolo2_decode_anchor_f32(const float *pred, int64_t G, int64_t A, int64_t C, const float *anchors,
                                          float *xyxy, float *conf, dg_stream_t s)
{
    DG_REQUIRE(G > 0 && A > 0 && C >= 0, "yolo2_decode_anchor: bad shape");
    DG_REQUIRE(pred && anchors && xyxy && conf, "yolo2_decode_anchor: null pointer");
    DG_REQUIRE((((uintptr_t)xyxy) & 15) == 0, "yolo2_decode_anchor: 16-byte alignment required");
    hipLaunchKernelGGL(k_yolo2_decode_anchor, dim3((unsigned)dg_cdiv(G * G * A, 256)), dim3(256), 0, (hipStream_t)s, pred, G, A, C,
                       anchors, (float4 *)xyxy, conf);
    DG_CHECK_LAUNCH("k_yolo2_decode_anchor");
    return DG_OK;
}

// ---------------------------------------------------------------- R-CNN box regression
// calcul: rcnn/utils.py:153-171; inverse: rcnn/rcnn.py:287-310 (variant 0), rcnn/fast_rcnn.py:543-560 (variant 1)
__global__ __launch_bounds__(256) void k_rcnn_calcul(const float4 *__restrict__ pos, int64_t n, const float *__restrict__ grt,
                                                     float4 *__restrict__ t)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float g_w = grt[2] - grt[0], g_h = grt[3] - grt[1];
    float g_x = grt[0] + g_w / 2, g_y = grt[1] + g_h / 2;
    float4 p = pos[i];
    float p_w = p.z - p.x, p_h = p.w - p.y;
    float p_x = p.x + p_w / 2, p_y = p.y + p_h / 2;
    t[i] = make_float4((g_x - p_x) / p_w, (g_y - p_y) / p_h, logf(g_w / p_w), logf(g_h / p_h));
}

__global__ __launch_bounds__(256) void k_rcnn_decode(const int4 *__restrict__ p_rect, const float4 *__restrict__ tset, int64_t n,
                                                     int variant, float4 *__restrict__ g)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int4 p = p_rect[i];
    float4 t = tset[i];
    float pw = (float)(p.z - p.x), ph = (float)(p.w - p.y);
    float px, py;
    if (variant == 0) { px = (float)p.x + pw / 2; py = (float)p.y + ph / 2; }
    else { px = (float)(p.z + p.x) / 2; py = (float)(p.w + p.y) / 2; }
    float gx = pw * t.x + px, gy = ph * t.y + py;
    float gw = pw * expf(t.z), gh = ph * expf(t.w);
    g[i] = make_float4(gx - gw / 2, gy - gh / 2, gx + gw / 2, gy + gh / 2);
}

extern "C" int dg_rcnn_calcul_f32(const float *positive, int64_t n, const float *grt, float *t, dg_stream_t s)
{
    DG_REQUIRE(n >= 0, "rcnn_calcul: n < 0");
    if (n == 0) return DG_OK;
    DG_REQUIRE(positive && grt && t, "rcnn_calcul: null pointer");
    DG_REQUIRE((((uintptr_t)positive | (uintptr_t)t) & 15) == 0, "rcnn_calcul: 16-byte alignment required");
    hipLaunchKernelGGL(k_rcnn_calcul, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, (const float4 *)positive, n, grt,
                       (float4 *)t);
    DG_CHECK_LAUNCH("k_rcnn_calcul");
    return DG_OK;
}

extern "C" int dg_rcnn_decode_f32(const int32_t *p_rect, const float *tset, int64_t n, int variant, float *g, dg_stream_t s)
{
    DG_REQUIRE(n >= 0, "rcnn_decode: n < 0");
    if (n == 0) return DG_OK;
    DG_REQUIRE(p_rect && tset && g, "rcnn_decode: null pointer");
    DG_REQUIRE((((uintptr_t)p_rect | (uintptr_t)tset | (uintptr_t)g) & 15) == 0, "rcnn_decode: 16-byte alignment required");
    hipLaunchKernelGGL(k_rcnn_decode, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, (const int4 *)p_rect,
                       (const float4 *)tset, n, variant, (float4 *)g);
    DG_CHECK_LAUNCH("k_rcnn_decode");
    return DG_OK;
}

// ---------------------------------------------------------------- YOLOv1 target encoder ("next" row 4)
// encode(bb,label,num_classes): yolo/yolo_v1/modules.py:151-174, float64 like NumPy.  bb = {left, top, w, h} relative
// to the image; the object's cell gets [x_cell, y_cell, w, h, 1] in both box slots and a one-hot class.
// One thread per box writes its cell (the target was zeroed by the caller's memset).  status: bit 0 = a cell index
// outside [-7,7) (NumPy raises IndexError), bit 1 = label outside 1..C.
__global__ __launch_bounds__(256) void k_yolo1_encode(const double *__restrict__ bb, const int64_t *__restrict__ label, int64_t n,
                                                      int C, double *__restrict__ target, int32_t *__restrict__ status)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    const int D = 10 + C;
    const double cell = 1.0 / 7.0;
    const double w = bb[4 * b + 2], h = bb[4 * b + 3];
    const double cx = bb[4 * b] + w / 2, cy = bb[4 * b + 1] + h / 2;
    int i = (int)(cx * 7), j = (int)(cy * 7);              // astype(int): truncation
    const double xc = (cx - (double)i * cell) / cell, yc = (cy - (double)j * cell) / cell;
    const int64_t lab = label[b];
    int bad = 0;
    if (i < -7 || i >= 7 || j < -7 || j >= 7) bad |= 1;
    if (lab < 1 || lab > C) bad |= 2;
    if (bad) { atomicOr(status, bad); return; }
    if (i < 0) i += 7;                                       // NumPy negative index
    if (j < 0) j += 7;
    double *t = target + ((b * 7 + j) * 7 + i) * D;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) { t[5 * s2] = xc; t[5 * s2 + 1] = yc; t[5 * s2 + 2] = w; t[5 * s2 + 3] = h; t[5 * s2 + 4] = 1.0; }
    t[10 + lab - 1] = 1.0;
}

extern "C" int dg_yolo1_encode_f64(const double *bb, const int64_t *label, int64_t n, int64_t num_classes, double *target,
                                   int32_t *status, dg_stream_t s)
{
    DG_REQUIRE(n >= 0 && num_classes >= 1 && num_classes <= 4096, "yolo1_encode: bad shape");
    if (n == 0) return DG_OK;
    DG_REQUIRE(bb && label && target && status, "yolo1_encode: null pointer");
    hipStream_t st = (hipStream_t)s;
    if (hipMemsetAsync(target, 0, (size_t)n * 49 * (10 + num_classes) * sizeof(double), st) != hipSuccess) {
        dg_set_error("yolo1_encode: memset failed");
        return DG_EHIP;
    }
    hipLaunchKernelGGL(k_yolo1_encode, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, st, bb, label, n, (int)num_classes, target, status);
    DG_CHECK_LAUNCH("k_yolo1_encode");
    return DG_OK;
}
