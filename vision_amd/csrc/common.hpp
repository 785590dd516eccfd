// common.hpp — shared host/device helpers for libdetgeom (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/detgeom.h"

typedef unsigned long long u64;
typedef unsigned int u32;

// ---- host-side error plumbing (thread-local text, negative status codes) ----------
void dg_set_error(const char *fmt, ...);

#define DG_REQUIRE(cond, ...)                                                                     \
    do {                                                                                          \
        if (!(cond)) { dg_set_error(__VA_ARGS__); return DG_EINVAL; }                             \
    } while (0)

#define DG_CHECK_LAUNCH(name)                                                                     \
    do {                                                                                          \
        hipError_t e_ = hipGetLastError();                                                        \
        if (e_ != hipSuccess) { dg_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); return DG_EHIP; } \
    } while (0)

// communicator handle of include/detgeom.h (comm.hip owns it; kmeans.hip keeps the captured chunk of its sharded loop here)
struct dg_comm {
    void *comm = nullptr;            // ncclComm_t
    int rank = 0, nranks = 1, device = 0;
    void *graph_exec = nullptr;      // hipGraphExec_t of the last captured dg_kmeans_iou_run_sharded chunk
    uint64_t graph_key[12] = {};     // the arguments it was captured with
    int graph_state = 0;             // 0 none, 1 key seen once (ran eagerly), 2 captured, -1 capture failed: stay eager
};

static inline int64_t dg_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t dg_round_up(int64_t a, int64_t b) { return dg_cdiv(a, b) * b; }

// ---- device: the IoU of SURVEY §8(a) I1 in the reference's operation order ---------
// yolo/yolo_v1/modules.py:84-97.  fmaxf/fminf drop a NaN operand where torch.max keeps it,
// but any NaN coordinate already makes that box's area (and so the denominator) NaN, so the
// quotient is NaN either way; compiled with -ffp-contract=off and IEEE division.
__device__ __forceinline__ float dg_area(float x1, float y1, float x2, float y2)
{
    return fabsf((x2 - x1) * (y2 - y1));
}
__device__ __forceinline__ float dg_iou(float ax1, float ay1, float ax2, float ay2, float aarea,
                                        float bx1, float by1, float bx2, float by2, float barea, float eps)
{
    float ix = fminf(ax2, bx2) - fmaxf(ax1, bx1);
    float iy = fminf(ay2, by2) - fmaxf(ay1, by1);
    ix = fmaxf(ix, 0.0f);
    iy = fmaxf(iy, 0.0f);
    float inter = ix * iy;
    return inter / (((aarea + barea) - inter) + eps);
}

// ---- device/host: pieces of the greedy NMS shared by nms.hip and the fused YOLOv1 decode + NMS kernel (decode.hip) -----
// Order of yolo/yolo_v1/yolo.py:194 `sort(0, descending=True)`: descending score <=> ascending key; NaN sorts first (as
// torch); ties -> lower index first (build-defined).  dg_score_desc() is the 32-bit score part of the key.
__device__ __forceinline__ u32 dg_score_desc(float score)
{
    u32 u = __float_as_uint(score);
    if (score != score) u = 0x7fffffffu;            // canonical +NaN: above +inf
    u32 ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~ord;
}
__device__ __forceinline__ u64 make_key(float score, u32 idx) { return ((u64)dg_score_desc(score) << 32) | idx; }

// !(inter/den < thr) without the division (yolo.py:201 keeps iou < thr): with p the float below thr and mid = (p+thr)/2, a
// correctly rounded quotient is < thr exactly when the real quotient is below mid (or equal to it when the tie rounds down
// to p, i.e. thr's mantissa is odd).  mid has 25 significant bits and den 24: mid*den is exact in double, so is the compare.
template <bool TIE_EVEN>
__device__ __forceinline__ bool nms_suppressed(float inter, float den, double mid)
{
    const double prod = mid * (double)den, di = (double)inter;
    return TIE_EVEN ? !(di < prod) : !(di <= prod);
}
// host: *mid for a threshold; returns TIE_EVEN (thr's mantissa even: a quotient exactly on mid rounds up to thr)
static inline bool dg_nms_threshold_mid(float threshold, double *mid)
{
    if (threshold != threshold) *mid = (double)threshold;
    else *mid = ((double)__builtin_nextafterf(threshold, -__builtin_inff()) + (double)threshold) * 0.5;
    uint32_t tbits;
    __builtin_memcpy(&tbits, &threshold, 4);
    return (tbits & 1u) == 0;
}
