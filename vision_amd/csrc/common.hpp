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
