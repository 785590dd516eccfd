// iou.hip — elementwise / pairwise / rois-vs-GT IoU kernels (gfx950).
// Reference arithmetic: yolo/yolo_v1/modules.py:73-98, :208-228; rcnn/utils.py:50-67.
#include "common.hpp"

// ---------------------------------------------------------------- elementwise (I1/I2)
// One pair per lane; each box is one 16-byte load (dwordx4), the result one dword store.
__global__ __launch_bounds__(256) void k_iou_elementwise_f32(const float4 *__restrict__ a, const float4 *__restrict__ b,
                                                             int64_t n, float eps, float *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float4 p = a[i], q = b[i];
        out[i] = dg_iou(p.x, p.y, p.z, p.w, dg_area(p.x, p.y, p.z, p.w),
                        q.x, q.y, q.z, q.w, dg_area(q.x, q.y, q.z, q.w), eps);
    }
}

__global__ __launch_bounds__(256) void k_iou_elementwise_f64(const double *__restrict__ a, const double *__restrict__ b,
                                                             int64_t n, double eps, double *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const double *p = a + 4 * i, *q = b + 4 * i;
        double ix = fmin(p[2], q[2]) - fmax(p[0], q[0]);
        double iy = fmin(p[3], q[3]) - fmax(p[1], q[1]);
        ix = fmax(ix, 0.0);
        iy = fmax(iy, 0.0);
        double inter = ix * iy;
        double a1 = fabs((p[2] - p[0]) * (p[3] - p[1]));
        double a2 = fabs((q[2] - q[0]) * (q[3] - q[1]));
        out[i] = inter / (((a1 + a2) - inter) + eps);
    }
}

// ---------------------------------------------------------------- pairwise N x M
// Output-write bound (4 B per pair).  A workgroup owns a 64-row x 1024-column tile:
// the 64 row boxes (+areas) are staged once in LDS and read back by broadcast; every lane
// keeps 4 consecutive column boxes in registers and emits one 16-byte store per row, so a
// wave writes 1 KiB contiguous per instruction.
#define PW_ROWS 64
#define PW_COLS 1024
__global__ __launch_bounds__(256) void k_iou_pairwise_f32(const float4 *__restrict__ a, int64_t n,
                                                          const float4 *__restrict__ b, int64_t m, float eps,
                                                          float *__restrict__ out, int vec_ok)
{
    __shared__ float4 rbox[PW_ROWS];
    __shared__ float rarea[PW_ROWS];
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.y * PW_ROWS;
    const int64_t j0 = (int64_t)blockIdx.x * PW_COLS + 4 * tid;
    if (tid < PW_ROWS) {
        int64_t i = i0 + tid;
        float4 p = (i < n) ? a[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        rbox[tid] = p;
        rarea[tid] = dg_area(p.x, p.y, p.z, p.w);
    }
    float4 c[4];
    float ca[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int64_t j = j0 + k;
        c[k] = (j < m) ? b[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        ca[k] = dg_area(c[k].x, c[k].y, c[k].z, c[k].w);
    }
    __syncthreads();
    const int rows = (int)((n - i0) < PW_ROWS ? (n - i0) : PW_ROWS);
    if (j0 >= m) return;
    if (vec_ok && j0 + 4 <= m) {
        float *o = out + i0 * m + j0;
        for (int r = 0; r < rows; ++r) {
            float4 p = rbox[r];
            float pa = rarea[r];
            float4 v;
            v.x = dg_iou(p.x, p.y, p.z, p.w, pa, c[0].x, c[0].y, c[0].z, c[0].w, ca[0], eps);
            v.y = dg_iou(p.x, p.y, p.z, p.w, pa, c[1].x, c[1].y, c[1].z, c[1].w, ca[1], eps);
            v.z = dg_iou(p.x, p.y, p.z, p.w, pa, c[2].x, c[2].y, c[2].z, c[2].w, ca[2], eps);
            v.w = dg_iou(p.x, p.y, p.z, p.w, pa, c[3].x, c[3].y, c[3].z, c[3].w, ca[3], eps);
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f t = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(o));          // written once, streamed
            o += m;
        }
    } else {
        for (int r = 0; r < rows; ++r) {
            float4 p = rbox[r];
            float pa = rarea[r];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (j0 + k < m)
                    out[(i0 + r) * m + j0 + k] = dg_iou(p.x, p.y, p.z, p.w, pa, c[k].x, c[k].y, c[k].z, c[k].w, ca[k], eps);
        }
    }
}

// ---------------------------------------------------------------- I3: rois vs one GT, float64
// rcnn/utils.py:50-67 for every proposal of the loop at rcnn/rcnn.py:35-40.
__global__ __launch_bounds__(256) void k_iou_rois_vs_gt_f64(const double *__restrict__ rois, const double *__restrict__ grt,
                                                            int64_t n, double *__restrict__ out, int32_t *__restrict__ flags)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *p = rois + 4 * i;
    double g0 = grt[0], g1 = grt[1], g2 = grt[2], g3 = grt[3];
    int bad = !(p[0] < p[2]) || !(p[1] < p[3]) || !(g0 < g2) || !(g1 < g3);
    double xl = p[0] > g0 ? p[0] : g0, yt = p[1] > g1 ? p[1] : g1;
    double xr = p[2] < g2 ? p[2] : g2, yb = p[3] < g3 ? p[3] : g3;
    double iou;
    if (xr < xl || yb < yt) {
        iou = 0.0;
    } else {
        double inter = (xr - xl) * (yb - yt);
        double a1 = (p[2] - p[0]) * (p[3] - p[1]);
        double a2 = (g2 - g0) * (g3 - g1);
        iou = inter / ((a1 + a2) - inter);
        bad |= !(iou >= 0.0) || !(iou <= 1.0);
    }
    out[i] = iou;
    flags[i] = bad;
}

// ---------------------------------------------------------------- C ABI
static inline int grid_for(int64_t n, int block) { int64_t g = dg_cdiv(n, block); return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

extern "C" int dg_iou_elementwise_f32(const float *bb1, const float *bb2, int64_t n, float eps, float *out, dg_stream_t s)
{
    DG_REQUIRE(n >= 0, "iou_elementwise: n < 0");
    if (n == 0) return DG_OK;
    DG_REQUIRE(bb1 && bb2 && out, "iou_elementwise: null pointer");
    DG_REQUIRE((((uintptr_t)bb1 | (uintptr_t)bb2) & 15) == 0, "iou_elementwise: boxes must be 16-byte aligned");
    hipLaunchKernelGGL(k_iou_elementwise_f32, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)s,
                       (const float4 *)bb1, (const float4 *)bb2, n, eps, out);
    DG_CHECK_LAUNCH("k_iou_elementwise_f32");
    return DG_OK;
}

extern "C" int dg_iou_elementwise_f64(const double *bb1, const double *bb2, int64_t n, double eps, double *out, dg_stream_t s)
{
    DG_REQUIRE(n >= 0, "iou_elementwise_f64: n < 0");
    if (n == 0) return DG_OK;
    DG_REQUIRE(bb1 && bb2 && out, "iou_elementwise_f64: null pointer");
    hipLaunchKernelGGL(k_iou_elementwise_f64, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)s, bb1, bb2, n, eps, out);
    DG_CHECK_LAUNCH("k_iou_elementwise_f64");
    return DG_OK;
}

extern "C" int dg_iou_pairwise_f32(const float *a, int64_t n, const float *b, int64_t m, float eps, float *out, dg_stream_t s)
{
    DG_REQUIRE(n >= 0 && m >= 0, "iou_pairwise: negative size");
    if (n == 0 || m == 0) return DG_OK;
    DG_REQUIRE(a && b && out, "iou_pairwise: null pointer");
    DG_REQUIRE((((uintptr_t)a | (uintptr_t)b) & 15) == 0, "iou_pairwise: boxes must be 16-byte aligned");
    DG_REQUIRE(dg_cdiv(n, PW_ROWS) <= 65535, "iou_pairwise: n too large (max 4,194,240 rows per call)");
    int vec_ok = (m % 4 == 0) && (((uintptr_t)out & 15) == 0);
    dim3 grid((unsigned)dg_cdiv(m, PW_COLS), (unsigned)dg_cdiv(n, PW_ROWS));
    hipLaunchKernelGGL(k_iou_pairwise_f32, grid, dim3(256), 0, (hipStream_t)s, (const float4 *)a, n, (const float4 *)b, m, eps, out, vec_ok);
    DG_CHECK_LAUNCH("k_iou_pairwise_f32");
    return DG_OK;
}

extern "C" int dg_iou_rois_vs_gt_f64(const double *rois, const double *grt, int64_t n, double *out, int32_t *flags, dg_stream_t s)
{
    DG_REQUIRE(n >= 0, "iou_rois_vs_gt: n < 0");
    if (n == 0) return DG_OK;
    DG_REQUIRE(rois && grt && out && flags, "iou_rois_vs_gt: null pointer");
    hipLaunchKernelGGL(k_iou_rois_vs_gt_f64, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, rois, grt, n, out, flags);
    DG_CHECK_LAUNCH("k_iou_rois_vs_gt_f64");
    return DG_OK;
}
