// loss.hip — YOLOv1 loss with its box geometry fused: in-loss decode, IoU, responsible-slot select, the five
// squared-error terms and d loss / d modely in one pass.
// Reference: Loss.forward, yolo/yolo_v1/modules.py:234-306 (about 60 small torch kernels per call + autograd).
//   object cells  t[...,4] > 0 (:243), background t[...,4] == 0 (:244)
//   decode        :262-268; the cell corner is float64 (:204-206): x1 = fp32(lt + x*(1/7) - w/2), x2 = w + x1;
//                 slot 2 takes its half-size from columns 5:7 (:264) -- kept when quirk != 0
//   IoU           :271-272 (module-level get_iou, eps 1e-7); responsible slot 1 iff iou1 >= iou2 (:275-276)
//   terms         :283-302; loss = (5*(xy+wh) + obj + .5*noobj + label)/n (:304-305)
// The reference does not detach the IoU target, so the responsible box also receives d(best IoU); ATen splits the
// gradient of min/max evenly on ties.  One thread per cell; term sums are double, reduced in a fixed order.
#include "common.hpp"

__device__ __forceinline__ float l1_sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__device__ __forceinline__ void l1_box(double lt, float c, float half_src, float ext, float &lo, float &hi)
{
    const float a = c * (float)(1.0 / 7.0);
    const float hw = half_src / 2.0f;
    lo = (float)((lt + (double)a) - (double)hw);
    hi = ext + lo;
}

// IoU of the predicted box b against the target box u, and d iou / d b
__device__ __forceinline__ float l1_iou_grad(const float (&b)[4], const float (&u)[4], float (&g)[4])
{
    const float xl = fmaxf(b[0], u[0]), yt = fmaxf(b[1], u[1]), xr = fminf(b[2], u[2]), yb = fminf(b[3], u[3]);
    const float ixr = xr - xl, iyr = yb - yt;
    const float ix = ixr < 0.f ? 0.f : ixr, iy = iyr < 0.f ? 0.f : iyr;
    const float I = ix * iy;
    const float w = b[2] - b[0], h = b[3] - b[1];
    const float A1 = fabsf(w * h), A2 = fabsf((u[2] - u[0]) * (u[3] - u[1]));
    const float U = ((A1 + A2) - I) + 1e-7f;
    const float iou = I / U;
    const float gI = 1.0f / U + I / (U * U);
    const float gA = -I / (U * U);
    const float gix = (ixr < 0.f) ? 0.f : gI * iy, giy = (iyr < 0.f) ? 0.f : gI * ix;
    const float gwh = gA * l1_sgn(w * h);
    const float gw = gwh * h, gh = gwh * w;
    const float fx2 = (b[2] < u[2]) ? 1.f : ((b[2] == u[2]) ? 0.5f : 0.f), fx1 = (b[0] > u[0]) ? 1.f : ((b[0] == u[0]) ? 0.5f : 0.f);
    const float fy2 = (b[3] < u[3]) ? 1.f : ((b[3] == u[3]) ? 0.5f : 0.f), fy1 = (b[1] > u[1]) ? 1.f : ((b[1] == u[1]) ? 0.5f : 0.f);
    g[0] = -gix * fx1 - gw; g[2] = gix * fx2 + gw;
    g[1] = -giy * fy1 - gh; g[3] = giy * fy2 + gh;
    return iou;
}

#define L1_THREADS 256
__global__ __launch_bounds__(L1_THREADS) void k_yolo1_loss(const float *__restrict__ y, const float *__restrict__ t, int64_t cells,
                                                           int D, int quirk, float gN, double *__restrict__ partial,
                                                           float *__restrict__ grad)
{
    __shared__ double red[L1_THREADS / 64][5];
    const int64_t cell = (int64_t)blockIdx.x * L1_THREADS + threadIdx.x;
    double term[5] = {0, 0, 0, 0, 0};                       // xy, wh, obj, noobj, label
    if (cell < cells) {
        const float *p = y + cell * D, *q = t + cell * D;
        float *gp = grad ? grad + cell * D : nullptr;
        const int ij = (int)(cell % 49), i = ij / 7, j = ij - 7 * i;
        const float tc = q[4];
        float g10[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};     // gradient of the two box slots
        if (tc > 0.f) {
            const double ltx = (double)j * (1.0 / 7.0), lty = (double)i * (1.0 / 7.0);
            float pv[10], qv[4];
#pragma unroll
            for (int c = 0; c < 10; ++c) pv[c] = p[c];
#pragma unroll
            for (int c = 0; c < 4; ++c) qv[c] = q[c];
            float b1[4], b2[4], u[4], g1[4], g2[4];
            l1_box(ltx, pv[0], pv[2], pv[2], b1[0], b1[2]);
            l1_box(lty, pv[1], pv[3], pv[3], b1[1], b1[3]);
            l1_box(ltx, pv[5], quirk ? pv[5] : pv[7], pv[7], b2[0], b2[2]);
            l1_box(lty, pv[6], quirk ? pv[6] : pv[8], pv[8], b2[1], b2[3]);
            l1_box(ltx, qv[0], qv[2], qv[2], u[0], u[2]);
            l1_box(lty, qv[1], qv[3], qv[3], u[1], u[3]);
            const float iou1 = l1_iou_grad(b1, u, g1), iou2 = l1_iou_grad(b2, u, g2);
            const bool r = !(iou1 >= iou2);                  // responsible slot: 0 or 1
            const float best = r ? iou2 : iou1;
            const float rx = r ? pv[5] : pv[0], ry = r ? pv[6] : pv[1], rw = r ? pv[7] : pv[2], rh = r ? pv[8] : pv[3],
                        rc = r ? pv[9] : pv[4];
            const float dx = rx - qv[0], dy = ry - qv[1];
            term[0] = (double)(dx * dx) + (double)(dy * dy);
            const float sw = l1_sgn(rw) * sqrtf(fabsf(rw)), sh = l1_sgn(rh) * sqrtf(fabsf(rh));
            const float dw = sw - sqrtf(qv[2]), dh = sh - sqrtf(qv[3]);
            term[1] = (double)(dw * dw) + (double)(dh * dh);
            const float dob = best - rc;
            term[2] = (double)(dob * dob);
            double lab = 0;
            for (int c = 10; c < D; ++c) {
                const float d = p[c] - q[c];
                lab += (double)(d * d);
                if (gp) gp[c] = 2.f * d * gN;
            }
            term[4] = lab;
            if (gp) {
                float gr[5];
                gr[0] = 10.f * dx * gN; gr[1] = 10.f * dy * gN;
                gr[2] = 10.f * dw * gN * (l1_sgn(rw) * (0.5f / sqrtf(fabsf(rw))) * l1_sgn(rw));
                gr[3] = 10.f * dh * gN * (l1_sgn(rh) * (0.5f / sqrtf(fabsf(rh))) * l1_sgn(rh));
                gr[4] = -2.f * dob * gN;
                const float gb = 2.f * dob * gN;             // d loss / d best iou
                const float inv7 = (float)(1.0 / 7.0);
#pragma unroll
                for (int a = 0; a < 2; ++a) {                // a = 0: x, 1: y
                    const float glo = gb * (r ? g2[a] : g1[a]), ghi = gb * (r ? g2[a + 2] : g1[a + 2]), gt = glo + ghi;
                    if (r && quirk) { gr[a] += gt * inv7 - gt * 0.5f; gr[a + 2] += ghi; }
                    else { gr[a] += gt * inv7; gr[a + 2] += ghi - gt * 0.5f; }
                }
#pragma unroll
                for (int c = 0; c < 5; ++c) { g10[c] = r ? 0.f : gr[c]; g10[5 + c] = r ? gr[c] : 0.f; }
            }
        } else {
            if (tc == 0.f) {
                const float d4 = p[4] - tc, d9 = p[9] - q[9];
                term[3] = (double)(d4 * d4) + (double)(d9 * d9);
                g10[4] = d4 * gN; g10[9] = d9 * gN;
            }
            if (gp) for (int c = 10; c < D; ++c) gp[c] = 0.f;
        }
        if (gp) {
#pragma unroll
            for (int c = 0; c < 10; ++c) gp[c] = g10[c];
        }
    }
    // block sums: wave shuffle tree, then the 4 waves in order
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double v = term[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        double v = 0;
#pragma unroll
        for (int w = 0; w < L1_THREADS / 64; ++w) v += red[w][threadIdx.x];
        partial[(int64_t)blockIdx.x * 5 + threadIdx.x] = v;
    }
}

// one workgroup: fixed-order sum of the block partials -> out[0] = loss, out[1..5] = xy, wh, obj, noobj, label sums
__global__ __launch_bounds__(256) void k_yolo1_loss_finish(const double *__restrict__ partial, int64_t nblocks, double n, float *__restrict__ out)
{
    __shared__ double red[4][5];
    double acc[5] = {0, 0, 0, 0, 0};
    for (int64_t b = threadIdx.x; b < nblocks; b += 256)
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[k] += partial[b * 5 + k];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) s[k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
        out[0] = (float)((5.0 * (s[0] + s[1]) + s[2] + 0.5 * s[3] + s[4]) / n);
#pragma unroll
        for (int k = 0; k < 5; ++k) out[1 + k] = (float)s[k];
    }
}

extern "C" size_t dg_yolo1_loss_workspace_bytes(int64_t n)
{
    if (n <= 0) return 0;
    return (size_t)dg_cdiv(n * 49, L1_THREADS) * 5 * sizeof(double);
}

extern "C" int dg_yolo1_loss_f32(const float *modely, const float *targety, int64_t n, int64_t num_classes, int quirk_slot2,
                                 float *loss6, float *grad, void *ws, size_t ws_bytes, dg_stream_t s)
{
    DG_REQUIRE(n > 0 && num_classes >= 0 && num_classes <= 4096, "yolo1_loss: bad shape");
    DG_REQUIRE(modely && targety && loss6 && ws, "yolo1_loss: null pointer");
    DG_REQUIRE((((uintptr_t)ws) & 7) == 0, "yolo1_loss: workspace must be 8-byte aligned");
    const size_t need = dg_yolo1_loss_workspace_bytes(n);
    if (ws_bytes < need) { dg_set_error("yolo1_loss: workspace %zu < %zu bytes", ws_bytes, need); return DG_ENOMEM; }
    const int64_t cells = n * 49, nblocks = dg_cdiv(cells, L1_THREADS);
    DG_REQUIRE(nblocks < (1ll << 31), "yolo1_loss: batch too large");
    hipStream_t st = (hipStream_t)s;
    hipLaunchKernelGGL(k_yolo1_loss, dim3((unsigned)nblocks), dim3(L1_THREADS), 0, st, modely, targety, cells, (int)(10 + num_classes),
                       quirk_slot2, 1.0f / (float)n, (double *)ws, grad);
    hipLaunchKernelGGL(k_yolo1_loss_finish, dim3(1), dim3(256), 0, st, (const double *)ws, nblocks, (double)n, loss6);
    DG_CHECK_LAUNCH("k_yolo1_loss");
    return DG_OK;
}
