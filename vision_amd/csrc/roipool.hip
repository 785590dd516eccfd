// roipool.hip — Fast R-CNN RoI pooling forward/backward for gfx950.
// Reference: slowroipool.foward, rcnn/fast_rcnn.py:351-377 (window :366-369, per-RoI adaptive pool
// :372-374).  Adaptive bin i over a crop of length L: [floor(i*L/P), ceil((i+1)*L/P)) (ATen).
//
// Forward: one wave per (RoI, 64-channel block).  The feature map is read from an NHWC staging
// copy so that the 64 lanes (= 64 consecutive channels) of a wave read 256 contiguous bytes per
// pixel; the 64x49 results are transposed through LDS ([channel][bin], odd stride 49 => conflict
// free) and leave as 16-byte stores over the contiguous (n,C,7,7) block of that RoI.
// blockIdx.x = channel block, so blocks that share an XCD (blockIdx % 8) share a channel slice
// of the map in that XCD's L2.
//
// Backward: one workgroup per (image, CB channels) keeps that slice of grad_in in LDS, walks the
// image's RoI list reading grad_out/argmax as contiguous runs, accumulates with LDS float atomics
// and writes the slice out once: no global atomics, grad_in written exactly once.
#include "common.hpp"

#define RP 7
#define RBINS 49

// ---------------------------------------------------------------- windows (:366-369 + python slicing)
__global__ __launch_bounds__(256) void k_roi_windows(const double *__restrict__ rois, const int64_t *__restrict__ roi_idx,
                                                     int64_t n, int B, int H, int W, int32_t *__restrict__ win,
                                                     int32_t *__restrict__ status)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *r = rois + 4 * i;
    int x1 = (int)floor(r[0] * (double)W), x2 = (int)ceil(r[2] * (double)W);
    int y1 = (int)floor(r[1] * (double)H), y2 = (int)ceil(r[3] * (double)H);
    if (x1 < 0) x1 = (x1 + W < 0) ? 0 : x1 + W;     // python: negative start wraps, then clamps
    if (y1 < 0) y1 = (y1 + H < 0) ? 0 : y1 + H;
    if (x2 > W) x2 = W;
    if (y2 > H) y2 = H;
    int64_t b = roi_idx[i];
    int st = 0;
    if (x2 <= x1 || y2 <= y1) { st |= 1; x1 = 0; y1 = 0; x2 = 1; y2 = 1; }   // keep later kernels in bounds
    if (b < 0 || b >= B) { st |= 2; b = 0; }
    if (st) atomicOr(status, st);
    int32_t *w = win + 5 * i;
    w[0] = x1; w[1] = y1; w[2] = x2; w[3] = y2; w[4] = (int32_t)b;
}

// ---------------------------------------------------------------- NCHW -> NHWC staging copy
__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float *__restrict__ src, int C, int HW, float *__restrict__ dst)
{
    __shared__ float tile[64][65];
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    src += (int64_t)b * C * HW;
    dst += (int64_t)b * C * HW;
    for (int k = ty; k < 64; k += 4) {
        int c = c0 + k, p = p0 + tx;
        tile[k][tx] = (c < C && p < HW) ? src[(int64_t)c * HW + p] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 64; k += 4) {
        int p = p0 + k, c = c0 + tx;
        if (c < C && p < HW) dst[(int64_t)p * C + c] = tile[tx][k];
    }
}

// ---------------------------------------------------------------- forward
template <int MODE>
__global__ __launch_bounds__(64) void k_roipool_fwd(const float *__restrict__ nhwc, int C, int H, int W,
                                                    const int32_t *__restrict__ win, int cblocks, float *__restrict__ out,
                                                    int32_t *__restrict__ argmax, int vec_ok)
{
    __shared__ __attribute__((aligned(16))) float sval[64 * RBINS];
    __shared__ __attribute__((aligned(16))) unsigned short sidx[64 * RBINS];
    const int lane = threadIdx.x;
    // 1-D grid, channel block fastest: blocks that share an XCD (blockIdx % 8) share a channel slice
    const int cbk = blockIdx.x % cblocks;
    const int64_t roi = blockIdx.x / cblocks;
    const int c0 = cbk * 64;
    const int32_t *w = win + 5 * roi;
    const int x1 = w[0], y1 = w[1], x2 = w[2], y2 = w[3], b = w[4];
    const int Lh = y2 - y1, Lw = x2 - x1;
    const int cvalid = (C - c0) < 64 ? (C - c0) : 64;
    const int c = c0 + (lane < cvalid ? lane : cvalid - 1);
    const float *base = nhwc + (int64_t)b * H * W * C + c;

#pragma unroll 1
    for (int ph = 0; ph < RP; ++ph) {
        const int hs = y1 + (ph * Lh) / RP, he = y1 + ((ph + 1) * Lh + RP - 1) / RP;
#pragma unroll 1
        for (int pw = 0; pw < RP; ++pw) {
            const int ws = x1 + (pw * Lw) / RP, we = x1 + ((pw + 1) * Lw + RP - 1) / RP;
            if (MODE == DG_POOL_MAX) {
                float best = -INFINITY;
                int bi = hs * W + ws;
                for (int y = hs; y < he; ++y) {
                    const float *row = base + (int64_t)y * W * C;
                    for (int x = ws; x < we; x += 4) {
                        // 4 independent loads in flight; positions past the bin are clamped re-reads
                        // and masked out of the update (ATen: first max wins, NaN always wins)
                        const int xa = x, xb = min(x + 1, we - 1), xc = min(x + 2, we - 1), xd = min(x + 3, we - 1);
                        const float va = row[(int64_t)xa * C], vb = row[(int64_t)xb * C];
                        const float vc = row[(int64_t)xc * C], vd = row[(int64_t)xd * C];
                        if ((va > best) || (va != va)) { best = va; bi = y * W + xa; }
                        if ((x + 1 < we) && ((vb > best) || (vb != vb))) { best = vb; bi = y * W + xb; }
                        if ((x + 2 < we) && ((vc > best) || (vc != vc))) { best = vc; bi = y * W + xc; }
                        if ((x + 3 < we) && ((vd > best) || (vd != vd))) { best = vd; bi = y * W + xd; }
                    }
                }
                sval[lane * RBINS + ph * RP + pw] = best;
                sidx[lane * RBINS + ph * RP + pw] = (unsigned short)bi;
            } else {
                float sum = 0.f;
                for (int y = hs; y < he; ++y) {
                    const float *row = base + (int64_t)y * W * C;
                    for (int x = ws; x < we; ++x) sum += row[(int64_t)x * C];
                }
                sval[lane * RBINS + ph * RP + pw] = sum / (float)(he - hs) / (float)(we - ws);
            }
        }
    }
    __syncthreads();
    // contiguous run of cvalid*49 outputs for this (roi, channel block)
    const int64_t obase = (roi * C + c0) * RBINS;
    const int total = cvalid * RBINS;
    if (vec_ok) {
        for (int e = lane * 4; e < total; e += 256) {
            *reinterpret_cast<float4 *>(out + obase + e) = *reinterpret_cast<const float4 *>(sval + e);
            if (MODE == DG_POOL_MAX && argmax) {
                const ushort4 u = *reinterpret_cast<const ushort4 *>(sidx + e);
                *reinterpret_cast<int4 *>(argmax + obase + e) = make_int4(u.x, u.y, u.z, u.w);
            }
        }
    } else {
        for (int e = lane; e < total; e += 64) {
            out[obase + e] = sval[e];
            if (MODE == DG_POOL_MAX && argmax) argmax[obase + e] = sidx[e];
        }
    }
}

// ---------------------------------------------------------------- backward: per-image RoI lists
__global__ __launch_bounds__(256) void k_roi_count(const int32_t *__restrict__ win, int64_t n, int32_t *__restrict__ cnt)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&cnt[win[5 * i + 4]], 1);
}
// exclusive scan of cnt[0..B) into off[0..B]; cursor[] = off[] ; single workgroup
__global__ __launch_bounds__(1024) void k_roi_scan(const int32_t *__restrict__ cnt, int B, int32_t *__restrict__ off,
                                                   int32_t *__restrict__ cursor)
{
    __shared__ int32_t part[1024];
    const int tid = threadIdx.x;
    const int per = (B + 1023) / 1024;
    int s = 0;
    for (int k = 0; k < per; ++k) { int i = tid * per + k; if (i < B) s += cnt[i]; }
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        int v = (tid >= d) ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int k = 0; k < per; ++k) {
        int i = tid * per + k;
        if (i < B) { off[i] = run; cursor[i] = run; run += cnt[i]; }
    }
    if (tid == 1023) off[B] = part[1023];
}
__global__ __launch_bounds__(256) void k_roi_fill(const int32_t *__restrict__ win, int64_t n, int32_t *__restrict__ cursor,
                                                  int32_t *__restrict__ list)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) list[atomicAdd(&cursor[win[5 * i + 4]], 1)] = (int32_t)i;
}

// grid: x = channel block (CB channels), y = image.  Dynamic LDS: CB*H*W floats.
template <int MODE>
__global__ __launch_bounds__(256) void k_roipool_bwd(const float *__restrict__ grad_out, const int32_t *__restrict__ argmax,
                                                     const int32_t *__restrict__ win, const int32_t *__restrict__ off,
                                                     const int32_t *__restrict__ list, int C, int H, int W, int CB,
                                                     float *__restrict__ grad_in)
{
    extern __shared__ __attribute__((aligned(16))) float plane[];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, c0 = blockIdx.x * CB;
    const int cb = (C - c0) < CB ? (C - c0) : CB;
    const int HW = H * W;
    for (int i = tid; i < cb * HW; i += 256) plane[i] = 0.f;
    __syncthreads();
    const int r0 = off[b], r1 = off[b + 1];
    const int per = cb * RBINS;                      // contiguous elements per RoI for this channel slice
    const int64_t work = (int64_t)(r1 - r0) * per;
    for (int64_t t = tid; t < work; t += 256) {
        const int rr = (int)(t / per), e = (int)(t - (int64_t)rr * per);
        const int roi = list[r0 + rr];
        const int64_t o = ((int64_t)roi * C + c0) * RBINS + e;
        const float g = grad_out[o];
        const int ch = e / RBINS;
        if (MODE == DG_POOL_MAX) {
            atomicAdd(&plane[ch * HW + argmax[o]], g);
        } else {
            const int bin = e - ch * RBINS, ph = bin / RP, pw = bin - ph * RP;
            const int32_t *w = win + 5 * (int64_t)roi;
            const int x1 = w[0], y1 = w[1], Lw = w[2] - x1, Lh = w[3] - y1;
            const int hs = y1 + (ph * Lh) / RP, he = y1 + ((ph + 1) * Lh + RP - 1) / RP;
            const int ws = x1 + (pw * Lw) / RP, we = x1 + ((pw + 1) * Lw + RP - 1) / RP;
            const float gd = g / (float)(he - hs) / (float)(we - ws);
            for (int y = hs; y < he; ++y)
                for (int x = ws; x < we; ++x) atomicAdd(&plane[ch * HW + y * W + x], gd);
        }
    }
    __syncthreads();
    float *gi = grad_in + ((int64_t)b * C + c0) * HW;
    for (int i = tid; i < cb * HW; i += 256) gi[i] = plane[i];
}

// ---------------------------------------------------------------- C ABI
extern "C" int dg_roi_windows(const double *rois, const int64_t *roi_idx, int64_t n, int64_t B, int64_t H, int64_t W,
                              int32_t *win, int32_t *status, dg_stream_t s)
{
    DG_REQUIRE(n >= 0 && B > 0 && H > 0 && W > 0, "roi_windows: bad shape");
    if (n == 0) return DG_OK;
    DG_REQUIRE(rois && roi_idx && win && status, "roi_windows: null pointer");
    hipLaunchKernelGGL(k_roi_windows, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, rois, roi_idx, n, (int)B, (int)H,
                       (int)W, win, status);
    DG_CHECK_LAUNCH("k_roi_windows");
    return DG_OK;
}

extern "C" int dg_nchw_to_nhwc_f32(const float *src, int64_t B, int64_t C, int64_t H, int64_t W, float *dst, dg_stream_t s)
{
    DG_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad shape");
    DG_REQUIRE(src && dst, "nchw_to_nhwc: null pointer");
    DG_REQUIRE(B <= 65535 && dg_cdiv(C, 64) <= 65535, "nchw_to_nhwc: shape too large");
    dim3 grid((unsigned)dg_cdiv(H * W, 64), (unsigned)dg_cdiv(C, 64), (unsigned)B);
    hipLaunchKernelGGL(k_nchw_to_nhwc, grid, dim3(256), 0, (hipStream_t)s, src, (int)C, (int)(H * W), dst);
    DG_CHECK_LAUNCH("k_nchw_to_nhwc");
    return DG_OK;
}

extern "C" int dg_roipool_fwd_f32(const float *img_nhwc, int64_t B, int64_t C, int64_t H, int64_t W,
                                  const int32_t *win, int64_t n, int64_t P, int mode, float *out, int32_t *argmax, dg_stream_t s)
{
    DG_REQUIRE(n >= 0 && B > 0 && C > 0 && H > 0 && W > 0, "roipool_fwd: bad shape");
    DG_REQUIRE(P == RP, "roipool_fwd: only output_size 7 is built (got %lld)", (long long)P);
    DG_REQUIRE(mode == DG_POOL_MAX || mode == DG_POOL_AVG, "roipool_fwd: bad mode");
    DG_REQUIRE(H * W <= 65535, "roipool_fwd: H*W > 65535 unsupported (16-bit staged argmax)");
    if (n == 0) return DG_OK;
    DG_REQUIRE(img_nhwc && win && out, "roipool_fwd: null pointer");
    int vec_ok = (C % 4 == 0) && ((((uintptr_t)out) & 15) == 0) && (!argmax || (((uintptr_t)argmax) & 15) == 0);
    hipStream_t st = (hipStream_t)s;
    const int64_t cblocks = dg_cdiv(C, 64);
    DG_REQUIRE(n * cblocks < (1ll << 31), "roipool_fwd: n * ceil(C/64) must be < 2^31");
    dim3 grid((unsigned)(n * cblocks));
    if (mode == DG_POOL_MAX)
        hipLaunchKernelGGL((k_roipool_fwd<DG_POOL_MAX>), grid, dim3(64), 0, st, img_nhwc, (int)C, (int)H, (int)W, win, (int)cblocks, out, argmax, vec_ok);
    else
        hipLaunchKernelGGL((k_roipool_fwd<DG_POOL_AVG>), grid, dim3(64), 0, st, img_nhwc, (int)C, (int)H, (int)W, win, (int)cblocks, out, argmax, vec_ok);
    DG_CHECK_LAUNCH("k_roipool_fwd");
    return DG_OK;
}

struct BwdWs { int32_t *cnt, *off, *cursor, *list; size_t total; };
static BwdWs bwd_layout(int64_t B, int64_t n, void *base)
{
    BwdWs w;
    char *p = (char *)base;
    size_t o = 0;
    auto take = [&](size_t bytes) { char *r = p + o; o += (bytes + 255) & ~(size_t)255; return r; };
    w.cnt = (int32_t *)take((size_t)B * 4);
    w.off = (int32_t *)take((size_t)(B + 1) * 4);
    w.cursor = (int32_t *)take((size_t)B * 4);
    w.list = (int32_t *)take((size_t)(n > 0 ? n : 1) * 4);
    w.total = o;
    return w;
}

extern "C" size_t dg_roipool_bwd_workspace_bytes(int64_t B, int64_t n)
{
    if (B <= 0 || n < 0) return 0;
    return bwd_layout(B, n, nullptr).total;
}

extern "C" int dg_roipool_bwd_f32(const float *grad_out, const int32_t *argmax, const int32_t *win, int64_t n,
                                  int64_t B, int64_t C, int64_t H, int64_t W, int64_t P, int mode, float *grad_in,
                                  void *ws, size_t ws_bytes, dg_stream_t s)
{
    DG_REQUIRE(n >= 0 && B > 0 && C > 0 && H > 0 && W > 0, "roipool_bwd: bad shape");
    DG_REQUIRE(P == RP, "roipool_bwd: only output_size 7 is built");
    DG_REQUIRE(mode == DG_POOL_MAX || mode == DG_POOL_AVG, "roipool_bwd: bad mode");
    DG_REQUIRE(grad_in && ws, "roipool_bwd: null pointer");
    DG_REQUIRE(n == 0 || (grad_out && win), "roipool_bwd: null pointer");
    DG_REQUIRE(mode != DG_POOL_MAX || n == 0 || argmax, "roipool_bwd: max mode needs argmax");
    DG_REQUIRE(B <= 65535 && n < (1ll << 31), "roipool_bwd: shape too large");
    BwdWs w = bwd_layout(B, n, ws);
    if (ws_bytes < w.total) { dg_set_error("roipool_bwd: workspace %zu < %zu bytes", ws_bytes, w.total); return DG_ENOMEM; }
    const int64_t HW = H * W;
    DG_REQUIRE(HW * 4 <= 64 * 1024, "roipool_bwd: feature plane H*W*4 > 64 KiB does not fit the LDS accumulator");
    hipStream_t st = (hipStream_t)s;
    if (hipMemsetAsync(w.cnt, 0, (size_t)B * 4, st) != hipSuccess) { dg_set_error("roipool_bwd: memset failed"); return DG_EHIP; }
    if (n > 0) hipLaunchKernelGGL(k_roi_count, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, st, win, n, w.cnt);
    hipLaunchKernelGGL(k_roi_scan, dim3(1), dim3(1024), 0, st, w.cnt, (int)B, w.off, w.cursor);
    if (n > 0) hipLaunchKernelGGL(k_roi_fill, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, st, win, n, w.cursor, w.list);
    // channels per workgroup: as many as fit ~32 KiB of LDS, but keep >= ~1024 workgroups when possible
    int CB = 1;
    while (CB < 16 && (int64_t)CB * 2 * HW * 4 <= 32 * 1024 && B * dg_cdiv(C, CB * 2) >= 1024) CB *= 2;
    dim3 grid((unsigned)dg_cdiv(C, CB), (unsigned)B);
    size_t lds = (size_t)CB * HW * 4;
    if (mode == DG_POOL_MAX)
        hipLaunchKernelGGL((k_roipool_bwd<DG_POOL_MAX>), grid, dim3(256), lds, st, grad_out, argmax, win, w.off, w.list, (int)C, (int)H, (int)W, CB, grad_in);
    else
        hipLaunchKernelGGL((k_roipool_bwd<DG_POOL_AVG>), grid, dim3(256), lds, st, grad_out, argmax, win, w.off, w.list, (int)C, (int)H, (int)W, CB, grad_in);
    DG_CHECK_LAUNCH("k_roipool_bwd");
    return DG_OK;
}
