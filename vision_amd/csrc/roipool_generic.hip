// roipool_generic.hip — RoI pooling for the shapes the specialised kernels of roipool.hip do not take: any output size
// PH x PW (the reference hands output_size straight to the adaptive pool, rcnn/fast_rcnn.py:352-355) and feature maps of
// any size (the 7x7 kernels stage 16-bit indices and keep a whole plane in LDS).  Same arithmetic: adaptive bin i over a
// crop of length L is [floor(i*L/P), ceil((i+1)*L/P)); max = first maximum in row-major order, a NaN always wins (ATen);
// avg = sum in row-major order / rows / columns.  Straightforward kernels (the 7x7 / VGG-conv5 case is the tuned one):
// forward one wave per (RoI, 64-channel block) with lanes = channels on the NHWC copy, results staged through LDS in
// groups of 32 bins so that stores are 128-byte runs; backward by global float atomics into a zeroed grad_in.
#include "common.hpp"

#define GB 32          // bins staged per round

template <int MODE>
__global__ __launch_bounds__(64) void k_roipool_fwd_generic(const float *__restrict__ nhwc, int C, int H, int W,
                                                            const int32_t *__restrict__ win, int cblocks, int64_t ntasks, int PH, int PW,
                                                            float *__restrict__ out, int32_t *__restrict__ argmax)
{
    __shared__ float sval[64][GB + 1];
    __shared__ int32_t sidx[64][GB + 1];
    const int lane = threadIdx.x;
    const int NB = PH * PW;
    for (int64_t task = blockIdx.x; task < ntasks; task += gridDim.x) {
        const int cbk = (int)(task % cblocks);
        const int64_t roi = task / cblocks;
        const int c0 = cbk * 64;
        const int32_t *w = win + 5 * roi;
        const int x1 = w[0], y1 = w[1], x2 = w[2], y2 = w[3], b = w[4];
        const int Lh = y2 - y1, Lw = x2 - x1;
        const int cvalid = (C - c0) < 64 ? (C - c0) : 64;
        const int c = c0 + (lane < cvalid ? lane : cvalid - 1);
        const float *base = nhwc + (int64_t)b * H * W * C + c;
        for (int bin0 = 0; bin0 < NB; bin0 += GB) {
            const int nb = (NB - bin0) < GB ? (NB - bin0) : GB;
            for (int j = 0; j < nb; ++j) {
                const int bin = bin0 + j, ph = bin / PW, pw = bin - ph * PW;
                const int hs = y1 + (int)(((int64_t)ph * Lh) / PH), he = y1 + (int)((((int64_t)ph + 1) * Lh + PH - 1) / PH);
                const int ws = x1 + (int)(((int64_t)pw * Lw) / PW), we = x1 + (int)((((int64_t)pw + 1) * Lw + PW - 1) / PW);
                if (MODE == DG_POOL_MAX) {
                    float best = -INFINITY;
                    int bi = hs * W + ws;
                    for (int y = hs; y < he; ++y) {
                        const float *row = base + (int64_t)y * W * C;
                        for (int x = ws; x < we; ++x) {
                            const float v = row[(int64_t)x * C];
                            if ((v > best) || (v != v)) { best = v; bi = y * W + x; }
                        }
                    }
                    sval[lane][j] = best;
                    sidx[lane][j] = bi;
                } else {
                    float sum = 0.f;
                    for (int y = hs; y < he; ++y) {
                        const float *row = base + (int64_t)y * W * C;
                        for (int x = ws; x < we; ++x) sum += row[(int64_t)x * C];
                    }
                    sval[lane][j] = sum / (float)(he - hs) / (float)(we - ws);
                }
            }
            __syncthreads();
            // [channel][bin] -> runs of nb consecutive outputs per channel
            const int total = cvalid * nb;
            for (int e = lane; e < total; e += 64) {
                const int cc = e / nb, j = e - cc * nb;
                const int64_t o = (roi * C + c0 + cc) * NB + bin0 + j;
                out[o] = sval[cc][j];
                if (MODE == DG_POOL_MAX && argmax) argmax[o] = sidx[cc][j];
            }
            __syncthreads();
        }
    }
}

// max: grad_in[b, c, argmax] += grad_out, one thread per element.  IDX16: the argmax is the uint16 form.
template <bool IDX16>
__global__ __launch_bounds__(256) void k_roipool_bwd_generic_max(const float *__restrict__ grad_out, const void *__restrict__ argmax,
                                                                 const int32_t *__restrict__ win, int64_t total, int C, int64_t HW, int NB,
                                                                 float *__restrict__ grad_in)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const int64_t rc = e / NB;                    // roi * C + c
        const int64_t roi = rc / C;
        const int c = (int)(rc - roi * C);
        const int b = win[5 * roi + 4];
        const int64_t a = IDX16 ? (int64_t)reinterpret_cast<const unsigned short *>(argmax)[e] : (int64_t)reinterpret_cast<const int32_t *>(argmax)[e];
        // (an index outside the plane cannot come from the forward; one handed in from elsewhere must not become a write outside
        //  grad_in -- the tuned kernel's planes live in LDS, where the hardware drops an out-of-range access)
        if ((uint64_t)a < (uint64_t)HW) atomicAdd(&grad_in[((int64_t)b * C + c) * HW + a], grad_out[e]);
    }
}

// avg: grad_out / rows / cols (the oracle's and autograd's order) added to every cell of the bin
__global__ __launch_bounds__(256) void k_roipool_bwd_generic_avg(const float *__restrict__ grad_out, const int32_t *__restrict__ win,
                                                                 int64_t total, int C, int H, int W, int PH, int PW,
                                                                 float *__restrict__ grad_in)
{
    const int NB = PH * PW;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const int64_t rc = e / NB;
        const int bin = (int)(e - rc * NB);
        const int64_t roi = rc / C;
        const int c = (int)(rc - roi * C);
        const int32_t *w = win + 5 * roi;
        const int x1 = w[0], y1 = w[1], Lw = w[2] - x1, Lh = w[3] - y1, b = w[4];
        const int ph = bin / PW, pw = bin - ph * PW;
        const int hs = y1 + (int)(((int64_t)ph * Lh) / PH), he = y1 + (int)((((int64_t)ph + 1) * Lh + PH - 1) / PH);
        const int ws = x1 + (int)(((int64_t)pw * Lw) / PW), we = x1 + (int)((((int64_t)pw + 1) * Lw + PW - 1) / PW);
        const float g = grad_out[e] / (float)(he - hs) / (float)(we - ws);
        float *plane = grad_in + ((int64_t)b * C + c) * (int64_t)H * W;
        for (int y = hs; y < he; ++y)
            for (int x = ws; x < we; ++x) atomicAdd(&plane[(int64_t)y * W + x], g);
    }
}

int dg_roipool_fwd_generic(const float *img_nhwc, int64_t B, int64_t C, int64_t H, int64_t W, const int32_t *win, int64_t n,
                           int64_t PH, int64_t PW, int mode, float *out, int32_t *argmax, hipStream_t st)
{
    DG_REQUIRE(H * W < (1ll << 31), "roipool_fwd: H*W >= 2^31");
    const int64_t cblocks = dg_cdiv(C, 64), ntasks = n * cblocks;
    const int64_t maxgrid = 256 * 16;
    dim3 grid((unsigned)(ntasks < maxgrid ? ntasks : maxgrid));
    if (mode == DG_POOL_MAX)
        hipLaunchKernelGGL(k_roipool_fwd_generic<DG_POOL_MAX>, grid, dim3(64), 0, st, img_nhwc, (int)C, (int)H, (int)W, win, (int)cblocks,
                           ntasks, (int)PH, (int)PW, out, argmax);
    else
        hipLaunchKernelGGL(k_roipool_fwd_generic<DG_POOL_AVG>, grid, dim3(64), 0, st, img_nhwc, (int)C, (int)H, (int)W, win, (int)cblocks,
                           ntasks, (int)PH, (int)PW, out, argmax);
    DG_CHECK_LAUNCH("k_roipool_fwd_generic");
    return DG_OK;
}

int dg_roipool_bwd_generic(const float *grad_out, const void *argmax, bool a16, const int32_t *win, int64_t n, int64_t B, int64_t C,
                           int64_t H, int64_t W, int64_t PH, int64_t PW, int mode, float *grad_in, hipStream_t st)
{
    if (hipMemsetAsync(grad_in, 0, (size_t)B * C * H * W * 4, st) != hipSuccess) { dg_set_error("roipool_bwd: memset failed"); return DG_EHIP; }
    const int64_t total = n * C * PH * PW;
    if (total == 0) return DG_OK;
    const int64_t blocks = dg_cdiv(total, 256);
    dim3 grid((unsigned)(blocks < 65536 ? blocks : 65536));
    if (mode == DG_POOL_MAX) {
        if (a16) hipLaunchKernelGGL(k_roipool_bwd_generic_max<true>, grid, dim3(256), 0, st, grad_out, argmax, win, total, (int)C, H * W, (int)(PH * PW), grad_in);
        else hipLaunchKernelGGL(k_roipool_bwd_generic_max<false>, grid, dim3(256), 0, st, grad_out, argmax, win, total, (int)C, H * W, (int)(PH * PW), grad_in);
    } else {
        hipLaunchKernelGGL(k_roipool_bwd_generic_avg, grid, dim3(256), 0, st, grad_out, win, total, (int)C, (int)H, (int)W, (int)PH, (int)PW, grad_in);
    }
    DG_CHECK_LAUNCH("k_roipool_bwd_generic");
    return DG_OK;
}
