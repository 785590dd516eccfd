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
#include <stdlib.h>

#define RP 7
#ifndef DG_FWD_STAGED_MAXW
#define DG_FWD_STAGED_MAXW 24
#endif
#ifndef DG_FWD_RECOVER_V2
#define DG_FWD_RECOVER_V2 1
#endif
#ifndef DG_FWD_WIDE
#define DG_FWD_WIDE 1
#endif
#ifndef DG_BWD_XCD_REMAP
#define DG_BWD_XCD_REMAP 1
#endif
#define RBINS 49

// ---------------------------------------------------------------- windows (:366-369 + python slicing)
__global__ __launch_bounds__(256) void k_roi_windows(const double *__restrict__ rois, const int64_t *__restrict__ roi_idx,
                                                     int64_t n, int B, int H, int W, int32_t *__restrict__ win,
                                                     int32_t *__restrict__ status)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *r = rois + 4 * i;
    // :366-369 then the crop `img[idx, :, y1:y2, x1:x2]` (:374): python slice bounds -- BOTH wrap when negative, both clamp to
    // [0, size]; .astype(int) of a NaN / infinity / value outside int64 is INT64_MIN on x86-64, which the slice clamps to 0
    auto bound = [](double v, int size) -> int {
        if (!(v >= -9223372036854775808.0 && v < 9223372036854775808.0)) return 0;
        if (v < -(double)size) return 0;
        if (v > (double)size) return size;
        const int q = (int)v;
        return q < 0 ? q + size : q;
    };
    int x1 = bound(floor(r[0] * (double)W), W), x2 = bound(ceil(r[2] * (double)W), W);
    int y1 = bound(floor(r[1] * (double)H), H), y2 = bound(ceil(r[3] * (double)H), H);
    int64_t b = roi_idx[i];
    int st = 0;
    if (x2 <= x1 || y2 <= y1) { st |= 1; x1 = 0; y1 = 0; x2 = 1; y2 = 1; }   // keep later kernels in bounds
    if (b < 0) b += B;                                  // `img[roi_idx[i]]`: a negative index counts from the last image
    if (b < 0 || b >= B) { st |= 2; b = 0; }
    if (st) atomicOr(status, st);
    int32_t *w = win + 5 * i;
    w[0] = x1; w[1] = y1; w[2] = x2; w[3] = y2; w[4] = (int32_t)b;
}

// ---------------------------------------------------------------- NCHW -> NHWC staging copy
__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float *__restrict__ src, int C, int HW, float *__restrict__ dst,
                                                      int32_t *__restrict__ nan_flag)
{
    __shared__ float tile[64][65];
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    src += (int64_t)b * C * HW;
    dst += (int64_t)b * C * HW;
    bool nan = false;
    for (int k = ty; k < 64; k += 4) {
        int c = c0 + k, p = p0 + tx;
        float v = (c < C && p < HW) ? src[(int64_t)c * HW + p] : 0.f;
        nan |= (v != v);
        tile[k][tx] = v;
    }
    if (nan_flag && __ballot(nan) != 0 && tx == 0) atomicOr(nan_flag, 1);   // lets the pooling kernel skip NaN handling
    __syncthreads();
    for (int k = ty; k < 64; k += 4) {
        int p = p0 + k, c = c0 + tx;
        if (c < C && p < HW) dst[(int64_t)p * C + c] = tile[tx][k];
    }
}

// ---------------------------------------------------------------- forward
// Max update in ATen's order (AdaptiveMaxPool2d CPU kernel): first maximum wins, a NaN always wins.
template <bool NANSAFE>
__device__ __forceinline__ void max_upd(float &best, int &bi, float v, int idx)
{
    if (NANSAFE) { if ((v > best) || (v != v)) { best = v; bi = idx; } }
    else         { const bool g = v > best; bi = g ? idx : bi; best = g ? v : best; }   // strict: -0 is not replaced by a later +0
}

// The pooled output is written once and not read again by this kernel: non-temporal 16-byte stores keep it
// from pushing the feature-map slice out of the XCD's L2 (with plain stores 40 % of the crop reads missed).
typedef float dg_v4f __attribute__((ext_vector_type(4)));
typedef int dg_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store4(float *p, float4 v)
{
    dg_v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<dg_v4f *>(p));
}
__device__ __forceinline__ void nt_store4(int32_t *p, int4 v)
{
    dg_v4i t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<dg_v4i *>(p));
}

// v_max_f32 / v_max3_f32 as such: fmaxf would first canonicalise operands that come straight from memory
__device__ __forceinline__ float vmaxf(float a, float b) { float o; asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b)); return o; }
__device__ __forceinline__ float vmax3f(float a, float b, float c) { float o; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c)); return o; }
// min against a wave-uniform limit (+inf keeps `a`, -inf resets it): a full-rate select with no VCC
__device__ __forceinline__ float vmin_uniform(float a, float lim) { float o; asm("v_min_f32 %0, %1, %2" : "=v"(o) : "s"(lim), "v"(a)); return o; }

#ifdef DG_STAMPS
// diagnostic build only (tools/ubench/roi_fwd_stamps.hip): where a task spends its cycles
__device__ unsigned long long dg_stamp_acc[8];
#define STAMP(var) unsigned long long var = __builtin_readcyclecounter()
#define STAMP_ADD(slot, a, b) stamp_loc[slot] += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(slot, a, b)
#endif

#define FWD_TASKS_PER_WG 8               // large launches: block-strided tasks per single-wave workgroup (see roipool_fwd)
#define RING_PX 64                       // crop pixels (x 64 channels x 4 B = 16 KiB) resident in LDS per wave
#define FWD_LDS_FLOATS (64 * RBINS + 64 * RBINS / 2)   // [64][49] f32 values + [64][49] u16 indices = 18,816 B

// Staged task: the crop's rows stream into an LDS ring by LDS-DMA (global_load_lds, 16 B per lane = 4 pixels x
// 64 channels per wave instruction); every crop row is fetched once and the next strip's rows are in flight
// while the current strip is pooled out of LDS.  For each row the 7 column bins are scanned together, XW
// pixels each (positions past a bin's end are clamped re-reads of its last pixel: idempotent), from
// per-(bin,j) LDS addresses computed once per task.  All 49 results stay in registers (the 7 strips are
// unrolled), so the [channel][bin] transpose buffer can alias the ring: 18.8 KB of LDS per wave, 8 waves/CU.
template <int XW, bool NANSAFE>
__device__ __forceinline__ void roi_task_staged(char *smem, const char *__restrict__ imgc0, int pixbytes, int W, int lane,
                                                int x1, int y1, int x2, int y2, int R,
                                                float (&oval)[RBINS], int (&oidx)[RBINS]
#ifdef DG_STAMPS
                                                , unsigned long long *stamp_loc
#endif
                                                )
{
    const int Lh = y2 - y1, Lw = x2 - x1, Lwp = (Lw + 3) & ~3;
    const int rowbytes = Lwp * 256;
    // adaptive column-bin edges (uniform -> scalar registers) and the byte offset of (bin, j) inside a ring row
    int ws[RP], we[RP], soff[RP][XW];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        ws[i] = (i * Lw) / RP;                          // relative to x1
        we[i] = ((i + 1) * Lw + RP - 1) / RP;
#pragma unroll
        for (int j = 0; j < XW; ++j) soff[i][j] = min(ws[i] + j, we[i] - 1) * 256 + lane * 4;   // per-lane (VGPR)
    }
    // LDS-DMA source offsets: this lane moves channel quad (lane & 15) of pixel (lane >> 4) of each group of 4 pixels
    const int pxl = lane >> 4;
    unsigned voff[6];                                   // Lwp <= 24
#pragma unroll
    for (int q = 0; q < 6; ++q) voff[q] = (unsigned)(min(4 * q + pxl, Lw - 1) * pixbytes + (lane & 15) * 16);
    const int nq = Lwp >> 2;
    int loaded = y1, lslot = 0;
    auto issue_rows = [&](int ylimit) {
        while (loaded < ylimit) {
            const char *rowg = imgc0 + (int64_t)(loaded * W + x1) * pixbytes;      // uniform
            char *rowl = smem + lslot * rowbytes;
#pragma unroll
            for (int q = 0; q < 6; ++q)
                if (q < nq)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rowg + voff[q]),
                                                     (__attribute__((address_space(3))) void *)(rowl + q * 1024), 16, 0, 0);
            lslot = (lslot + 1 == R) ? 0 : lslot + 1;
            ++loaded;
        }
    };
    STAMP(tp0);
    issue_rows(min(y2, y1 + R));
    STAMP(tp1);
    STAMP_ADD(0, tp0, tp1);
    int slot = 0, prev_hs = y1;
#pragma unroll
    for (int ph = 0; ph < RP; ++ph) {
        const int hs = y1 + (ph * Lh) / RP, he = y1 + ((ph + 1) * Lh + RP - 1) / RP;
        STAMP(t0);
        issue_rows(he);                                    // normally already issued
        STAMP(t1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // rows < he have landed
        STAMP(t2);
        issue_rows(min(y2, hs + R));                       // prefetch; only overwrites rows < hs
        STAMP(t3);
        STAMP_ADD(0, t0, t1); STAMP_ADD(1, t1, t2); STAMP_ADD(0, t2, t3);
        slot += hs - prev_hs;                              // hs advances by <= maxrows <= R
        if (slot >= R) slot -= R;
        prev_hs = hs;
        float best[RP];
        int code[RP];
#pragma unroll
        for (int i = 0; i < RP; ++i) { best[i] = -INFINITY; code[i] = 0; }
        int sl = slot, so = slot * rowbytes;
        // code of the running maximum.  NaN-free maps: r*W + j (a clamped re-read can never win a strict >, so the
        // index is base + code); NaN-exact: r*8 + j, decoded with the clamp.
        int rcode = 0;
        for (int r = 0; r < he - hs; ++r) {
#pragma unroll
            for (int j = 0; j < XW; ++j) {
                const int cj = rcode + j;
#pragma unroll
                for (int i = 0; i < RP; ++i)
                    max_upd<NANSAFE>(best[i], code[i], *(const float *)(smem + (soff[i][j] + so)), cj);
            }
            rcode += NANSAFE ? 8 : W;
            so += rowbytes;
            if (++sl == R) { sl = 0; so = 0; }
        }
        STAMP(t4);
        const int ibase = hs * W + x1;
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            oval[ph * RP + i] = best[i];
            if (NANSAFE) oidx[ph * RP + i] = ibase + (code[i] >> 3) * W + min(ws[i] + (code[i] & 7), we[i] - 1);
            else oidx[ph * RP + i] = ibase + ws[i] + code[i];
        }
        // every ring read of this strip has returned before the next strip may issue LDS-DMA over these rows (issue
        // order alone does not keep a DMA write behind a queued ds_read; see roi_task_exact)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        STAMP(t5);
        STAMP_ADD(2, t3, t4); STAMP_ADD(3, t4, t5);
    }
}

// Staged task for NaN-free maps, specialised on the crop width LW (1..24): the adaptive column bins are
// compile-time constants, so every LDS read is `row base + immediate`, no position is read twice for one bin
// and no per-lane address table exists.  Per (row, bin): the row's maximum over the bin's pixels (v_max3),
// then one strict compare against the running maximum, which records the row -- the first row holding the
// maximum wins, as in ATen.  The column is recovered once per bin at the end of the strip: the first pixel of
// the winning row that equals the maximum (the row is still in the ring: the prefetch issued at the start of
// the strip only overwrote rows above it).  The value written is that pixel's own bits (-0 stays -0).
// SHARE: adjacent strips share their boundary row unless ph*Lh is a multiple of 7 (adaptive bins: floor / ceil edges), so
// the per-bin maxima of the row scanned last seed the next strip as its row 0 and that row is not read again: Lh row
// scans per task instead of up to Lh + 6.  Used by the uint16-argmax kernel, whose compute side is what bounds it (17.2 ->
// 16.1 ms); the int32 kernel sits on the device's store rate and ran 2 % slower with it (20.3 -> 20.7 ms), so it keeps the
// plain strips.
template <int LW, bool SHARE>
__device__ __forceinline__ void roi_task_exact(char *smem, const char *__restrict__ imgc0, int pixbytes, int W, int lane,
                                               int x1, int y1, int y2, int R, float (&oval)[RBINS], int (&oidx)[RBINS]
#ifdef DG_STAMPS
                                               , unsigned long long *stamp_loc
#endif
                                               )
{
    constexpr int Lwp = (LW + 3) & ~3, NQ = Lwp >> 2, rowbytes = Lwp * 256;
    const int Lh = y2 - y1;
    const int pxl = lane >> 4;
    unsigned voff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) voff[q] = (unsigned)(min(4 * q + pxl, LW - 1) * pixbytes + (lane & 15) * 16);
    int loaded = y1, lslot = 0;
    auto issue_rows = [&](int ylimit) {
        while (loaded < ylimit) {
            const char *rowg = imgc0 + (int64_t)(loaded * W + x1) * pixbytes;      // uniform
            char *rowl = smem + lslot * rowbytes;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rowg + voff[q]),
                                                 (__attribute__((address_space(3))) void *)(rowl + q * 1024), 16, 0, 0);
            lslot = (lslot + 1 == R) ? 0 : lslot + 1;
            ++loaded;
        }
    };
    STAMP(tp0);
    issue_rows(min(y2, y1 + R));
    STAMP(tp1);
    STAMP_ADD(0, tp0, tp1);
    const char *lbase = smem + lane * 4;
    int slot = 0, prev_hs = y1;
    float rm[RP];                                          // per-bin maxima of the row scanned last (SHARE: live across strips)
#pragma unroll
    for (int i = 0; i < RP; ++i) rm[i] = -INFINITY;
#pragma unroll
    for (int ph = 0; ph < RP; ++ph) {
        const int hs = y1 + (ph * Lh) / RP, he = y1 + ((ph + 1) * Lh + RP - 1) / RP;
        STAMP(t0);
        issue_rows(he);                                    // normally already issued
        STAMP(t1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // rows < he have landed
        STAMP(t2);
        issue_rows(min(y2, hs + R));                       // prefetch; only overwrites rows < hs
        STAMP(t3);
        STAMP_ADD(0, t0, t1); STAMP_ADD(1, t1, t2); STAMP_ADD(0, t2, t3);
        slot += hs - prev_hs;                              // hs advances by <= maxrows <= R
        if (slot >= R) slot -= R;
        prev_hs = hs;
        // shared: this strip's first row is the previous strip's last one (wave-uniform); strip 0 never shares
        const int shared = (SHARE && ph > 0 && ((ph * Lh) % RP) != 0) ? 1 : 0;
        const float lim = shared ? INFINITY : -INFINITY;
        float best[RP];
        int brow[RP];
#pragma unroll
        for (int i = 0; i < RP; ++i) { best[i] = (SHARE && ph > 0) ? vmin_uniform(rm[i], lim) : -INFINITY; brow[i] = 0; }
        int sl = slot + shared;
        if (SHARE && sl >= R) sl -= R;
        int so = sl * rowbytes;
        for (int r = shared; r < he - hs; ++r) {
            const char *rowp = lbase + so;
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                const int s = (i * LW) / RP, w = ((i + 1) * LW + RP - 1) / RP - s;      // folded: i and LW are constants
                rm[i] = *(const float *)(rowp + s * 256);
                int j = 1;
#pragma unroll
                for (; j + 1 < w; j += 2)
                    rm[i] = vmax3f(rm[i], *(const float *)(rowp + (s + j) * 256), *(const float *)(rowp + (s + j + 1) * 256));
                if (j < w) rm[i] = vmaxf(rm[i], *(const float *)(rowp + (s + j) * 256));
                brow[i] = (rm[i] > best[i]) ? r : brow[i];
                best[i] = vmaxf(best[i], rm[i]);
            }
            so += rowbytes;
            if (++sl == R) { sl = 0; so = 0; }
        }
        STAMP(t4);
        const int ibase = hs * W + x1;
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const int s = (i * LW) / RP, w = ((i + 1) * LW + RP - 1) / RP - s;
            int rs = slot + brow[i];                       // ring slot of the winning row (per lane)
            rs = (rs >= R) ? rs - R : rs;
            const char *rowp = lbase + rs * rowbytes;
#if DG_FWD_RECOVER_V2
            // first cell of the winning row that equals the maximum: the last one is implied, a one-cell bin needs nothing.
            // The value is the running maximum itself; only a maximum of zero can differ from the winning cell's bits
            // (max(-0,+0)), and the caller repairs those after the task (rare, outside the strips).
            int c = w - 1;
#pragma unroll
            for (int j = w - 2; j >= 0; --j) {
                const float a = *(const float *)(rowp + (s + j) * 256);
                c = (a == best[i]) ? j : c;
            }
            oval[ph * RP + i] = best[i];
#else
            int c = 0;
            float bv = best[i];
#pragma unroll
            for (int j = w - 1; j >= 0; --j) {
                const float a = *(const float *)(rowp + (s + j) * 256);
                const bool hit = (a == best[i]);
                c = hit ? j : c;
                bv = hit ? a : bv;
            }
            oval[ph * RP + i] = bv;
#endif
            oidx[ph * RP + i] = ibase + brow[i] * W + s + c;
        }
        // The reads above must have returned before the next strip issues LDS-DMA into the rows they came from:
        // issue order alone does not keep a DMA write behind a ds_read that is still queued.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        STAMP(t5);
        STAMP_ADD(2, t3, t4); STAMP_ADD(3, t4, t5);
    }
}

// Staged task for avg mode (the reference class as written: AdaptiveAvgPool2d, rcnn/fast_rcnn.py:354), specialised
// on the crop width like roi_task_exact: same LDS-DMA row ring, per (row, bin) the sum of the bin's pixels, bin sums
// accumulated over the strip's rows, result = sum / rows / columns (the direct path's order).
template <int LW>
__device__ __forceinline__ void roi_task_avg(char *smem, const char *__restrict__ imgc0, int pixbytes, int W, int lane,
                                             int x1, int y1, int y2, int R, float (&oval)[RBINS])
{
    constexpr int Lwp = (LW + 3) & ~3, NQ = Lwp >> 2, rowbytes = Lwp * 256;
    const int Lh = y2 - y1;
    const int pxl = lane >> 4;
    unsigned voff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) voff[q] = (unsigned)(min(4 * q + pxl, LW - 1) * pixbytes + (lane & 15) * 16);
    int loaded = y1, lslot = 0;
    auto issue_rows = [&](int ylimit) {
        while (loaded < ylimit) {
            const char *rowg = imgc0 + (int64_t)(loaded * W + x1) * pixbytes;      // uniform
            char *rowl = smem + lslot * rowbytes;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rowg + voff[q]),
                                                 (__attribute__((address_space(3))) void *)(rowl + q * 1024), 16, 0, 0);
            lslot = (lslot + 1 == R) ? 0 : lslot + 1;
            ++loaded;
        }
    };
    issue_rows(min(y2, y1 + R));
    const char *lbase = smem + lane * 4;
    int slot = 0, prev_hs = y1;
#pragma unroll
    for (int ph = 0; ph < RP; ++ph) {
        const int hs = y1 + (ph * Lh) / RP, he = y1 + ((ph + 1) * Lh + RP - 1) / RP;
        issue_rows(he);                                    // normally already issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // rows < he have landed
        issue_rows(min(y2, hs + R));                       // prefetch; only overwrites rows < hs
        slot += hs - prev_hs;
        if (slot >= R) slot -= R;
        prev_hs = hs;
        float acc[RP];
#pragma unroll
        for (int i = 0; i < RP; ++i) acc[i] = 0.f;
        int sl = slot, so = slot * rowbytes;
        for (int r = 0; r < he - hs; ++r) {
            const char *rowp = lbase + so;
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                const int s = (i * LW) / RP, w = ((i + 1) * LW + RP - 1) / RP - s;      // folded: i and LW are constants
                float rs = *(const float *)(rowp + s * 256);
#pragma unroll
                for (int j = 1; j < w; ++j) rs += *(const float *)(rowp + (s + j) * 256);
                acc[i] += rs;
            }
            so += rowbytes;
            if (++sl == R) { sl = 0; so = 0; }
        }
        const float rows = (float)(he - hs);
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const int s = (i * LW) / RP, w = ((i + 1) * LW + RP - 1) / RP - s;
            oval[ph * RP + i] = acc[i] / rows / (float)w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // ring reads returned before the next strip's DMA (see roi_task_exact)
    }
}

// One adaptive max bin straight from the map (any crop size): lane = channel, `base` points at this lane's channel of
// the image.  First maximum in row-major order; with NANSAFE a NaN always wins (ATen).  Index = y*W + x.
template <bool NANSAFE>
__device__ __forceinline__ void direct_max_bin(const float *__restrict__ base, int W, int C, int hs, int he, int ws, int we,
                                               float &best, int &bi)
{
    best = -INFINITY;
    bi = hs * W + ws;
    if (we - ws > 4) {
        // wide bins (crops over ~28 columns): 8 loads in flight per batch
        for (int y = hs; y < he; ++y) {
            const float *row = base + (int64_t)y * W * C;
            for (int x = ws; x < we; x += 8) {
                int xs[8];
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) xs[q] = min(x + q, we - 1);      // past the end: clamped re-reads
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = row[(int64_t)xs[q] * C];
#pragma unroll
                for (int q = 0; q < 8; ++q) max_upd<NANSAFE>(best, bi, v[q], y * W + xs[q]);
            }
        }
    } else {
        // bins of at most 4 columns: two rows (8 loads) in flight; past the bin's end or the last
        // row: clamped re-reads (idempotent: a repeat can never win the strict compare)
        const int xb = min(ws + 1, we - 1), xc = min(ws + 2, we - 1), xd = min(ws + 3, we - 1);
        for (int y = hs; y < he; y += 2) {
            const int y2r = min(y + 1, he - 1);
            const float *row = base + (int64_t)y * W * C, *row2 = base + (int64_t)y2r * W * C;
            const float va = row[(int64_t)ws * C], vb = row[(int64_t)xb * C];
            const float vc = row[(int64_t)xc * C], vd = row[(int64_t)xd * C];
            const float ua = row2[(int64_t)ws * C], ub = row2[(int64_t)xb * C];
            const float uc = row2[(int64_t)xc * C], ud = row2[(int64_t)xd * C];
            max_upd<NANSAFE>(best, bi, va, y * W + ws);
            max_upd<NANSAFE>(best, bi, vb, y * W + xb);
            max_upd<NANSAFE>(best, bi, vc, y * W + xc);
            max_upd<NANSAFE>(best, bi, vd, y * W + xd);
            max_upd<NANSAFE>(best, bi, ua, y2r * W + ws);
            max_upd<NANSAFE>(best, bi, ub, y2r * W + xb);
            max_upd<NANSAFE>(best, bi, uc, y2r * W + xc);
            max_upd<NANSAFE>(best, bi, ud, y2r * W + xd);
        }
    }
}

// Wide crops (more than 24 cells wide, or strips taller than the LDS ring holds): one-pass exact argmax straight from the
// map with NO per-cell address arithmetic.  All bin geometry is wave-uniform, so per task each lane keeps 7 x XW byte
// offsets (column of position j of bin i, clamped to the bin, + this lane's channel) and every load is
// `global_load_dword v, voff[i][j], s[row base]`; a row of the crop is 7 x XW coalesced 256-byte loads held in
// registers.  Per cell: one compare, one select of the position code (r*8 + j, kept in XW registers that advance by 8 per
// row), one select of the value (its own bits: -0 stays -0).  A clamped re-read repeats the bin's last cell and cannot win
// the strict compare; with NANSAFE a NaN always wins (ATen) and the decode clamps.  Results go straight into the
// [channel][bin] LDS image.
// Round 3: the rows of the whole crop are ONE stream through NB register buffers -- NB - 1 rows are in flight while one is
// compared, and the stream does not stop at strip boundaries (a row that two strips share is compared twice from the same
// registers).  Round 2's version kept one row in flight and restarted at each of the seven strips: a task of 26 row steps
// waited out an L2 round trip at nearly every one of them (8 waves per CU cannot hide it), which is what held the
// U(0.1,0.9) proposal mix at 0.24 of the store-bound rate.
// (The direct path this replaces spent ~10 VALU instructions per cell on 64-bit address and index arithmetic.)
// What bounds it now is the compare + select per cell (v_cmp / v_cndmask are half rate: 20 cycles per cell and wave with the
// v_max; in-kernel stamps put 60 % of the U(0.1,0.9) mix's time in this scan, 7 % in the output phase).  Measured and not kept:
// cells only maxed (v_max3), the winning ROW recorded per bin and row, the column found after the last strip by loading
// the winning row's cells again -- each lane has its own row then, so those loads touch 64 lines instead of 2 and the
// mix ran 9.5 ms instead of 6.1.
template <int XW, bool NANSAFE, int NB>
__device__ __forceinline__ void roi_task_wide(const char *__restrict__ imgc0, int pixbytes, int W, int lane, int x1, int y1, int x2,
                                              int y2, float *__restrict__ sval, unsigned short *__restrict__ sidx)
{
    const int Lh = y2 - y1, Lw = x2 - x1;
    int ws[RP], we[RP];
    // XW >= 5 is the widest bin itself and every bin is at most two cells narrower: positions 0 .. XW-3 exist in all seven
    // bins, so their offsets (j pixels + this lane's channel) are shared and the bin's first column goes into the scalar
    // base; only the last two positions are clamped per bin.  XW + 12 offset registers instead of 7 XW -- what pays for the
    // third row buffer.  (XW == 4 also serves narrower bins and keeps the full table.)
    constexpr bool SHARED = XW >= 5;
    constexpr int NT = SHARED ? 1 : RP;
    unsigned voff[NT][XW], voffA[RP], voffB[RP];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        ws[i] = (i * Lw) / RP;                          // relative to x1 (uniform)
        we[i] = ((i + 1) * Lw + RP - 1) / RP;
        const int w = we[i] - ws[i];
        voffA[i] = (unsigned)(min(XW - 2, w - 1) * pixbytes + lane * 4);
        voffB[i] = (unsigned)(min(XW - 1, w - 1) * pixbytes + lane * 4);
#pragma unroll
        for (int j = 0; j < XW; ++j)
            if (i < NT) voff[i < NT ? i : 0][j] = (unsigned)((SHARED ? j : min(ws[i] + j, we[i] - 1)) * pixbytes + lane * 4);
    }
    auto load_row = [&](int y, float (&v)[RP][XW]) {
        const char *rowg = imgc0 + (int64_t)(y * W + x1) * pixbytes;                   // uniform
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const char *bing = SHARED ? rowg + ws[i] * pixbytes : rowg;                // uniform
#pragma unroll
            for (int j = 0; j < XW; ++j) {
                const unsigned o = !SHARED ? voff[i < NT ? i : 0][j] : (j == XW - 2 ? voffA[i] : (j == XW - 1 ? voffB[i] : voff[0][j]));
                v[i][j] = *reinterpret_cast<const float *>(bing + o);
            }
        }
    };
    // the strip being filled: rows hs .. he - 1 of the map (uniform)
    int ph = 0, hs = y1, he = y1 + (Lh + RP - 1) / RP;
    float best[RP];
    int code[RP], vcode[XW];
    auto reset = [&]() {
#pragma unroll
        for (int i = 0; i < RP; ++i) { best[i] = -INFINITY; code[i] = 0; }
#pragma unroll
        for (int j = 0; j < XW; ++j) vcode[j] = j;
    };
    auto scan_row = [&](const float (&v)[RP][XW]) {
#pragma unroll
        for (int j = 0; j < XW; ++j) {
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                const float a = v[i][j];
                if (NANSAFE) {
                    const bool g = (a > best[i]) || (a != a);
                    code[i] = g ? vcode[j] : code[i];
                    best[i] = g ? a : best[i];
                } else {
                    // value by v_max (full rate; v_cmp / v_cndmask are half rate): equal to the selected value except
                    // that max(-0, +0) = +0 while the strict compare keeps the earlier -0 -- repaired when the strip ends
                    code[i] = (a > best[i]) ? vcode[j] : code[i];
                    best[i] = vmaxf(best[i], a);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < XW; ++j) vcode[j] += 8;
    };
    auto finish_strip = [&]() {
        const int ibase = hs * W + x1;
        int pos[RP];
#pragma unroll
        for (int i = 0; i < RP; ++i) pos[i] = ibase + (code[i] >> 3) * W + min(ws[i] + (code[i] & 7), we[i] - 1);
        if (!NANSAFE) {
            // a maximum of zero may carry the wrong sign (see scan_row): take the winning cell's own bits (rare, uniform branch)
            bool anyz = false;
#pragma unroll
            for (int i = 0; i < RP; ++i) anyz = anyz || (best[i] == 0.0f);
            if (__ballot(anyz) != 0ull) {
#pragma unroll
                for (int i = 0; i < RP; ++i)
                    if (best[i] == 0.0f)
                        best[i] = *reinterpret_cast<const float *>(imgc0 + (int64_t)pos[i] * pixbytes + lane * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            sval[lane * RBINS + ph * RP + i] = best[i];
            sidx[lane * RBINS + ph * RP + i] = (unsigned short)pos[i];
        }
    };
    reset();
    float buf[NB][RP][XW];                                // used in rotation; the row loop is unrolled by NB so that none is ever copied
#pragma unroll
    for (int b = 0; b + 1 < NB; ++b) load_row(min(y1 + b, y2 - 1), buf[b]);
#pragma unroll 1
    for (int y = y1; y < y2; y += NB) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int yy = y + b;
            if (yy < y2) {                                  // uniform
                load_row(min(yy + NB - 1, y2 - 1), buf[(b + NB - 1) % NB]);   // past the crop: a re-read whose result is not used
                scan_row(buf[b]);
                // strips that end with this row (short crops: several may); the next strip starts on this row or the next
                // (floor((ph+1) Lh / 7) >= ceil((ph+1) Lh / 7) - 1), never earlier
                while (ph < RP && yy == he - 1) {
                    finish_strip();
                    ++ph;
                    if (ph == RP) break;
                    hs = y1 + (ph * Lh) / RP; he = y1 + ((ph + 1) * Lh + RP - 1) / RP;
                    reset();
                    if (hs > yy) break;
                    scan_row(buf[b]);
                }
            }
        }
    }
}

// Which code serves a max-mode task: 0 = LDS-ring strips (roi_task_exact / roi_task_staged), 1 = register-streamed rows
// (roi_task_wide, run by the kernel's PART 1 instantiation), 2 = the direct path.  One definition for both instantiations.
__device__ __forceinline__ int fwd_max_task_class(int Lw, int Lh, bool may_nan, int cvalid, int C, bool aligned16, bool small_image)
{
    const int Lwp = (Lw + 3) & ~3;
    const int R = RING_PX / Lwp;
    // widest / tallest adaptive bin of a length L = 7q + r: q (r == 0), q + 1 (r == 1), q + 2 (r >= 2)
    const int qw = Lw / RP, rw = Lw - qw * RP, qh = Lh / RP, rh = Lh - qh * RP;
    const int maxw = qw + (rw > 1 ? 2 : rw), maxrows = qh + (rh > 1 ? 2 : rh);
    const bool staged = (R >= maxrows) && (Lwp <= 24) && (maxw <= 4 || (!may_nan && Lw <= 16)) && (cvalid == 64) && ((C & 3) == 0) &&
                        aligned16 && small_image;
    if (staged) return 0;
    if (DG_FWD_WIDE && cvalid == 64 && small_image && maxw <= 8) return 1;
    return 2;
}

// A16: the argmax goes out as uint16 (argmax16) instead of int32 (argmax).  A compile-time choice: with both store
// sequences in one kernel the int32 path ran 0.6-0.8 ms slower (the scheduler's interleaving of the output phase moved).
// PART 0: every task but the class-1 ones (they are skipped); PART 1 (max mode only): the class-1 tasks alone -- a kernel of
// its own, because with roi_task_wide compiled into the main kernel the narrow-crop path ran 3 % slower (19.2 -> 19.9 ms).
template <int MODE, bool A16, int PART>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_roipool_fwd(const float *__restrict__ nhwc, int C, int H, int W,
                                                    const int32_t *__restrict__ win, int cblocks, int64_t ntasks,
                                                    const int32_t *__restrict__ nan_flag,
                                                    float *__restrict__ out, int32_t *__restrict__ argmax,
                                                    unsigned short *__restrict__ argmax16, int vec_ok)
{
    // vec_ok: bit 0 = 16-byte stores of out / argmax, bit 1 = 16-byte stores of argmax16
    __shared__ __attribute__((aligned(16))) float smemf[FWD_LDS_FLOATS];
    float *sval = smemf;
    unsigned short *sidx = reinterpret_cast<unsigned short *>(smemf + 64 * RBINS);
    const int lane0 = threadIdx.x;
    const bool may_nan = nan_flag ? (__builtin_amdgcn_readfirstlane(nan_flag[0]) != 0) : true;
#ifdef DG_STAMPS
    unsigned long long stamp_loc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    STAMP(tk0);
#endif
    // Persistent single-wave workgroups.  Task = (RoI, 64-channel block), channel block fastest; the grid is a
    // multiple of 8, so a workgroup (and the XCD it sits on, blockIdx % 8) keeps one channel slice of the map.
    const bool aligned16 = (((uintptr_t)nhwc) & 15) == 0, small_image = (int64_t)H * W * C * 4 < (1ll << 31);
    auto run_task = [&](int64_t task) {
        // opaque per task: otherwise every lane-derived LDS/global address of the loop body (about 100 of them) is
        // hoisted out of the task loop and held in registers for the kernel's lifetime
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int cbk = (int)(task % cblocks);
        const int64_t roi = task / cblocks;
        const int c0 = cbk * 64;
        const int32_t *w = win + 5 * roi;
        const int x1 = __builtin_amdgcn_readfirstlane(w[0]), y1 = __builtin_amdgcn_readfirstlane(w[1]);
        const int x2 = __builtin_amdgcn_readfirstlane(w[2]), y2 = __builtin_amdgcn_readfirstlane(w[3]);
        const int b = __builtin_amdgcn_readfirstlane(w[4]);
        const int Lh = y2 - y1, Lw = x2 - x1;
        const int cvalid = (C - c0) < 64 ? (C - c0) : 64;

        STAMP(ta);
        // output phase; instantiated at the end of each path (staged / direct) so that the two paths do not merge
        // with 98 result registers apparently live (the allocator then carries them around the whole task loop)
        auto emit = [&]() {
        STAMP(te0);
        __syncthreads();
        // contiguous run of cvalid*49 outputs for this (roi, channel block)
        const int64_t obase = (roi * C + c0) * RBINS;
        const int total = cvalid * RBINS;
        if (vec_ok & 1) {
            // 64*49/4 = 784 16-byte pieces = 12.25 wave-wide rounds; reads of a batch are issued before its stores
            if (cvalid == 64) {
                const float *sv = sval + lane * 4;
                const unsigned short *si = sidx + lane * 4;
                float *go = out + obase + lane * 4;
                int32_t *ga = argmax + obase + lane * 4;
#pragma unroll
                for (int h = 0; h < 12; h += 6) {
                    float4 v[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) v[k] = *reinterpret_cast<const float4 *>(sv + (h + k) * 256);
#pragma unroll
                    for (int k = 0; k < 6; ++k) nt_store4(go + (h + k) * 256, v[k]);
                }
                if (lane < 16) nt_store4(go + 12 * 256, *reinterpret_cast<const float4 *>(sv + 12 * 256));
                if (MODE == DG_POOL_MAX && !A16 && argmax) {
#pragma unroll
                    for (int h = 0; h < 12; h += 6) {
                        ushort4 u[6];
#pragma unroll
                        for (int k = 0; k < 6; ++k) u[k] = *reinterpret_cast<const ushort4 *>(si + (h + k) * 256);
#pragma unroll
                        for (int k = 0; k < 6; ++k) nt_store4(ga + (h + k) * 256, make_int4(u[k].x, u[k].y, u[k].z, u[k].w));
                    }
                    if (lane < 16) {
                        const ushort4 u = *reinterpret_cast<const ushort4 *>(si + 12 * 256);
                        nt_store4(ga + 12 * 256, make_int4(u.x, u.y, u.z, u.w));
                    }
                }
            } else {
                for (int e = lane * 4; e < total; e += 256) {
                    *reinterpret_cast<float4 *>(out + obase + e) = *reinterpret_cast<const float4 *>(sval + e);
                    if (MODE == DG_POOL_MAX && !A16 && argmax) {
                        const ushort4 u = *reinterpret_cast<const ushort4 *>(sidx + e);
                        *reinterpret_cast<int4 *>(argmax + obase + e) = make_int4(u.x, u.y, u.z, u.w);
                    }
                }
            }
        } else {
            for (int e = lane; e < total; e += 64) {
                out[obase + e] = sval[e];
                if (MODE == DG_POOL_MAX && !A16 && argmax) argmax[obase + e] = sidx[e];
            }
        }
        // 16-bit argmax (H*W <= 65535): the LDS image goes out as it is, 64*49/8 = 392 16-byte pieces = 6.125 rounds
        if (MODE == DG_POOL_MAX && A16 && argmax16) {
            if ((vec_ok & 2) && cvalid == 64) {
                const unsigned short *si = sidx + lane * 8;
                unsigned short *ga = argmax16 + obase + lane * 8;
                int4 u[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) u[k] = *reinterpret_cast<const int4 *>(si + k * 512);
#pragma unroll
                for (int k = 0; k < 6; ++k) nt_store4(reinterpret_cast<int32_t *>(ga + k * 512), u[k]);
                if (lane < 8) nt_store4(reinterpret_cast<int32_t *>(ga + 6 * 512), *reinterpret_cast<const int4 *>(si + 6 * 512));
            } else {
                for (int e = lane; e < total; e += 64) argmax16[obase + e] = sidx[e];
            }
        }
        __syncthreads();      // the buffer is reused by the next task's ring
        STAMP(te1);
        STAMP_ADD(5, te0, te1);
        };
        bool staged = false;
        const int kind = (MODE == DG_POOL_MAX) ? fwd_max_task_class(Lw, Lh, may_nan, cvalid, C, aligned16, small_image) : 2;
        if (PART == 0 && kind == 1) return;               // the PART 1 launch does this task
        if (MODE == DG_POOL_MAX && PART == 0) {
            const int Lwp = (Lw + 3) & ~3;
            const int R = RING_PX / Lwp;
            const int qw = Lw / RP, rw = Lw - qw * RP;
            const int maxw = qw + (rw > 1 ? 2 : rw);
            staged = kind == 0;
            if (staged) {
                float oval[RBINS];
                int oidx[RBINS];
                char *smem = reinterpret_cast<char *>(smemf);
#ifdef DG_STAMPS
#define STAMP_ARG , stamp_loc
#else
#define STAMP_ARG
#endif
                const char *imgc0 = reinterpret_cast<const char *>(nhwc + ((int64_t)b * H * W) * C + c0);
#define EXACT(LWV) case LWV: roi_task_exact<LWV, A16>(smem, imgc0, C * 4, W, lane, x1, y1, y2, R, oval, oidx STAMP_ARG); break
#define STAGED(XWV) roi_task_staged<XWV, true>(smem, imgc0, C * 4, W, lane, x1, y1, x2, y2, R, oval, oidx STAMP_ARG)
                if (!may_nan) {
                    switch (Lw) {
                    EXACT(1); EXACT(2); EXACT(3); EXACT(4); EXACT(5); EXACT(6); EXACT(7); EXACT(8);
                    EXACT(9); EXACT(10); EXACT(11); EXACT(12); EXACT(13); EXACT(14); EXACT(15); EXACT(16);
                    default:                           // 17..24 wide: generic code, 3 or 4 positions per bin
                        if (maxw <= 3) roi_task_staged<3, false>(smem, imgc0, C * 4, W, lane, x1, y1, x2, y2, R, oval, oidx STAMP_ARG);
                        else roi_task_staged<4, false>(smem, imgc0, C * 4, W, lane, x1, y1, x2, y2, R, oval, oidx STAMP_ARG);
                        break;
                    }
                } else {
                    switch (maxw) {
                    case 1: STAGED(1); break;
                    case 2: STAGED(2); break;
                    case 3: STAGED(3); break;
                    default: STAGED(4); break;
                    }
                }
#undef STAGED
#undef EXACT
                STAMP(tw0);
#if DG_FWD_RECOVER_V2
                if (!may_nan) {
                    // signed zeros: a zero maximum takes the winning cell's own bits (min over |values| by v_min3: 24 ops)
                    float mz = fabsf(oval[0]);
#pragma unroll
                    for (int i = 1; i < RBINS; i += 2) mz = fminf(mz, fminf(fabsf(oval[i]), fabsf(oval[i + 1])));
                    if (__ballot(mz == 0.0f) != 0ull) {
#pragma unroll
                        for (int i = 0; i < RBINS; ++i)
                            if (oval[i] == 0.0f) oval[i] = *reinterpret_cast<const float *>(imgc0 + (int64_t)oidx[i] * (C * 4) + lane * 4);
                    }
                }
#endif
                // all ring reads are done (results are in registers): the transpose buffer may overwrite the ring
#pragma unroll
                for (int i = 0; i < RBINS; ++i) {
                    sval[lane * RBINS + i] = oval[i];
                    sidx[lane * RBINS + i] = (unsigned short)oidx[i];
                }
                STAMP(tw1);
                STAMP_ADD(7, tw0, tw1);
                emit();
            }
        }
        if (MODE == DG_POOL_AVG && PART == 0) {
            const int Lwp = (Lw + 3) & ~3;
            const int R = RING_PX / Lwp;
            const int qh = Lh / RP, rh = Lh - qh * RP;
            const int maxrows = qh + (rh > 1 ? 2 : rh);
            staged = (R >= maxrows) && (Lw <= 24) && (cvalid == 64) && ((C & 3) == 0) && ((((uintptr_t)nhwc) & 15) == 0) &&
                     ((int64_t)H * W * C * 4 < (1ll << 31));
            if (staged) {
                float oval[RBINS];
                char *smem = reinterpret_cast<char *>(smemf);
                const char *imgc0 = reinterpret_cast<const char *>(nhwc + ((int64_t)b * H * W) * C + c0);
#define AVGW(LWV) case LWV: roi_task_avg<LWV>(smem, imgc0, C * 4, W, lane, x1, y1, y2, R, oval); break
                switch (Lw) {
                AVGW(1); AVGW(2); AVGW(3); AVGW(4); AVGW(5); AVGW(6); AVGW(7); AVGW(8); AVGW(9); AVGW(10); AVGW(11); AVGW(12);
                AVGW(13); AVGW(14); AVGW(15); AVGW(16); AVGW(17); AVGW(18); AVGW(19); AVGW(20); AVGW(21); AVGW(22); AVGW(23);
                default: roi_task_avg<24>(smem, imgc0, C * 4, W, lane, x1, y1, y2, R, oval); break;
                }
#undef AVGW
#pragma unroll
                for (int i = 0; i < RBINS; ++i) sval[lane * RBINS + i] = oval[i];
                emit();
            }
        }
        if (MODE == DG_POOL_MAX && PART == 1) {
            // ---- wide / tall crops, whole channel blocks: register-streamed rows, see roi_task_wide
            const int qw = Lw / RP, rw = Lw - qw * RP;
            const int maxw = qw + (rw > 1 ? 2 : rw);
            {
                const char *imgc0 = reinterpret_cast<const char *>(nhwc + ((int64_t)b * H * W) * C + c0);
#define WIDE(XWV, NBV) do { if (may_nan) roi_task_wide<XWV, true, NBV>(imgc0, C * 4, W, lane, x1, y1, x2, y2, sval, sidx); \
                            else roi_task_wide<XWV, false, NBV>(imgc0, C * 4, W, lane, x1, y1, x2, y2, sval, sidx); } while (0)
                switch (maxw) {                               // row buffers: as many as the registers hold
                case 1: case 2: case 3: case 4: WIDE(4, 3); break;
                case 5: WIDE(5, 3); break;
                case 6: WIDE(6, 3); break;
                case 7: WIDE(7, 3); break;
                default: WIDE(8, 2); break;
                }
#undef WIDE
                emit();
                staged = true;
            }
        }
        if (PART == 0 && !staged) {
            // ---- direct path (any crop size, partial channel blocks): reads the map from L2/HBM per bin
            const int c = c0 + (lane < cvalid ? lane : cvalid - 1);
            const float *base = nhwc + (int64_t)b * H * W * C + c;
#pragma unroll 1
            for (int ph = 0; ph < RP; ++ph) {
                const int hs = y1 + (ph * Lh) / RP, he = y1 + ((ph + 1) * Lh + RP - 1) / RP;
#pragma unroll 1
                for (int pw = 0; pw < RP; ++pw) {
                    const int ws = x1 + (pw * Lw) / RP, we = x1 + ((pw + 1) * Lw + RP - 1) / RP;
                    if (MODE == DG_POOL_MAX) {
                        float best;
                        int bi;
                        if (may_nan) direct_max_bin<true>(base, W, C, hs, he, ws, we, best, bi);
                        else direct_max_bin<false>(base, W, C, hs, he, ws, we, best, bi);
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
            emit();
        }
        STAMP(tb);
        STAMP_ADD(4, ta, tb);                                 // whole task; slot 5 (output phase) is part of it
    };
    if (PART == 0) {
#pragma unroll 1
        for (int64_t task = blockIdx.x; task < ntasks; task += gridDim.x) run_task(task);
    } else {
        // Class-1 tasks are usually few and far between: 64 tasks are classified at once, one per lane, and only the
        // flagged ones are run (a wave-serial test per task would cost a dependent global load each).  A wave keeps one
        // channel block (and with it its XCD's slice of the maps) when the grid is a multiple of cblocks.
        const bool keep_cb = (gridDim.x % cblocks) == 0;
        const int64_t nrois = ntasks / cblocks;
        const int64_t units = keep_cb ? nrois : ntasks;                       // what a lane indexes: RoIs or tasks
        const int64_t wave0 = keep_cb ? (int64_t)(blockIdx.x / cblocks) : (int64_t)blockIdx.x;
        const int64_t nwaves = keep_cb ? (int64_t)(gridDim.x / cblocks) : (int64_t)gridDim.x;
        const int mycb = keep_cb ? (int)(blockIdx.x % cblocks) : 0;
#pragma unroll 1
        for (int64_t base = wave0 * 64; base < units; base += nwaves * 64) {
            const int64_t u = base + lane0;
            bool flag = false;
            if (u < units) {
                const int64_t roi = keep_cb ? u : u / cblocks;
                const int cbk = keep_cb ? mycb : (int)(u % cblocks);
                const int32_t *w = win + 5 * roi;
                const int cv = (C - cbk * 64) < 64 ? (C - cbk * 64) : 64;
                flag = fwd_max_task_class(w[2] - w[0], w[3] - w[1], may_nan, cv, C, aligned16, small_image) == 1;
            }
            unsigned long long m = __ballot(flag);
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const int64_t uu = base + l;
                run_task(keep_cb ? uu * cblocks + mycb : uu);
            }
        }
    }
#ifdef DG_STAMPS
    STAMP(tk1);
    STAMP_ADD(6, tk0, tk1);
    if (lane0 == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&dg_stamp_acc[i], stamp_loc[i]);
#endif
}

// ---------------------------------------------------------------- backward: per-image RoI lists
// RoIs usually arrive grouped by image, so a wave sees one or two image ids: one atomic per distinct id
// per wave (leader lane adds the group's size) instead of 64 colliding atomics.
__device__ __forceinline__ int wave_group_add(int32_t *counter_base, int key, bool valid, int *rank_in_group)
{
    const int lane = threadIdx.x & 63;
    int start = 0;
    *rank_in_group = 0;
    u64 todo = __ballot(valid);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int k = __shfl(key, leader, 64);
        const u64 grp = __ballot(valid && key == k);
        int base = 0;
        if (lane == leader) base = atomicAdd(&counter_base[k], __builtin_popcountll(grp));
        base = __shfl(base, leader, 64);
        if (valid && key == k) { start = base; *rank_in_group = __builtin_popcountll(grp & ((1ull << lane) - 1ull)); }
        todo &= ~grp;
    }
    return start;
}
__global__ __launch_bounds__(256) void k_roi_count(const int32_t *__restrict__ win, int64_t n, int32_t *__restrict__ cnt)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    int r;
    wave_group_add(cnt, valid ? win[5 * i + 4] : 0, valid, &r);
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
    const bool valid = i < n;
    int r;
    const int start = wave_group_add(cursor, valid ? win[5 * i + 4] : 0, valid, &r);
    if (valid) list[start + r] = (int32_t)i;
}

// one image: its list is every RoI in order (no count / scan / fill: three launches that cost a single-image backward 15 % of its time)
__global__ __launch_bounds__(256) void k_roi_list_single(int64_t n, int32_t *__restrict__ off, int32_t *__restrict__ list)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) list[i] = (int32_t)i;
    if (i == 0) { off[0] = 0; off[1] = (int32_t)n; }
}

// LDS float accumulate.  Measured on MI355X (tools/ubench/lds_atomic.hip): a wave-wide LDS f32 atomic add costs
// ~194 cycles whatever the address pattern, an i32 LDS atomic 5-9 cycles, so the add is built from a plain read
// and a 32-bit compare-and-swap; it only retries when two lanes really hit the same word.
__device__ __forceinline__ void lds_fadd(float *p, float g)
{
    int *ip = reinterpret_cast<int *>(p);
    int old = *reinterpret_cast<volatile int *>(ip);
    while (true) {
        const int assumed = old;
        old = atomicCAS(ip, assumed, __float_as_int(__int_as_float(assumed) + g));
        if (old == assumed) break;
    }
}

// four accumulates at p, p+stride, p+2*stride, p+3*stride with their reads and compare-and-swaps issued together
__device__ __forceinline__ void lds_fadd4(float *p, int stride, const float (&g)[4])
{
    int *ip = reinterpret_cast<int *>(p);
    int old[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) old[c] = *reinterpret_cast<volatile int *>(ip + c * stride);
    bool done[4] = {false, false, false, false};
    while (true) {
        int got[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            got[c] = done[c] ? old[c] : atomicCAS(ip + c * stride, old[c], __float_as_int(__int_as_float(old[c]) + g[c]));
        bool all = true;
#pragma unroll
        for (int c = 0; c < 4; ++c) { done[c] = done[c] || (got[c] == old[c]); old[c] = got[c]; all = all && done[c]; }
        if (all) break;
    }
}

// four channels of one pixel in a [pixel][4] plane: one 16-byte read and two 64-bit compare-and-swap adds
__device__ __forceinline__ void lds_fadd_px4(float *p, const float (&g)[4])
{
    unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const u64x2 cur = *reinterpret_cast<volatile u64x2 *>(p);
    unsigned long long old[2] = {cur.x, cur.y};
    bool done[2] = {false, false};
    while (true) {
        unsigned long long got[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float lo = __int_as_float((int)(unsigned)old[h]) + g[2 * h], hi = __int_as_float((int)(unsigned)(old[h] >> 32)) + g[2 * h + 1];
            const unsigned long long want = ((unsigned long long)(unsigned)__float_as_int(hi) << 32) | (unsigned)__float_as_int(lo);
            got[h] = done[h] ? old[h] : atomicCAS(q + h, old[h], want);
        }
        bool all = true;
#pragma unroll
        for (int h = 0; h < 2; ++h) { done[h] = done[h] || (got[h] == old[h]); old[h] = got[h]; all = all && done[h]; }
        if (all) break;
    }
}

// grid: x = channel block (CB channels), y = image.  Dynamic LDS: CB*H*W floats.
// VEC: CB*49 % 4 == 0 and 16-byte aligned tensors -> each thread moves 4 consecutive elements of one RoI
// (16-byte loads of grad_out and argmax), two such groups in flight per thread.
// IDENT (one image): the image's list is every RoI in order -- `list` and `off` are not read, the list has `nident` entries
// (no list-building launch in front of the kernel).
template <int MODE, bool VEC, int NT, bool A16, bool IDENT = false>
__global__ __launch_bounds__(NT) void k_roipool_bwd(const float *__restrict__ grad_out, const void *__restrict__ argmax_v,
                                                     const int32_t *__restrict__ win, const int32_t *__restrict__ off,
                                                     const int32_t *__restrict__ list_, int C, int H, int W, int CB,
                                                     float *__restrict__ grad_in, int nident)
{
    struct List {                                             // list[i], or i itself
        const int32_t *p;
        __device__ __forceinline__ int operator[](int i) const { return IDENT ? i : p[i]; }
    };
    const List list{list_};
    extern __shared__ __attribute__((aligned(16))) float plane[];
    // argmax as the forward wrote it: int32, or uint16 (A16: half the index bytes; H*W <= 65535 always holds here)
    const int32_t *argmax = reinterpret_cast<const int32_t *>(argmax_v);
    const unsigned short *argmax16 = reinterpret_cast<const unsigned short *>(argmax_v);
    auto idx4 = [&](int64_t o) {
        if (A16) { const ushort4 u = *reinterpret_cast<const ushort4 *>(argmax16 + o); return make_int4(u.x, u.y, u.z, u.w); }
        return *reinterpret_cast<const int4 *>(argmax + o);
    };
    const int tid = threadIdx.x;
    // blockIdx.x -> channel slice.  Workgroups are dealt round-robin over the 8 XCDs and gridDim.x is the fastest index, so
    // with the identity mapping neighbouring slices -- whose 784-byte pieces of every RoI share a 128-byte line -- always
    // sit on different XCDs and that line is fetched from memory twice.  When gridDim.x is a multiple of 8, XCD j takes
    // the contiguous slices [j*gridDim.x/8, (j+1)*gridDim.x/8): neighbours run side by side under one L2.
    const int nsl = gridDim.x;
    const int slice = (nsl % 8 == 0 && DG_BWD_XCD_REMAP) ? (int)(blockIdx.x % 8) * (nsl / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int b = blockIdx.y, c0 = slice * CB;
    const int cb = (C - c0) < CB ? (C - c0) : CB;
    const int HW = H * W;
    for (int i = tid; i < cb * HW; i += NT) plane[i] = 0.f;
    __syncthreads();
    // gridDim.z > 1 (few images): the image's RoI list is split over z, every part adds its LDS slice into a
    // pre-zeroed grad_in with global float atomics (contiguous 256-byte wave accesses)
    const int rlo = IDENT ? 0 : off[b], rhi = IDENT ? nident : off[b + 1];
    const int rchunk = (rhi - rlo + (int)gridDim.z - 1) / (int)gridDim.z;
    const int r0 = min(rhi, rlo + (int)blockIdx.z * rchunk), r1 = min(rhi, r0 + rchunk);
    const int per = cb * RBINS;                      // contiguous elements per RoI for this channel slice
    if (MODE == DG_POOL_MAX && VEC) {
        const int per4 = per >> 2;                   // float4 groups per RoI
        const int64_t work = (int64_t)(r1 - r0) * per4;
        int64_t t = tid;
        {
            constexpr int U = 4;                            // groups of 4 elements in flight per thread (8: no further gain)
            for (; t + (U - 1) * NT < work; t += U * NT) {
                int64_t o[U];
                int e[U];
                float4 g[U];
                int4 ix[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t tu = t + u * NT;
                    const int r = (int)(tu / per4), q = (int)(tu - (int64_t)r * per4);
                    o[u] = ((int64_t)list[r0 + r] * C + c0) * RBINS + 4 * q;
                    e[u] = 4 * q;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) { g[u] = *reinterpret_cast<const float4 *>(grad_out + o[u]); ix[u] = idx4(o[u]); }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    lds_fadd(&plane[((e[u] + 0) / RBINS) * HW + ix[u].x], g[u].x);
                    lds_fadd(&plane[((e[u] + 1) / RBINS) * HW + ix[u].y], g[u].y);
                    lds_fadd(&plane[((e[u] + 2) / RBINS) * HW + ix[u].z], g[u].z);
                    lds_fadd(&plane[((e[u] + 3) / RBINS) * HW + ix[u].w], g[u].w);
                }
            }
        }
        for (; t + NT < work; t += 2 * NT) {           // two independent 16-byte pairs in flight
            const int64_t t2 = t + NT;
            const int ra = (int)(t / per4), qa = (int)(t - (int64_t)ra * per4);
            const int rb = (int)(t2 / per4), qb = (int)(t2 - (int64_t)rb * per4);
            const int64_t oa = ((int64_t)list[r0 + ra] * C + c0) * RBINS + 4 * qa;
            const int64_t ob = ((int64_t)list[r0 + rb] * C + c0) * RBINS + 4 * qb;
            const float4 ga = *reinterpret_cast<const float4 *>(grad_out + oa);
            const int4 ia = idx4(oa);
            const float4 gb = *reinterpret_cast<const float4 *>(grad_out + ob);
            const int4 ib = idx4(ob);
            const int ea = 4 * qa, eb = 4 * qb;
            lds_fadd(&plane[((ea + 0) / RBINS) * HW + ia.x], ga.x);
            lds_fadd(&plane[((ea + 1) / RBINS) * HW + ia.y], ga.y);
            lds_fadd(&plane[((ea + 2) / RBINS) * HW + ia.z], ga.z);
            lds_fadd(&plane[((ea + 3) / RBINS) * HW + ia.w], ga.w);
            lds_fadd(&plane[((eb + 0) / RBINS) * HW + ib.x], gb.x);
            lds_fadd(&plane[((eb + 1) / RBINS) * HW + ib.y], gb.y);
            lds_fadd(&plane[((eb + 2) / RBINS) * HW + ib.z], gb.z);
            lds_fadd(&plane[((eb + 3) / RBINS) * HW + ib.w], gb.w);
        }
        for (; t < work; t += NT) {
            const int ra = (int)(t / per4), qa = (int)(t - (int64_t)ra * per4);
            const int64_t oa = ((int64_t)list[r0 + ra] * C + c0) * RBINS + 4 * qa;
            const float4 ga = *reinterpret_cast<const float4 *>(grad_out + oa);
            const int4 ia = idx4(oa);
            const int ea = 4 * qa;
            lds_fadd(&plane[((ea + 0) / RBINS) * HW + ia.x], ga.x);
            lds_fadd(&plane[((ea + 1) / RBINS) * HW + ia.y], ga.y);
            lds_fadd(&plane[((ea + 2) / RBINS) * HW + ia.z], ga.z);
            lds_fadd(&plane[((ea + 3) / RBINS) * HW + ia.w], ga.w);
        }
    } else if (MODE == DG_POOL_AVG) {
        // avg mode, one wave per RoI (all channels of the slice, at most 4).  The bin gradients become per-pixel shares
        // g/(rows*cols) in a per-wave LDS scratch; the lanes then cover whole crop rows (64 / width rows per pass), each
        // pixel sums the shares of the bins that contain it (1..2 per axis, more for crops under 7; the bin ranges come
        // from reciprocal multiplies, not integer divisions) and is added ONCE to the LDS plane, which is laid out
        // [pixel][4 channels] so that one 16-byte read and two 64-bit compare-and-swaps serve the four channels.  The
        // scatter form (every bin looping over its pixels) costs several times as many LDS read-modify-writes.  The
        // window and bin gradients of the next RoI are fetched while the current one is processed.
        __shared__ __attribute__((aligned(16))) float share[NT / 64][4 * 64];    // cb == 4: [wave][bin][4]; cb < 4: [wave][channel][bin]; the host keeps CB <= 4 in avg mode
        const int lane = tid & 63, wave = tid >> 6;
        // software pipeline over the wave's RoIs: the RoI index is fetched two ahead, window and bin gradients one ahead
        const int nr = r1 - r0;
        int rr = wave;
        int roi1 = (rr < nr) ? list[r0 + rr] : 0;                       // current
        int roi2 = (rr + NT / 64 < nr) ? list[r0 + rr + NT / 64] : 0;               // next
        int4 wn = make_int4(0, 0, 1, 1);
        float gn[4] = {0.f, 0.f, 0.f, 0.f};
        auto fetch_roi = [&](int roi) {
            const int32_t *w = win + 5 * (int64_t)roi;
            wn = make_int4(w[0], w[1], w[2], w[3]);
            if (lane < RBINS) {
                const float *g = grad_out + ((int64_t)roi * C + c0) * RBINS + lane;
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) gn[ch] = (ch < cb) ? g[ch * RBINS] : 0.f;
            }
        };
        if (rr < nr) fetch_roi(roi1);
        for (; rr < nr; rr += NT / 64) {
            const int x1 = __builtin_amdgcn_readfirstlane(wn.x), y1 = __builtin_amdgcn_readfirstlane(wn.y);
            const int Lw = __builtin_amdgcn_readfirstlane(wn.z) - x1, Lh = __builtin_amdgcn_readfirstlane(wn.w) - y1;
            float gc[4];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) gc[ch] = gn[ch];
            roi1 = roi2;
            if (rr + NT / 64 < nr) fetch_roi(roi1);
            roi2 = (rr + 2 * (NT / 64) < nr) ? list[r0 + rr + 2 * (NT / 64)] : 0;
            const bool small = Lh <= 64 && Lw <= 64;
            const bool two_max = Lh >= RP && Lw >= RP;                   // uniform
            // a / L for 0 <= a < 2^10 and L <= 64 without the ~35-instruction integer division: (a + 0.5) / L is at
            // least 1/128 away from every integer, far more than the error of rcp and one multiply
            const float rLh = __builtin_amdgcn_rcpf((float)Lh), rLw = __builtin_amdgcn_rcpf((float)Lw);
            auto fdiv = [](int a, float rcp) { return (int)(((float)a + 0.5f) * rcp); };
            if (lane < RBINS) {
                const int ph = lane / RP, pw = lane - ph * RP;
                const float kh = (float)(((ph + 1) * Lh + RP - 1) / RP - (ph * Lh) / RP), kw = (float)(((pw + 1) * Lw + RP - 1) / RP - (pw * Lw) / RP);
                if (cb == 4) {                       // the usual slice: the four channels of a bin side by side
                    // g * (1 / (kh*kw)): one division per bin instead of eight; within 2 ulp of ATen's g / kh / kw, far
                    // inside the accumulation-order tolerance of this kernel
                    const float inv = 1.0f / (kh * kw);
                    *reinterpret_cast<float4 *>(&share[wave][lane * 4]) = make_float4(gc[0] * inv, gc[1] * inv, gc[2] * inv, gc[3] * inv);
                } else {
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) share[wave][ch * 64 + lane] = gc[ch] / kh / kw;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float *pl = plane + y1 * W + x1;
            // lanes cover whole crop rows: 64 / Lw rows per pass (a 9-wide crop keeps 63 lanes busy; fixed 8x8 tiles kept
            // 81 of 256).  Crops wider than 64 cells walk 64-cell row segments.
            const int rpp = Lw <= 64 ? fdiv(64, rLw) : 1;                // rows per pass
            const int lyy = Lw <= 64 ? fdiv(lane, rLw) : 0, lxx = Lw <= 64 ? lane - lyy * Lw : lane;
            const bool lane_on = Lw <= 64 ? lyy < rpp : true;
            for (int ty = 0; ty < Lh; ty += rpp) {
                for (int tx = 0; tx < Lw; tx += 64) {
                    const int ry = ty + lyy, rx = tx + lxx;
                    if (lane_on && ry < Lh && rx < Lw) {
                        int ph0, ph1, pw0, pw1;
                        // bins whose range [floor(i*L/7), ceil((i+1)*L/7)) contains the coordinate
                        if (small) { ph0 = fdiv(ry * RP, rLh); ph1 = fdiv((ry + 1) * RP + Lh - 1, rLh) - 1; pw0 = fdiv(rx * RP, rLw); pw1 = fdiv((rx + 1) * RP + Lw - 1, rLw) - 1; }
                        else { ph0 = (ry * RP) / Lh; ph1 = ((ry + 1) * RP + Lh - 1) / Lh - 1; pw0 = (rx * RP) / Lw; pw1 = ((rx + 1) * RP + Lw - 1) / Lw - 1; }
                        if (cb == 4) {                       // plane is [pixel][4 channels]: one 16-byte read per share, two 64-bit CAS
                            float sum[4] = {0.f, 0.f, 0.f, 0.f};
                            if (two_max) {
                                // crops of at least 7x7: a pixel lies in one or two bins per axis -- four fixed reads, the
                                // repeats (ph1 == ph0 / pw1 == pw0) masked out by selects (not 0/1 factors: 0 * inf), no divergent loops
                                const bool mh = ph1 > ph0, mw = pw1 > pw0, mhw = mh && mw;
                                const float4 a = *reinterpret_cast<const float4 *>(&share[wave][(ph0 * RP + pw0) * 4]);
                                const float4 b2 = *reinterpret_cast<const float4 *>(&share[wave][(ph0 * RP + pw1) * 4]);
                                const float4 c2 = *reinterpret_cast<const float4 *>(&share[wave][(ph1 * RP + pw0) * 4]);
                                const float4 d2 = *reinterpret_cast<const float4 *>(&share[wave][(ph1 * RP + pw1) * 4]);
                                sum[0] = ((a.x + (mw ? b2.x : 0.f)) + (mh ? c2.x : 0.f)) + (mhw ? d2.x : 0.f);
                                sum[1] = ((a.y + (mw ? b2.y : 0.f)) + (mh ? c2.y : 0.f)) + (mhw ? d2.y : 0.f);
                                sum[2] = ((a.z + (mw ? b2.z : 0.f)) + (mh ? c2.z : 0.f)) + (mhw ? d2.z : 0.f);
                                sum[3] = ((a.w + (mw ? b2.w : 0.f)) + (mh ? c2.w : 0.f)) + (mhw ? d2.w : 0.f);
                            } else {
                                for (int ph = ph0; ph <= ph1; ++ph)
                                    for (int pw = pw0; pw <= pw1; ++pw) {
                                        const float4 sp = *reinterpret_cast<const float4 *>(&share[wave][(ph * RP + pw) * 4]);
                                        sum[0] += sp.x; sum[1] += sp.y; sum[2] += sp.z; sum[3] += sp.w;
                                    }
                            }
                            lds_fadd_px4(&plane[((y1 + ry) * W + (x1 + rx)) * 4], sum);
                        } else {
                            for (int ch = 0; ch < cb; ++ch) {
                                float sum = 0.f;
                                for (int ph = ph0; ph <= ph1; ++ph)
                                    for (int pw = pw0; pw <= pw1; ++pw) sum += share[wave][ch * 64 + ph * RP + pw];
                                lds_fadd(&pl[ch * HW + ry * W + rx], sum);
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();               // the scratch is rewritten for the next RoI
        }
    } else {
        const int64_t work = (int64_t)(r1 - r0) * per;
        for (int64_t t = tid; t < work; t += NT) {
            const int rr = (int)(t / per), e = (int)(t - (int64_t)rr * per);
            const int roi = list[r0 + rr];
            const int64_t o = ((int64_t)roi * C + c0) * RBINS + e;
            lds_fadd(&plane[(e / RBINS) * HW + (A16 ? (int)argmax16[o] : argmax[o])], grad_out[o]);
        }
    }
    __syncthreads();
    float *gi = grad_in + ((int64_t)b * C + c0) * HW;
    if (MODE == DG_POOL_AVG && cb == 4) {              // [pixel][4] in LDS -> four NCHW planes
        for (int i = tid; i < 4 * HW; i += NT) {
            const int ch = i / HW, px = i - ch * HW;
            if (gridDim.z == 1) gi[i] = plane[px * 4 + ch];
            else atomicAdd(&gi[i], plane[px * 4 + ch]);
        }
    } else if (gridDim.z == 1) {
        for (int i = tid; i < cb * HW; i += NT) gi[i] = plane[i];
    } else {
        for (int i = tid; i < cb * HW; i += NT) atomicAdd(&gi[i], plane[i]);
    }
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

extern "C" int dg_nchw_to_nhwc_f32(const float *src, int64_t B, int64_t C, int64_t H, int64_t W, float *dst, int32_t *nan_flag,
                                   dg_stream_t s)
{
    DG_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad shape");
    DG_REQUIRE(src && dst, "nchw_to_nhwc: null pointer");
    DG_REQUIRE(B <= 65535 && dg_cdiv(C, 64) <= 65535, "nchw_to_nhwc: shape too large");
    dim3 grid((unsigned)dg_cdiv(H * W, 64), (unsigned)dg_cdiv(C, 64), (unsigned)B);
    hipLaunchKernelGGL(k_nchw_to_nhwc, grid, dim3(256), 0, (hipStream_t)s, src, (int)C, (int)(H * W), dst, nan_flag);
    DG_CHECK_LAUNCH("k_nchw_to_nhwc");
    return DG_OK;
}

// roipool_generic.hip: any output size, any map size
int dg_roipool_fwd_generic(const float *img_nhwc, int64_t B, int64_t C, int64_t H, int64_t W, const int32_t *win, int64_t n,
                           int64_t PH, int64_t PW, int mode, float *out, int32_t *argmax, hipStream_t st);
int dg_roipool_bwd_generic(const float *grad_out, const void *argmax, bool a16, const int32_t *win, int64_t n, int64_t B, int64_t C,
                           int64_t H, int64_t W, int64_t PH, int64_t PW, int mode, float *grad_in, hipStream_t st);

static int roipool_fwd(const float *img_nhwc, int64_t B, int64_t C, int64_t H, int64_t W, const int32_t *win, int64_t n, int64_t PH,
                       int64_t PW, int mode, const int32_t *nan_flag, float *out, int32_t *argmax, uint16_t *argmax16, dg_stream_t s)
{
    DG_REQUIRE(n >= 0 && B > 0 && C > 0 && H > 0 && W > 0, "roipool_fwd: bad shape");
    DG_REQUIRE(PH >= 1 && PW >= 1 && PH * PW <= (1 << 20), "roipool_fwd: output size must be >= 1 x 1 (got %lld x %lld)", (long long)PH, (long long)PW);
    DG_REQUIRE(mode == DG_POOL_MAX || mode == DG_POOL_AVG, "roipool_fwd: bad mode");
    if (n == 0) return DG_OK;
    DG_REQUIRE(img_nhwc && win && out, "roipool_fwd: null pointer");
    if (PH != RP || PW != RP || H * W > 65535) {
        // the general kernel: output sizes other than 7 x 7, maps beyond the 16-bit staged index
        DG_REQUIRE(!argmax16, "roipool_fwd: the uint16 argmax exists for 7 x 7 outputs on maps of at most 65535 cells");
        return dg_roipool_fwd_generic(img_nhwc, B, C, H, W, win, n, PH, PW, mode, out, argmax, (hipStream_t)s);
    }
    DG_REQUIRE((((uintptr_t)argmax16) & 1) == 0, "roipool_fwd: misaligned argmax16");
    int vec_ok = ((C % 4 == 0) && ((((uintptr_t)out) & 15) == 0) && (!argmax || (((uintptr_t)argmax) & 15) == 0)) ? 1 : 0;
    if (argmax16 && C % 8 == 0 && (((uintptr_t)argmax16) & 15) == 0) vec_ok |= 2;
    hipStream_t st = (hipStream_t)s;
    const int64_t cblocks = dg_cdiv(C, 64);
    const int64_t ntasks = n * cblocks;
    // 256 CUs x 8 single-wave workgroups (18.8 KB LDS each) are resident at a time.  Large launches do NOT use that many
    // persistent workgroups: with about FWD_TASKS_PER_WG block-strided tasks per workgroup (grid a multiple of 8, so a
    // workgroup still keeps one channel block and its XCD) the same kernel runs 6-16 % faster on config 3, depending on
    // the box (20.4 -> 17.1-17.6 ms, 18.5 -> 17.4 ms; 4 to 16 tasks per workgroup are alike, one task per workgroup is back
    // at 20.3 ms: a workgroup start costs about as much as it saves).  What it buys is placement by the dispatcher -- a
    // workgroup goes wherever a slot frees up -- instead of a fixed 1/2048 share of the tasks per wave; giving half of the
    // persistent waves a higher s_setprio changed nothing, so it is not the age/priority order of the two waves of a SIMD.
    // Grids of 8k-32k workgroups are slower (23-25 ms: a workgroup's stride then equals about one image, every wave
    // walks across all maps and the XCD's L2 no longer holds the slices in use).
    // DG_FWD_TPW overrides the tasks per workgroup (0 = the persistent grid), for experiments.
    static const int tpw_env = getenv("DG_FWD_TPW") ? atoi(getenv("DG_FWD_TPW")) : FWD_TASKS_PER_WG;
    const int64_t resident = 256 * 8;
    int64_t ngrid = ntasks < resident ? ntasks : resident;
    if (tpw_env > 0 && ntasks >= resident * 4 * tpw_env) ngrid = (ntasks / tpw_env + 7) & ~(int64_t)7;
    dim3 grid((unsigned)ngrid);
    // the wide-crop kernel (tasks of very unequal cost): 4 x the resident workgroups, for the same reason (mix U(0.1,0.9):
    // 8.4 -> 8.9 M RoIs/s; 4096 to 32768 workgroups are alike)
    const int64_t wide_cap = resident * 4;
    dim3 grid_wide((unsigned)(ntasks < wide_cap ? ntasks : wide_cap));
#define FWD_LAUNCH(MODE, A16, PART) \
    hipLaunchKernelGGL((k_roipool_fwd<MODE, A16, PART>), (PART == 0 ? grid : grid_wide), dim3(64), 0, st, img_nhwc, (int)C, (int)H, (int)W, win, (int)cblocks, ntasks, nan_flag, out, argmax, argmax16, vec_ok)
    // max mode: the main kernel, then the wide-crop kernel over the tasks the first one left (disjoint outputs)
    if (mode == DG_POOL_MAX && argmax16) { FWD_LAUNCH(DG_POOL_MAX, true, 0); if (DG_FWD_WIDE) FWD_LAUNCH(DG_POOL_MAX, true, 1); }
    else if (mode == DG_POOL_MAX) { FWD_LAUNCH(DG_POOL_MAX, false, 0); if (DG_FWD_WIDE) FWD_LAUNCH(DG_POOL_MAX, false, 1); }
    else FWD_LAUNCH(DG_POOL_AVG, false, 0);
#undef FWD_LAUNCH
    DG_CHECK_LAUNCH("k_roipool_fwd");
    return DG_OK;
}

extern "C" int dg_roipool_fwd_f32(const float *img_nhwc, int64_t B, int64_t C, int64_t H, int64_t W,
                                  const int32_t *win, int64_t n, int64_t P, int mode, const int32_t *nan_flag,
                                  float *out, int32_t *argmax, dg_stream_t s)
{
    return roipool_fwd(img_nhwc, B, C, H, W, win, n, P, P, mode, nan_flag, out, argmax, nullptr, s);
}

extern "C" int dg_roipool_fwd_any_f32(const float *img_nhwc, int64_t B, int64_t C, int64_t H, int64_t W,
                                      const int32_t *win, int64_t n, int64_t PH, int64_t PW, int mode, const int32_t *nan_flag,
                                      float *out, int32_t *argmax, dg_stream_t s)
{
    return roipool_fwd(img_nhwc, B, C, H, W, win, n, PH, PW, mode, nan_flag, out, argmax, nullptr, s);
}

extern "C" int dg_roipool_fwd_a16_f32(const float *img_nhwc, int64_t B, int64_t C, int64_t H, int64_t W,
                                      const int32_t *win, int64_t n, int64_t P, int mode, const int32_t *nan_flag,
                                      float *out, uint16_t *argmax16, dg_stream_t s)
{
    return roipool_fwd(img_nhwc, B, C, H, W, win, n, P, P, mode, nan_flag, out, nullptr, argmax16, s);
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

static int roipool_bwd(const float *grad_out, const void *argmax, bool a16, const int32_t *win, int64_t n,
                       int64_t B, int64_t C, int64_t H, int64_t W, int64_t PH, int64_t PW, int mode, float *grad_in,
                       void *ws, size_t ws_bytes, dg_stream_t s)
{
    DG_REQUIRE(n >= 0 && B > 0 && C > 0 && H > 0 && W > 0, "roipool_bwd: bad shape");
    DG_REQUIRE(PH >= 1 && PW >= 1, "roipool_bwd: output size must be >= 1 x 1");
    DG_REQUIRE(mode == DG_POOL_MAX || mode == DG_POOL_AVG, "roipool_bwd: bad mode");
    DG_REQUIRE(grad_in && ws, "roipool_bwd: null pointer");
    DG_REQUIRE(n == 0 || (grad_out && win), "roipool_bwd: null pointer");
    DG_REQUIRE(mode != DG_POOL_MAX || n == 0 || argmax, "roipool_bwd: max mode needs argmax");
    DG_REQUIRE(B <= 65535 && n < (1ll << 31), "roipool_bwd: shape too large");
    BwdWs w = bwd_layout(B, n, ws);
    if (ws_bytes < w.total) { dg_set_error("roipool_bwd: workspace %zu < %zu bytes", ws_bytes, w.total); return DG_ENOMEM; }
    const int64_t HW = H * W;
    hipStream_t st = (hipStream_t)s;
    // The tuned kernel keeps whole planes of grad_in in LDS: 7 x 7 outputs, one plane within 128 KiB (max mode; H*W <= 32768)
    // or a [pixel][4] plane within 64 KiB beside the 8 KiB of shares (avg mode, as before).  Everything else -- other output
    // sizes, larger maps -- goes through the general kernel (global float atomics into a zeroed grad_in).
    const int64_t plane_limit = (mode == DG_POOL_MAX ? 128 : 56) * 1024;
    if (PH != RP || PW != RP || HW * 4 > plane_limit || (a16 && HW > 65535)) {
        DG_REQUIRE(!(a16 && HW > 65535), "roipool_bwd: a uint16 argmax cannot index a map of more than 65535 cells");
        if (n == 0) return hipMemsetAsync(grad_in, 0, (size_t)B * C * HW * 4, st) == hipSuccess ? DG_OK : DG_EHIP;
        return dg_roipool_bwd_generic(grad_out, argmax, a16, win, n, B, C, H, W, PH, PW, mode, grad_in, st);
    }
    // channels per workgroup: as many as fit ~32 KiB of LDS, but keep >= ~1024 workgroups when possible
    int CB = 1;
    while (CB < 4 && (int64_t)CB * 2 * HW * 4 <= 32 * 1024 && CB * 2 <= C) CB *= 2;      // 4 channels: 16-byte vector path
    while (mode == DG_POOL_MAX && CB < 16 && (int64_t)CB * 2 * HW * 4 <= 32 * 1024 && B * dg_cdiv(C, CB * 2) >= 1024) CB *= 2;
    // Max mode with enough images: one 1024-thread workgroup per (image, 16 channels) -- 3136 contiguous bytes of
    // grad_out / argmax per RoI, so only 1 line in 25 is shared with a neighbour (4-channel slices: 784 bytes, 1 in 7, and
    // FETCH_SIZE read 1.17x the algorithmic bytes); the 16 planes take 16*H*W*4 bytes of LDS (121.6 KB at 38x50), one
    // workgroup per CU.  Measured at 256 images x 2000 RoIs: 18.9 ms (4 channels, 256 threads) -> 18.1 (8, 512) -> 17.6
    // (16, 1024); the wider slices with 256 threads: 27.7 ms (too few waves per CU).
    int NT = 256;
    if (mode == DG_POOL_MAX && C % 16 == 0 && (int64_t)16 * HW * 4 <= 144 * 1024 && B * (C / 16) >= 256) { CB = 16; NT = 1024; }
    else if (mode == DG_POOL_MAX && C % 8 == 0 && (int64_t)8 * HW * 4 <= 72 * 1024 && B * (C / 8) >= 512) { CB = 8; NT = 512; }
    // (one image of 2000 RoIs, round 3: the wider slices do not help below those image counts -- 4 channels x 8 list splits
    //  103.5 us, 8 x 8 103.3, 16 x 8 105.7, and every other split count 121-136 us)
    int64_t S = 1;                                     // RoI-list splits when there are too few (image, slice) pairs
    while (S < 16 && B * dg_cdiv(C, CB) * S < 1024 && n / (B * S * 2) >= 64) S *= 2;
    const bool ident = B == 1 && mode == DG_POOL_MAX && NT == 256;   // one image, max mode: the kernel takes the RoIs in order itself
    if (ident) {
    } else if (B == 1) {
        hipLaunchKernelGGL(k_roi_list_single, dim3((unsigned)dg_cdiv(n > 0 ? n : 1, 256)), dim3(256), 0, st, n, w.off, w.list);
    } else {
        if (hipMemsetAsync(w.cnt, 0, (size_t)B * 4, st) != hipSuccess) { dg_set_error("roipool_bwd: memset failed"); return DG_EHIP; }
        if (n > 0) hipLaunchKernelGGL(k_roi_count, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, st, win, n, w.cnt);
        hipLaunchKernelGGL(k_roi_scan, dim3(1), dim3(1024), 0, st, w.cnt, (int)B, w.off, w.cursor);
        if (n > 0) hipLaunchKernelGGL(k_roi_fill, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, st, win, n, w.cursor, w.list);
    }
    dim3 grid((unsigned)dg_cdiv(C, CB), (unsigned)B, (unsigned)S);
    size_t lds = (size_t)CB * HW * 4;
    if (S > 1 && hipMemsetAsync(grad_in, 0, (size_t)B * C * HW * 4, st) != hipSuccess) { dg_set_error("roipool_bwd: memset failed"); return DG_EHIP; }
    // vector path: every channel slice is whole groups of 4 elements and 16-byte aligned
    const bool vec = mode == DG_POOL_MAX && CB % 4 == 0 && C % CB == 0 && (((uintptr_t)grad_out) & 15) == 0 &&
                     (((uintptr_t)argmax) & (a16 ? 7 : 15)) == 0;
#define BWD_LAUNCH(MODE, VEC, NT, A16) \
    hipLaunchKernelGGL((k_roipool_bwd<MODE, VEC, NT, A16>), grid, dim3(NT), lds, st, grad_out, argmax, win, w.off, w.list, (int)C, (int)H, (int)W, CB, grad_in, 0)
#define BWD_LAUNCH_IDENT(VEC, A16) \
    hipLaunchKernelGGL((k_roipool_bwd<DG_POOL_MAX, VEC, 256, A16, true>), grid, dim3(256), lds, st, grad_out, argmax, win, w.off, w.list, (int)C, (int)H, (int)W, CB, grad_in, (int)n)
    if (mode == DG_POOL_MAX && vec && NT == 512) { if (a16) BWD_LAUNCH(DG_POOL_MAX, true, 512, true); else BWD_LAUNCH(DG_POOL_MAX, true, 512, false); }
    else if (mode == DG_POOL_MAX && vec && NT == 1024) { if (a16) BWD_LAUNCH(DG_POOL_MAX, true, 1024, true); else BWD_LAUNCH(DG_POOL_MAX, true, 1024, false); }
    else if (ident && NT == 256 && vec) { if (a16) BWD_LAUNCH_IDENT(true, true); else BWD_LAUNCH_IDENT(true, false); }
    else if (ident && NT == 256) { if (a16) BWD_LAUNCH_IDENT(false, true); else BWD_LAUNCH_IDENT(false, false); }
    else if (mode == DG_POOL_MAX && vec) { if (a16) BWD_LAUNCH(DG_POOL_MAX, true, 256, true); else BWD_LAUNCH(DG_POOL_MAX, true, 256, false); }
    else if (mode == DG_POOL_MAX) { if (a16) BWD_LAUNCH(DG_POOL_MAX, false, 256, true); else BWD_LAUNCH(DG_POOL_MAX, false, 256, false); }
    else BWD_LAUNCH(DG_POOL_AVG, false, 512, false);
#undef BWD_LAUNCH
#undef BWD_LAUNCH_IDENT
    DG_CHECK_LAUNCH("k_roipool_bwd");
    return DG_OK;
}

extern "C" int dg_roipool_bwd_f32(const float *grad_out, const int32_t *argmax, const int32_t *win, int64_t n,
                                  int64_t B, int64_t C, int64_t H, int64_t W, int64_t P, int mode, float *grad_in,
                                  void *ws, size_t ws_bytes, dg_stream_t s)
{
    return roipool_bwd(grad_out, argmax, false, win, n, B, C, H, W, P, P, mode, grad_in, ws, ws_bytes, s);
}

extern "C" int dg_roipool_bwd_any_f32(const float *grad_out, const int32_t *argmax, const int32_t *win, int64_t n,
                                      int64_t B, int64_t C, int64_t H, int64_t W, int64_t PH, int64_t PW, int mode, float *grad_in,
                                      void *ws, size_t ws_bytes, dg_stream_t s)
{
    return roipool_bwd(grad_out, argmax, false, win, n, B, C, H, W, PH, PW, mode, grad_in, ws, ws_bytes, s);
}

extern "C" int dg_roipool_bwd_a16_f32(const float *grad_out, const uint16_t *argmax16, const int32_t *win, int64_t n,
                                      int64_t B, int64_t C, int64_t H, int64_t W, int64_t P, int mode, float *grad_in,
                                      void *ws, size_t ws_bytes, dg_stream_t s)
{
    return roipool_bwd(grad_out, argmax16, true, win, n, B, C, H, W, P, P, mode, grad_in, ws, ws_bytes, s);
}
