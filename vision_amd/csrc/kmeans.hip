// kmeans.hip — YOLOv2-style anchor k-means with d = 1 - IoU (gfx950).
// The reference only fixes the sklearn-like API (clustering/cluster_prac.py:38-40); the Lloyd step
// is build-defined and specified in oracle/oracle.c (orc_kmeans_step / orc_kmeans_update):
//   d = 1 - inter/((w*h + cw*ch) - inter), fp32, ties -> lowest index; sums in int64 fixed point
//   (llrint(x * 2^32)) so any sharding / reduction order gives identical bits.
// Step kernel: grid-stride, two boxes per lane per 16-byte load, centroids in SGPRs, per-lane
// int64 partial sums in registers -> wave shuffle reduce -> LDS -> one partial row per workgroup
// (plain stores); a single-workgroup finish kernel folds the rows into acc[3k+2].
#include "common.hpp"

#define KM_BLOCK 256
#define KM_MAXBLOCKS 1024

__device__ __forceinline__ float kdist(float w, float h, float cw, float ch)
{
    float inter = fminf(w, cw) * fminf(h, ch);
    float uni = ((w * h) + (cw * ch)) - inter;
    return 1.0f - inter / uni;
}
__device__ __forceinline__ long long fx32(float v) { return __double2ll_rn((double)v * 4294967296.0); }

__device__ __forceinline__ long long wave_sum_ll(long long v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Per-lane accumulators live in LDS ([cluster][w|h][thread] u64 + [cluster][thread] u32): integer LDS atomics to
// a lane-private word cost ~6 LDS cycles and run beside the VALU, where 3*k predicated 64-bit register adds
// (half-rate v_cndmask) were the larger half of the kernel.  Addresses differ only by multiples of 2 KiB between
// clusters, so every wave access is conflict-free whatever the labels are.
template <int KMAX>
__device__ __forceinline__ void km_one(float w, float h, const float (&cw)[KMAX], const float (&ch)[KMAX], int k,
                                       unsigned long long *sacc, unsigned *scnt, long long &inertia, int &changed,
                                       uint8_t *lab)
{
    int best = 0;
    float bd = kdist(w, h, cw[0], ch[0]);
#pragma unroll
    for (int c = 1; c < KMAX; ++c) {
        if (c < k) {
            float d = kdist(w, h, cw[c], ch[c]);
            if (d < bd) { bd = d; best = c; }
        }
    }
    atomicAdd(&sacc[(best * 2 + 0) * KM_BLOCK], (unsigned long long)fx32(w));
    atomicAdd(&sacc[(best * 2 + 1) * KM_BLOCK], (unsigned long long)fx32(h));
    atomicAdd(&scnt[best * KM_BLOCK], 1u);
    inertia += fx32(bd);
    changed += (*lab != (uint8_t)best);
    *lab = (uint8_t)best;
}

template <int KMAX>
__global__ __launch_bounds__(KM_BLOCK) void k_kmeans_step(const float2 *__restrict__ wh, int64_t n, const float *__restrict__ cent,
                                                          int k, uint8_t *__restrict__ labels, long long *__restrict__ partials)
{
    __shared__ unsigned long long sacc_all[KMAX * 2 * KM_BLOCK];
    __shared__ unsigned scnt_all[KMAX * KM_BLOCK];
    __shared__ long long red[KM_BLOCK / 64][3 * KMAX + 2];
    const int tid = threadIdx.x;
    float cw[KMAX], ch[KMAX];
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        cw[c] = (c < k) ? cent[2 * c] : 0.f;
        ch[c] = (c < k) ? cent[2 * c + 1] : 0.f;
        sacc_all[(c * 2 + 0) * KM_BLOCK + tid] = 0;
        sacc_all[(c * 2 + 1) * KM_BLOCK + tid] = 0;
        scnt_all[c * KM_BLOCK + tid] = 0;
    }
    unsigned long long *sacc = sacc_all + tid;     // lane-private column: no barrier needed until the final reduce
    unsigned *scnt = scnt_all + tid;
    long long inertia = 0;
    int changed = 0;

    // two boxes per lane per 16-byte load, two loads in flight; an 8-byte-aligned shard start is peeled off first
    const int head = ((reinterpret_cast<uintptr_t>(wh) & 8) && n > 0) ? 1 : 0;
    if (head && blockIdx.x == 0 && tid == 0) {
        const float2 a = wh[0];
        km_one<KMAX>(a.x, a.y, cw, ch, k, sacc, scnt, inertia, changed, labels);
    }
    wh += head; n -= head;
    const int64_t npair = n >> 1;
    const float4 *wh4 = reinterpret_cast<const float4 *>(wh);
    uint8_t *lab = labels + head;
    const int64_t stride = (int64_t)gridDim.x * KM_BLOCK;
    int64_t i = (int64_t)blockIdx.x * KM_BLOCK + tid;
    for (; i + stride < npair; i += 2 * stride) {
        const float4 a = wh4[i], b = wh4[i + stride];
        km_one<KMAX>(a.x, a.y, cw, ch, k, sacc, scnt, inertia, changed, lab + 2 * i);
        km_one<KMAX>(a.z, a.w, cw, ch, k, sacc, scnt, inertia, changed, lab + 2 * i + 1);
        km_one<KMAX>(b.x, b.y, cw, ch, k, sacc, scnt, inertia, changed, lab + 2 * (i + stride));
        km_one<KMAX>(b.z, b.w, cw, ch, k, sacc, scnt, inertia, changed, lab + 2 * (i + stride) + 1);
    }
    for (; i < npair; i += stride) {
        const float4 a = wh4[i];
        km_one<KMAX>(a.x, a.y, cw, ch, k, sacc, scnt, inertia, changed, lab + 2 * i);
        km_one<KMAX>(a.z, a.w, cw, ch, k, sacc, scnt, inertia, changed, lab + 2 * i + 1);
    }
    if ((n & 1) && blockIdx.x == 0 && tid == 0) {
        const float2 a = wh[n - 1];
        km_one<KMAX>(a.x, a.y, cw, ch, k, sacc, scnt, inertia, changed, lab + (n - 1));
    }

    // block reduction: wave shuffles, then LDS across the 4 waves
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        long long a = wave_sum_ll((long long)sacc[(c * 2 + 0) * KM_BLOCK]), b2 = wave_sum_ll((long long)sacc[(c * 2 + 1) * KM_BLOCK]);
        long long c2 = wave_sum_ll((long long)scnt[c * KM_BLOCK]);
        if (lane == 0) { red[wave][c] = a; red[wave][KMAX + c] = b2; red[wave][2 * KMAX + c] = c2; }
    }
    {
        long long a = wave_sum_ll((long long)changed), b2 = wave_sum_ll(inertia);
        if (lane == 0) { red[wave][3 * KMAX] = a; red[wave][3 * KMAX + 1] = b2; }
    }
    __syncthreads();
    if (tid < 3 * KMAX + 2) {
        long long v = 0;
#pragma unroll
        for (int w = 0; w < KM_BLOCK / 64; ++w) v += red[w][tid];
        partials[(int64_t)blockIdx.x * (3 * KMAX + 2) + tid] = v;
    }
}

// folds the per-workgroup rows (layout with KMAX columns) into acc (layout with k columns); one workgroup per column
__global__ __launch_bounds__(256) void k_kmeans_finish(const long long *__restrict__ partials, int nblocks, int KMAX, int k,
                                                       long long *__restrict__ acc)
{
    __shared__ long long red[4];
    const int cols = 3 * KMAX + 2, col = blockIdx.x;
    long long v = 0;
    for (int r = threadIdx.x; r < nblocks; r += 256) v += partials[(int64_t)r * cols + col];
    v = wave_sum_ll(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long t = (red[0] + red[1]) + (red[2] + red[3]);
        const int grp = col / KMAX, c = col - grp * KMAX;
        if (grp < 3) { if (c < k) acc[grp * k + c] = t; }
        else acc[3 * k + c] = t;                    // changed, inertia
    }
}

__global__ void k_kmeans_update(const long long *__restrict__ acc, int k, float *__restrict__ cent,
                                long long *__restrict__ history, int64_t slot)
{
    const int c = threadIdx.x;
    if (c < k) {
        long long cnt = acc[2 * k + c];
        if (cnt > 0) {
            cent[2 * c] = (float)((double)acc[c] / (double)cnt / 4294967296.0);
            cent[2 * c + 1] = (float)((double)acc[k + c] / (double)cnt / 4294967296.0);
        }
    }
    if (c == 0 && history) history[slot] = acc[3 * k];
}

__global__ __launch_bounds__(256) void k_labels_widen(const uint8_t *__restrict__ l, int64_t n, int32_t *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = l[i];
}

static inline int km_kmax(int64_t k) { return k <= 5 ? 5 : (k <= 9 ? 9 : 16); }
static inline int km_blocks(int64_t n) { int64_t b = dg_cdiv(n, KM_BLOCK * 4); return (int)(b < 1 ? 1 : (b > KM_MAXBLOCKS ? KM_MAXBLOCKS : b)); }

extern "C" size_t dg_kmeans_workspace_bytes(int64_t n, int64_t k)
{
    if (n < 0 || k <= 0 || k > 16) return 0;
    return (size_t)KM_MAXBLOCKS * (3 * 16 + 2) * 8;
}

extern "C" int dg_kmeans_iou_step(const float *wh, int64_t n, const float *cent, int64_t k, uint8_t *labels, int64_t *acc,
                                  void *ws, size_t ws_bytes, dg_stream_t s)
{
    DG_REQUIRE(n >= 0 && k >= 1 && k <= 16, "kmeans_step: need n >= 0 and 1 <= k <= 16");
    DG_REQUIRE(cent && acc && ws, "kmeans_step: null pointer");
    DG_REQUIRE(n == 0 || (wh && labels), "kmeans_step: null pointer");
    DG_REQUIRE((((uintptr_t)wh) & 7) == 0 && (((uintptr_t)ws) & 7) == 0, "kmeans_step: misaligned pointer");
    if (ws_bytes < dg_kmeans_workspace_bytes(n, k)) { dg_set_error("kmeans_step: workspace too small"); return DG_ENOMEM; }
    hipStream_t st = (hipStream_t)s;
    const int kmax = km_kmax(k), nb = km_blocks(n);
    long long *partials = (long long *)ws;
    if (kmax == 5)
        hipLaunchKernelGGL((k_kmeans_step<5>), dim3(nb), dim3(KM_BLOCK), 0, st, (const float2 *)wh, n, cent, (int)k, labels, partials);
    else if (kmax == 9)
        hipLaunchKernelGGL((k_kmeans_step<9>), dim3(nb), dim3(KM_BLOCK), 0, st, (const float2 *)wh, n, cent, (int)k, labels, partials);
    else
        hipLaunchKernelGGL((k_kmeans_step<16>), dim3(nb), dim3(KM_BLOCK), 0, st, (const float2 *)wh, n, cent, (int)k, labels, partials);
    hipLaunchKernelGGL(k_kmeans_finish, dim3(3 * kmax + 2), dim3(256), 0, st, partials, nb, kmax, (int)k, (long long *)acc);
    DG_CHECK_LAUNCH("k_kmeans_step");
    return DG_OK;
}

extern "C" int dg_kmeans_update(const int64_t *acc, int64_t k, float *cent, int64_t *history, int64_t slot, dg_stream_t s)
{
    DG_REQUIRE(k >= 1 && k <= 16 && acc && cent, "kmeans_update: bad argument");
    hipLaunchKernelGGL(k_kmeans_update, dim3(1), dim3(64), 0, (hipStream_t)s, (const long long *)acc, (int)k, cent,
                       (long long *)history, slot);
    DG_CHECK_LAUNCH("k_kmeans_update");
    return DG_OK;
}

// `iters` Lloyd iterations back to back on one device (step + finish + update per iteration, no host round trip);
// history[slot0 + t] = changed count of iteration t.  A converged iteration reproduces the same centroids, so
// running past convergence is idempotent and n_iter_ is read from the history afterwards.
extern "C" int dg_kmeans_iou_run(const float *wh, int64_t n, float *cent, int64_t k, uint8_t *labels, int64_t *acc,
                                 int64_t *history, int64_t slot0, int64_t iters, void *ws, size_t ws_bytes, dg_stream_t s)
{
    DG_REQUIRE(iters >= 0, "kmeans_run: iters < 0");
    for (int64_t t = 0; t < iters; ++t) {
        int rc = dg_kmeans_iou_step(wh, n, cent, k, labels, acc, ws, ws_bytes, s);
        if (rc) return rc;
        rc = dg_kmeans_update(acc, k, cent, history, slot0 + t, s);
        if (rc) return rc;
    }
    return DG_OK;
}

extern "C" int dg_kmeans_labels_i32(const uint8_t *labels, int64_t n, int32_t *out, dg_stream_t s)
{
    DG_REQUIRE(n >= 0, "kmeans_labels: n < 0");
    if (n == 0) return DG_OK;
    DG_REQUIRE(labels && out, "kmeans_labels: null pointer");
    int64_t g = dg_cdiv(n, 256);
    hipLaunchKernelGGL(k_labels_widen, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)s, labels, n, out);
    DG_CHECK_LAUNCH("k_labels_widen");
    return DG_OK;
}
