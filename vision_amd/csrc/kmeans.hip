// kmeans.hip — YOLOv2-style anchor k-means with d = 1 - IoU (gfx950).
// The reference only fixes the sklearn-like API (clustering/cluster_prac.py:38-40); the Lloyd step
// is build-defined and specified in oracle/oracle.c (orc_kmeans_step / orc_kmeans_update):
//   d = 1 - inter/((w*h + cw*ch) - inter), fp32, ties -> lowest index; sums in int64 fixed point
//   (llrint(x * 2^32)) so any sharding / reduction order gives identical bits.
// Step kernel: grid-stride, two boxes per lane per 16-byte load, centroids in SGPRs, per-lane partial sums in LDS ->
// wave shuffle reduce -> one row per workgroup, added with 64-bit global atomics into one of KM_REPL replica rows of a
// "partial sums" buffer (replica = blockIdx % 8: workgroups that share an XCD share a row).  Nothing in a launch waits
// for another workgroup: the replicas are folded, and the centroids moved, by the PROLOGUE of the next launch (every
// workgroup does it for itself: 8 x (3k+2) int64), or by k_kmeans_finish after the last one.  A Lloyd iteration is one
// launch (+ one all-reduce of the replica buffer when the boxes are sharded over GPUs).
#include "common.hpp"
#include <stdlib.h>
#include <string.h>

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
//
// Fixed-point conversion: for |x| < 2^19, fma((double)x, 2^32, 1.5*2^52) lands in the binade whose ulp is 1, so the
// add itself rounds x*2^32 to nearest-even and the result's bit pattern is KM_MAGIC + llrint(x*2^32).  The raw bit
// patterns are summed (mod 2^64) and count*KM_MAGIC is taken off once per workgroup: two f64 ops per value instead
// of the ten of a double->int64 conversion.  Values outside that range (or NaN) take the plain conversion, biased
// by KM_MAGIC so the bookkeeping stays uniform.
#define KM_MAGIC 0x4338000000000000ull
#define KM_FAST_LIMIT 524288.0f

template <int KMAX, bool FULL>
__device__ __forceinline__ int km_assign(float w, float h, const float (&cw)[KMAX], const float (&ch)[KMAX], int k, float &bd)
{
    int best = 0;
    bd = kdist(w, h, cw[0], ch[0]);
#pragma unroll
    for (int c = 1; c < KMAX; ++c) {
        if (FULL || c < k) {
            float d = kdist(w, h, cw[c], ch[c]);
            if (d < bd) { bd = d; best = c; }
        }
    }
    return best;
}

// Assignment pre-filter.  The five IEEE divisions of km_assign are two thirds of the kernel, yet the winner is
// rarely in doubt.  The key of cluster c is a = inter * rcp(area + A_c), within 2.5 * 2^-23 (relative) of x = inter /
// (area + A_c); the IoU is q = inter / (area + A_c - inter) = x / (1 - x), increasing in x, and for x_a > x_b
// q_a / q_b = (x_a / x_b) * (1 - x_b) / (1 - x_a) >= x_a / x_b: a relative gap between two keys is at least as wide between
// the quotients (round 2: the subtraction of `inter` left the key, one operation per cluster).  The cluster index
// replaces the low four mantissa bits (another 15 ulp), and an unsigned max / median-of-three pair keeps the two
// largest keys m1 >= m2.  When m1 - m2 > 256 ulp (>= 2^-16 relative) and m1 >= 2^-5 (so q >= x >= 2^-5), the exact
// quotients differ by more than q * 2^-17 >= 2^-22, i.e. by several ulps of d = fl(1 - q): that cluster wins the exact
// comparison strictly, whatever the index order, and only its d is evaluated exactly.  Anything else (near ties, tiny
// overlaps, NaN/inf/negative keys, which the unsigned max drags into m1) takes km_assign.  So does ANY key above 1/2: with
// positive boxes and centroids inter <= min(area, A_c) <= (area + A_c) / 2, so x <= 1/2 always; a box with a negative side
// can make area + A_c negative and x anything -- x / (1 - x) has a pole at 1, beyond which a larger key is a SMALLER
// quotient (found by tools/stress_kmeans.py: a box (-0.3, 0.4) among 15 centroids went to the wrong one; the bound was +inf).  Callers bound |w|+|h| < 2^19
// and centroid areas < 2^60 so that rcp(area + A_c) stays a normal number.
#define KM_GAP_MIN 256u
#define KM_KEY_LO 0x3D000000u           /* 2^-5 */
#define KM_KEY_HI 0x3F000040u           /* just above 1/2: see below; NaN, inf and sign-bit keys compare above it too */
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned umax3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int KMAX, bool FULL>
__device__ __forceinline__ bool km_filter(float w, float h, const float (&cw)[KMAX], const float (&ch)[KMAX], int k, unsigned gap_min,
                                          int &best)
{
    const float area = w * h;
    unsigned key[KMAX];
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        key[c] = 0u;
        if (FULL || c < k) {
            const float inter = fminf(w, cw[c]) * fminf(h, ch[c]);
            const float sum = area + (cw[c] * ch[c]);
            key[c] = (__float_as_uint(inter * __builtin_amdgcn_rcpf(sum)) & ~15u) | (unsigned)c;
        }
    }
    // the two largest keys: max3 / med3 of the first three, then one med3 + one max per further key
    unsigned m1, m2;
    if (KMAX >= 3) {
        m1 = umax3(key[0], key[1], key[2]);
        m2 = umed3(key[0], key[1], key[2]);
#pragma unroll
        for (int c = 3; c < KMAX; ++c) { m2 = umed3(m1, m2, key[c]); m1 = max(m1, key[c]); }
    } else {
        m1 = 0; m2 = 0;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) { m2 = umed3(m1, m2, key[c]); m1 = max(m1, key[c]); }
    }
    best = (int)(m1 & 15u);
    return (m1 - m2 > gap_min) && (m1 - KM_KEY_LO < KM_KEY_HI - KM_KEY_LO);
}

template <bool FAST>
__device__ __forceinline__ unsigned long long fx32_biased(float v)
{
    if (FAST) return (unsigned long long)__double_as_longlong(fma((double)v, 4294967296.0, 6755399441055744.0));
    return (unsigned long long)fx32(v) + KM_MAGIC;
}

template <bool FAST, bool INERTIA = true>
__device__ __forceinline__ void km_accum(float w, float h, float bd, int best, unsigned long long *sacc, unsigned *scnt,
                                         unsigned long long &inertia, int &changed, uint8_t *lab, unsigned old)
{
    atomicAdd(&sacc[(best * 2 + 0) * KM_BLOCK], fx32_biased<FAST>(w));
    atomicAdd(&sacc[(best * 2 + 1) * KM_BLOCK], fx32_biased<FAST>(h));
    atomicAdd(&scnt[best * KM_BLOCK], 1u);
    if (INERTIA) inertia += fx32_biased<FAST>(bd);
    changed += (old != (unsigned)best);          // old label: loaded together with the box, a whole iteration ahead
    *lab = (uint8_t)best;
}

template <int KMAX, bool FULL, bool INERTIA = true>
__device__ __forceinline__ void km_one(float w, float h, const float (&cw)[KMAX], const float (&ch)[KMAX], int k,
                                       unsigned long long *sacc, unsigned *scnt, unsigned long long &inertia, int &changed,
                                       uint8_t *lab)
{
    float bd;
    const int best = km_assign<KMAX, FULL>(w, h, cw, ch, k, bd);
    if ((fabsf(w) + fabsf(h)) + fabsf(bd) < KM_FAST_LIMIT) km_accum<true, INERTIA>(w, h, bd, best, sacc, scnt, inertia, changed, lab, *lab);
    else km_accum<false, INERTIA>(w, h, bd, best, sacc, scnt, inertia, changed, lab, *lab);
}

// INERTIA = false (the iterations of dg_kmeans_iou_run before its last): the sum of distances is not formed, so the
// winner's exact distance is not evaluated either; acc[3k+1] is then written as 0.  A Lloyd iteration needs the sums,
// the counts and the changed count; the inertia is reported for the final assignment only (as sklearn does).
// Partial-sums buffer ("part"): KM_REPL rows of km_stride(k) int64, row = {sum_w[k], sum_h[k], count[k], changed,
// inertia_fx, padding to whole 128-byte lines}.  A launch ADDS into part_out (which must arrive zeroed), reads part_prev
// (the previous iteration's buffer, all-reduced when sharded; NULL = plain step with the centroids as given) and zeroes
// part_zero (the buffer the NEXT launch adds into; NULL = nobody needs it).  With three buffers in rotation no zeroing
// ever races: the buffer zeroed by launch t was last read by launch t-1.
#ifndef KM_REPL
#define KM_REPL 8
#endif
__host__ __device__ static inline int km_stride(int k) { return ((3 * k + 2) + 15) & ~15; }
__device__ __forceinline__ float km_centroid(long long sum, long long cnt)      // as k_kmeans_finish / orc_kmeans_update
{
    return (float)(((double)sum / (double)cnt) * 0x1p-32);       // == / 4294967296.0: scaling by a power of two is exact
}

#ifdef DG_KM_STAMPS
// diagnostic build only (tools/km_stamps.py): where the first wave of workgroup 0 spends a launch, in 10 ns ticks
__device__ unsigned long long km_stamps[8];
__device__ unsigned long long km_blk[1024][2];      // entry / exit time of every workgroup of the last launch
extern "C" int dg_debug_km_blocks(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(km_blk), sizeof(km_blk)) == hipSuccess ? 0 : -4; }
#define KM_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); if (blockIdx.x == 0) { km_stamps[i] = t_; km_stamps[7] = gridDim.x; } \
                              if ((i) == 0) km_blk[blockIdx.x][0] = t_; if ((i) == 5) km_blk[blockIdx.x][1] = t_; } } while (0)
extern "C" int dg_debug_km_stamps(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(km_stamps), sizeof(km_stamps)) == hipSuccess ? 0 : -4; }
#else
#define KM_STAMP(i)
#endif

template <int KMAX, bool FULL, bool INERTIA>
__global__ __launch_bounds__(KM_BLOCK) void k_kmeans_step(const float2 *__restrict__ wh, int64_t n, float *cent, int k,
                                                          uint8_t *__restrict__ labels, const long long *part_prev,
                                                          long long *part_out, long long *part_zero,
                                                          long long *__restrict__ history, int64_t slot)
{
    constexpr int COLS = 3 * KMAX + 2;
    __shared__ unsigned long long sacc_all[KMAX * 2 * KM_BLOCK];
    __shared__ unsigned scnt_all[KMAX * KM_BLOCK];
    __shared__ float2 scent[16];
    __shared__ long long s_prev[3 * 16 + 2];
    const int tid = threadIdx.x;
    const int pstride = km_stride(k), cols = 3 * k + 2;
    KM_STAMP(0);
    // Every global load the prologue and the first loop iteration need is issued up front, so that their latencies
    // overlap: the previous sums, the centroids, the first four boxes and their old labels.
    // two boxes per lane per 16-byte load, two loads in flight; an 8-byte-aligned shard start is peeled off first
    const int head = ((reinterpret_cast<uintptr_t>(wh) & 8) && n > 0) ? 1 : 0;
    const float2 *wh_head = wh;
    wh += head; n -= head;
    const int64_t npair = n >> 1;
    const float4 *wh4 = reinterpret_cast<const float4 *>(wh);
    uint8_t *lab = labels + head;
    const int64_t stride = (int64_t)gridDim.x * KM_BLOCK;
    int64_t i = (int64_t)blockIdx.x * KM_BLOCK + tid;
    bool have = i + stride < npair;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    // the old labels of the four boxes (for the changed count) travel with the boxes: two adjacent bytes per pair
    auto lab2 = [&](int64_t pair) { return (unsigned)lab[2 * pair] | ((unsigned)lab[2 * pair + 1] << 8); };
    unsigned la = 0, lb = 0;
    if (have) { a = wh4[i]; b = wh4[i + stride]; la = lab2(i); lb = lab2(i + stride); }
    long long pv[KM_REPL];
#pragma unroll
    for (int r = 0; r < KM_REPL; ++r) pv[r] = (part_prev && tid < cols) ? part_prev[r * pstride + tid] : 0;
    float2 c_old = make_float2(0.f, 0.f);
    if (tid < k) c_old = make_float2(cent[2 * tid], cent[2 * tid + 1]);
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        sacc_all[(c * 2 + 0) * KM_BLOCK + tid] = 0;
        sacc_all[(c * 2 + 1) * KM_BLOCK + tid] = 0;
        scnt_all[c * KM_BLOCK + tid] = 0;
    }
    if (tid < cols) {                                  // fold the previous iteration's replicas: one column per thread
        long long t = 0;
#pragma unroll
        for (int r = 0; r < KM_REPL; ++r) t += pv[r];
        s_prev[tid] = t;
    }
    __syncthreads();
    KM_STAMP(1);
    if (tid < 16) {
        float2 c = c_old;
        if (part_prev && tid < k) {
            const long long cnt = s_prev[2 * k + tid];
            if (cnt > 0) { c.x = km_centroid(s_prev[tid], cnt); c.y = km_centroid(s_prev[k + tid], cnt); }
        }
        scent[tid] = c;
        // bookkeeping by one workgroup.  Writing the derived centroids back in place is safe although other workgroups
        // may still be reading `cent`: the update is idempotent -- a cluster with members takes its centroid from the
        // sums alone, an empty one keeps a value this launch does not change -- so a workgroup that reads the new
        // centroids derives the same ones again.
        if (blockIdx.x == 0 && part_prev && tid < k) { cent[2 * tid] = c.x; cent[2 * tid + 1] = c.y; }
    }
    if (blockIdx.x == 0) {
        if (part_zero)
            for (int z = tid; z < KM_REPL * pstride; z += KM_BLOCK) part_zero[z] = 0;
        if (part_prev && history && tid == 0) history[slot] = s_prev[3 * k];
    }
    __syncthreads();
    float cw[KMAX], ch[KMAX];
    bool cent_ok = true;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        // wave-uniform values: back into scalar registers
        cw[c] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(scent[c < 16 ? c : 0].x)));
        ch[c] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(scent[c < 16 ? c : 0].y)));
        cent_ok = cent_ok && (fabsf(cw[c] * ch[c]) < 0x1p60f);
    }
    KM_STAMP(2);
    const unsigned gap_min = cent_ok ? KM_GAP_MIN : 0xffffffffu;     // a gap nobody reaches: exact path only
    unsigned long long *sacc = sacc_all + tid;     // lane-private column: no barrier needed until the final reduce
    unsigned *scnt = scnt_all + tid;
    unsigned long long inertia = 0;                // biased like the LDS sums: KM_MAGIC per box
    int changed = 0;

    if (head && blockIdx.x == 0 && tid == 0) {
        const float2 t = wh_head[0];
        km_one<KMAX, FULL, INERTIA>(t.x, t.y, cw, ch, k, sacc, scnt, inertia, changed, labels);
    }
    // Waves that share a SIMD are served oldest first, so the workgroup that reached a CU first would finish its (equal)
    // share first and leave the youngest alone on the SIMD at half the issue rate.  Rotating the priority every pass
    // gives every wave each level equally often (priority outranks age) and they finish together.
    unsigned prio = (blockIdx.x >> 8);
    while (have) {
        switch ((prio++) & 3u) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
        // the next pair of loads is issued before this pair is worked on
        const int64_t inext = i + 2 * stride;
        const bool more = inext + stride < npair;
        float4 na = a, nb = b;
        unsigned nla = 0, nlb = 0;
        if (more) { na = wh4[inext]; nb = wh4[inext + stride]; nla = lab2(inext); nlb = lab2(inext + stride); }
        float d0, d1, d2, d3;
        int b0, b1, b2, b3;
        const float magwh = ((fabsf(a.x) + fabsf(a.y)) + (fabsf(a.z) + fabsf(a.w))) + ((fabsf(b.x) + fabsf(b.y)) + (fabsf(b.z) + fabsf(b.w)));
        const bool f0 = km_filter<KMAX, FULL>(a.x, a.y, cw, ch, k, gap_min, b0), f1 = km_filter<KMAX, FULL>(a.z, a.w, cw, ch, k, gap_min, b1);
        const bool f2 = km_filter<KMAX, FULL>(b.x, b.y, cw, ch, k, gap_min, b2), f3 = km_filter<KMAX, FULL>(b.z, b.w, cw, ch, k, gap_min, b3);
        if ((f0 && f1) && (f2 && f3) && magwh < KM_FAST_LIMIT) {
            if (INERTIA) {
                const float2 c0 = scent[b0], c1 = scent[b1], c2 = scent[b2], c3 = scent[b3];
                d0 = kdist(a.x, a.y, c0.x, c0.y); d1 = kdist(a.z, a.w, c1.x, c1.y);
                d2 = kdist(b.x, b.y, c2.x, c2.y); d3 = kdist(b.z, b.w, c3.x, c3.y);
            } else {
                d0 = d1 = d2 = d3 = 0.f;
            }
        } else {
            b0 = km_assign<KMAX, FULL>(a.x, a.y, cw, ch, k, d0); b1 = km_assign<KMAX, FULL>(a.z, a.w, cw, ch, k, d1);
            b2 = km_assign<KMAX, FULL>(b.x, b.y, cw, ch, k, d2); b3 = km_assign<KMAX, FULL>(b.z, b.w, cw, ch, k, d3);
        }
        // one range test for the four boxes: every term is >= 0, so a sum below the limit bounds each of them
        const float mag = magwh + ((fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3)));
        if (mag < KM_FAST_LIMIT) {
            km_accum<true, INERTIA>(a.x, a.y, d0, b0, sacc, scnt, inertia, changed, lab + 2 * i, la & 255u);
            km_accum<true, INERTIA>(a.z, a.w, d1, b1, sacc, scnt, inertia, changed, lab + 2 * i + 1, la >> 8);
            km_accum<true, INERTIA>(b.x, b.y, d2, b2, sacc, scnt, inertia, changed, lab + 2 * (i + stride), lb & 255u);
            km_accum<true, INERTIA>(b.z, b.w, d3, b3, sacc, scnt, inertia, changed, lab + 2 * (i + stride) + 1, lb >> 8);
        } else {
            km_accum<false, INERTIA>(a.x, a.y, d0, b0, sacc, scnt, inertia, changed, lab + 2 * i, la & 255u);
            km_accum<false, INERTIA>(a.z, a.w, d1, b1, sacc, scnt, inertia, changed, lab + 2 * i + 1, la >> 8);
            km_accum<false, INERTIA>(b.x, b.y, d2, b2, sacc, scnt, inertia, changed, lab + 2 * (i + stride), lb & 255u);
            km_accum<false, INERTIA>(b.z, b.w, d3, b3, sacc, scnt, inertia, changed, lab + 2 * (i + stride) + 1, lb >> 8);
        }
        i = inext; a = na; b = nb; la = nla; lb = nlb; have = more;
    }
    __builtin_amdgcn_s_setprio(0);
    for (; i < npair; i += stride) {                     // what the four-box loop leaves: same scheme, two boxes
        const float4 t = wh4[i];
        const unsigned lt = lab2(i);
        float d0, d1;
        int b0, b1;
        const float magwh = (fabsf(t.x) + fabsf(t.y)) + (fabsf(t.z) + fabsf(t.w));
        const bool f0 = km_filter<KMAX, FULL>(t.x, t.y, cw, ch, k, gap_min, b0), f1 = km_filter<KMAX, FULL>(t.z, t.w, cw, ch, k, gap_min, b1);
        if (f0 && f1 && magwh < KM_FAST_LIMIT) {
            if (INERTIA) {
                const float2 c0 = scent[b0], c1 = scent[b1];
                d0 = kdist(t.x, t.y, c0.x, c0.y); d1 = kdist(t.z, t.w, c1.x, c1.y);
            } else {
                d0 = d1 = 0.f;
            }
        } else {
            b0 = km_assign<KMAX, FULL>(t.x, t.y, cw, ch, k, d0); b1 = km_assign<KMAX, FULL>(t.z, t.w, cw, ch, k, d1);
        }
        if (magwh + (fabsf(d0) + fabsf(d1)) < KM_FAST_LIMIT) {
            km_accum<true, INERTIA>(t.x, t.y, d0, b0, sacc, scnt, inertia, changed, lab + 2 * i, lt & 255u);
            km_accum<true, INERTIA>(t.z, t.w, d1, b1, sacc, scnt, inertia, changed, lab + 2 * i + 1, lt >> 8);
        } else {
            km_accum<false, INERTIA>(t.x, t.y, d0, b0, sacc, scnt, inertia, changed, lab + 2 * i, lt & 255u);
            km_accum<false, INERTIA>(t.z, t.w, d1, b1, sacc, scnt, inertia, changed, lab + 2 * i + 1, lt >> 8);
        }
    }
    if ((n & 1) && blockIdx.x == 0 && tid == 0) {
        const float2 t = wh[n - 1];
        km_one<KMAX, FULL, INERTIA>(t.x, t.y, cw, ch, k, sacc, scnt, inertia, changed, lab + (n - 1));
    }

    KM_STAMP(3);
    // Block reduction through LDS, no shuffles (17 x 64-bit wave reductions by ds_bpermute took 3.5 us per launch): the
    // accumulators already sit in LDS as [column][thread]; 16 threads sum one column's 256 entries (16 each), a second
    // step sums the 16 partials.  red[] is reused: [COLS][16] partials.
    static_assert(KM_BLOCK == 256, "16 threads x 16 entries per column");
    __shared__ unsigned long long s_inertia[KM_BLOCK];
    __shared__ unsigned s_changed[KM_BLOCK];
    __shared__ unsigned long long s_cpart[COLS][16];
    // `inertia` carries KM_MAGIC per box this thread accumulated (mod 2^64): taken off here, with the thread's own counts
    if (INERTIA) {
        unsigned mine = 0;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) mine += scnt[c * KM_BLOCK];
        inertia -= (unsigned long long)mine * KM_MAGIC;
    }
    s_inertia[tid] = INERTIA ? inertia : 0ull;
    s_changed[tid] = (unsigned)changed;
    __syncthreads();
    {
        const int sub = tid & 15;
        for (int col = tid >> 4; col < COLS; col += KM_BLOCK / 16) {
            unsigned long long t = 0;
            if (col < 2 * KMAX) {
                const unsigned long long *src = sacc_all + ((col % KMAX) * 2 + col / KMAX) * KM_BLOCK;
#pragma unroll
                for (int j = 0; j < 16; ++j) t += src[j * 16 + sub];
            } else if (col < 3 * KMAX) {
                const unsigned *src = scnt_all + (col - 2 * KMAX) * KM_BLOCK;
#pragma unroll
                for (int j = 0; j < 16; ++j) t += src[j * 16 + sub];
            } else if (col == 3 * KMAX) {
#pragma unroll
                for (int j = 0; j < 16; ++j) t += s_changed[j * 16 + sub];
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) t += s_inertia[j * 16 + sub];
            }
            s_cpart[col][sub] = t;
        }
    }
    __syncthreads();
    KM_STAMP(4);
    if (tid < COLS) {
        unsigned long long v = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) v += s_cpart[tid][j];
        const int grp = tid / KMAX, c = tid - grp * KMAX;
        // the fixed-point sums carry KM_MAGIC per box (mod 2^64): taken off with the cluster's count / the box count
        if (grp < 2) {
            unsigned long long cnt = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) cnt += s_cpart[2 * KMAX + c][j];
            v -= cnt * KM_MAGIC;
        }
        // this workgroup's row into its replica: 64-bit no-return atomics (device scope); the next launch on the stream
        // reads the sums with plain loads
        const int col = (grp < 3) ? (c < k ? grp * k + c : -1) : 3 * k + c;
        if (col >= 0 && v != 0)
            atomicAdd(reinterpret_cast<unsigned long long *>(part_out) + (blockIdx.x % KM_REPL) * pstride + col, v);
    }
    KM_STAMP(5);
}

// folds the replicas of the last iteration (all-reduced when sharded): acc_out[3k+2] = the sums (may be NULL),
// centroid = sum/count (int64 fixed point 2^32 -> fp32; an empty cluster keeps its centroid), history[slot] = changed
__global__ __launch_bounds__(64) void k_kmeans_finish(const long long *__restrict__ part, int k, float *__restrict__ cent,
                                                      long long *__restrict__ acc_out, long long *__restrict__ history, int64_t slot)
{
    __shared__ long long s[3 * 16 + 2];
    const int tid = threadIdx.x, stride = km_stride(k), cols = 3 * k + 2;
    if (tid < cols) {
        long long t = 0;
        for (int r = 0; r < KM_REPL; ++r) t += part[r * stride + tid];
        s[tid] = t;
        if (acc_out) acc_out[tid] = t;
    }
    __syncthreads();
    if (cent && tid < k) {
        const long long cnt = s[2 * k + tid];
        if (cnt > 0) { cent[2 * tid] = km_centroid(s[tid], cnt); cent[2 * tid + 1] = km_centroid(s[k + tid], cnt); }
    }
    if (tid == 0 && history) history[slot] = s[3 * k];
}

__global__ __launch_bounds__(256) void k_labels_widen(const uint8_t *__restrict__ l, int64_t n, int32_t *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = l[i];
}

// Zeroes small buffers of 64-bit words (the partial-sums buffers of the loops below).  A kernel, not hipMemsetAsync: captured in a HIP graph (ROCm 7.2) the memset node of
// this chunk was not carried out again on the graph's SECOND replay when the device had been synchronised between the eager
// run and the capture -- the buffers kept the previous replay's sums and the centroids collapsed
// (test_kmeans_sharded_loop_library_comm_vs_oracle found it once a fit stopped copying its labels to the host).
__global__ void k_km_zero(long long *__restrict__ p, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}

static inline void km_zero(void *p, size_t bytes, hipStream_t st)          // bytes: a multiple of 8
{
    hipLaunchKernelGGL(k_km_zero, dim3(bytes > 65536 ? 16 : 1), dim3(256), 0, st, (long long *)p, (int)(bytes / 8));
}

static inline int km_kmax(int64_t k) { return k <= 5 ? 5 : (k <= 9 ? 9 : 16); }
// One resident round: as many workgroups as the device holds at once for this variant (registers and LDS decide),
// so that no workgroup starts on a mostly idle chip after the others have finished.
template <typename K>
static int km_resident_blocks(K kernel)
{
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, KM_BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    const int b = per_cu * cus;
    return b > KM_MAXBLOCKS ? KM_MAXBLOCKS : b;
}
static inline int km_blocks(int64_t n, int resident)
{
    // at least 4 boxes per thread (one pass of the four-box loop); measured on 1.25 M boxes: 16 per thread 10.1 us, 8: 9.0, 4: 8.6
    static const int bpt = getenv("DG_KM_BPT") ? atoi(getenv("DG_KM_BPT")) : 4;
    int64_t b = dg_cdiv(n, KM_BLOCK * bpt);
    return (int)(b < 1 ? 1 : (b > resident ? resident : b));
}

extern "C" size_t dg_kmeans_part_elems(int64_t k)
{
    if (k <= 0 || k > 16) return 0;
    return (size_t)KM_REPL * km_stride((int)k);
}
// workspace of dg_kmeans_iou_step / dg_kmeans_iou_run: three partial-sums buffers in rotation
extern "C" size_t dg_kmeans_workspace_bytes(int64_t n, int64_t k)
{
    if (n < 0 || k <= 0 || k > 16) return 0;
    return 3 * dg_kmeans_part_elems(k) * 8;
}

// one launch: [centroids <- part_prev;] step; part_out += this shard's rows; part_zero <- 0
static int km_launch(const float *wh, int64_t n, float *cent, int64_t k, uint8_t *labels, const int64_t *part_prev, int64_t *part_out,
                     int64_t *part_zero, int64_t *history, int64_t slot, bool inertia, hipStream_t st)
{
    const int kmax = km_kmax(k);
#define KM_LAUNCH1(KM, FULL, INERTIA)                                                                                           \
    do {                                                                                                                        \
        static const int resident = km_resident_blocks(k_kmeans_step<KM, FULL, INERTIA>);                                       \
        hipLaunchKernelGGL((k_kmeans_step<KM, FULL, INERTIA>), dim3(km_blocks(n, resident)), dim3(KM_BLOCK), 0, st,             \
                           (const float2 *)wh, n, cent, (int)k, labels, (const long long *)part_prev, (long long *)part_out,    \
                           (long long *)part_zero, (long long *)history, slot);                                                 \
    } while (0)
#define KM_LAUNCH(KM, FULL)                                                                                                     \
    do {                                                                                                                        \
        if (inertia) KM_LAUNCH1(KM, FULL, true);                                                                                \
        else KM_LAUNCH1(KM, FULL, false);                                                                                       \
    } while (0)
    // k == 5 and k == 9 (the YOLOv2 anchor counts) get kernels without the per-cluster `c < k` branches
    if (k == 5) KM_LAUNCH(5, true);
    else if (kmax == 5) KM_LAUNCH(5, false);
    else if (k == 9) KM_LAUNCH(9, true);
    else if (kmax == 9) KM_LAUNCH(9, false);
    else KM_LAUNCH(16, false);
#undef KM_LAUNCH
#undef KM_LAUNCH1
    DG_CHECK_LAUNCH("k_kmeans_step");
    return DG_OK;
}

static int km_check(const char *fn, const float *wh, int64_t n, const float *cent, int64_t k, const uint8_t *labels)
{
    DG_REQUIRE(n >= 0 && k >= 1 && k <= 16, "%s: need n >= 0 and 1 <= k <= 16", fn);
    DG_REQUIRE(cent, "%s: null centroids", fn);
    DG_REQUIRE(n == 0 || (wh && labels), "%s: null pointer", fn);
    DG_REQUIRE((((uintptr_t)wh) & 7) == 0, "%s: wh must be 8-byte aligned", fn);
    return DG_OK;
}
static int km_check_ws(const char *fn, int64_t n, int64_t k, const void *ws, size_t ws_bytes)
{
    DG_REQUIRE(ws && (((uintptr_t)ws) & 15) == 0, "%s: workspace null or misaligned", fn);
    if (ws_bytes < dg_kmeans_workspace_bytes(n, k)) { dg_set_error("%s: workspace too small", fn); return DG_ENOMEM; }
    return DG_OK;
}
static int km_finish(const int64_t *part, int64_t k, float *cent, int64_t *acc_out, int64_t *history, int64_t slot, hipStream_t st)
{
    hipLaunchKernelGGL(k_kmeans_finish, dim3(1), dim3(64), 0, st, (const long long *)part, (int)k, cent, (long long *)acc_out,
                       (long long *)history, slot);
    DG_CHECK_LAUNCH("k_kmeans_finish");
    return DG_OK;
}

extern "C" int dg_kmeans_iou_step(const float *wh, int64_t n, const float *cent, int64_t k, uint8_t *labels, int64_t *acc,
                                  void *ws, size_t ws_bytes, dg_stream_t s)
{
    int rc = km_check("kmeans_step", wh, n, cent, k, labels);
    if (rc) return rc;
    DG_REQUIRE(acc, "kmeans_step: null acc");
    if ((rc = km_check_ws("kmeans_step", n, k, ws, ws_bytes))) return rc;
    hipStream_t st = (hipStream_t)s;
    int64_t *part = (int64_t *)ws;
    // the workspace arrives with arbitrary contents
    km_zero(part, dg_kmeans_part_elems(k) * 8, st);
    if ((rc = km_launch(wh, n, const_cast<float *>(cent), k, labels, nullptr, part, nullptr, nullptr, 0, true, st))) return rc;
    return km_finish(part, k, nullptr, acc, nullptr, 0, st);
}

// One iteration of the SHARDED Lloyd loop as one launch (declaration: include/detgeom.h).
extern "C" int dg_kmeans_iou_iter(const float *wh, int64_t n, float *cent, int64_t k, uint8_t *labels, const int64_t *part_prev,
                                  int64_t *part_out, int64_t *part_zero, int want_inertia, int64_t *history, int64_t slot,
                                  dg_stream_t s)
{
    int rc = km_check("kmeans_iter", wh, n, cent, k, labels);
    if (rc) return rc;
    DG_REQUIRE(part_out && part_out != part_prev && part_zero != part_out && (part_zero == nullptr || part_zero != part_prev),
               "kmeans_iter: part_out must be given and the three buffers must differ");
    return km_launch(wh, n, cent, k, labels, part_prev, part_out, part_zero, history, slot, want_inertia != 0, (hipStream_t)s);
}

extern "C" int dg_kmeans_finish(const int64_t *part, int64_t k, float *cent, int64_t *acc_out, int64_t *history, int64_t slot,
                                dg_stream_t s)
{
    DG_REQUIRE(k >= 1 && k <= 16 && part, "kmeans_finish: bad argument");
    return km_finish(part, k, cent, acc_out, history, slot, (hipStream_t)s);
}

// `iters` Lloyd iterations back to back on one device (one launch per iteration, no host round trip);
// history[slot0 + t] = changed count of iteration t.  A converged iteration reproduces the same centroids, so
// running past convergence is idempotent and n_iter_ is read from the history afterwards.
extern "C" int dg_kmeans_iou_run(const float *wh, int64_t n, float *cent, int64_t k, uint8_t *labels, int64_t *acc,
                                 int64_t *history, int64_t slot0, int64_t iters, void *ws, size_t ws_bytes, dg_stream_t s)
{
    DG_REQUIRE(iters >= 0, "kmeans_run: iters < 0");
    int rc = km_check("kmeans_run", wh, n, cent, k, labels);
    if (rc) return rc;
    DG_REQUIRE(acc, "kmeans_run: null acc");
    if ((rc = km_check_ws("kmeans_run", n, k, ws, ws_bytes))) return rc;
    if (iters == 0) return DG_OK;
    hipStream_t st = (hipStream_t)s;
    const size_t pe = dg_kmeans_part_elems(k);
    int64_t *part[3] = {(int64_t *)ws, (int64_t *)ws + pe, (int64_t *)ws + 2 * pe};
    km_zero(ws, 3 * pe * 8, st);
    for (int64_t t = 0; t < iters; ++t) {
        // launch t: centroids <- sums of t-1 (and history[slot0+t-1]); adds into buffer t%3; zeroes buffer (t+1)%3
        rc = km_launch(wh, n, cent, k, labels, t ? part[(t - 1) % 3] : nullptr, part[t % 3], part[(t + 1) % 3], history, slot0 + t - 1,
                       t + 1 == iters, st);                                        // inertia: last iteration only
        if (rc) return rc;
    }
    return km_finish(part[(iters - 1) % 3], k, cent, acc, history, slot0 + iters - 1, st);
}

// ---- the sharded loop issued by the library: [launch; RCCL all-reduce of that launch's buffer] x iters; finish -------------
static int km_sharded_issue(dg_comm_t *c, const float *wh, int64_t n, float *cent, int64_t k, uint8_t *labels, int64_t *acc,
                            int64_t *history, int64_t slot0, int64_t iters, void *ws, hipStream_t st)
{
    const size_t pe = dg_kmeans_part_elems(k);
    int64_t *part[3] = {(int64_t *)ws, (int64_t *)ws + pe, (int64_t *)ws + 2 * pe};
    km_zero(ws, 3 * pe * 8, st);
    int rc;
    for (int64_t t = 0; t < iters; ++t) {
        // same rotation as dg_kmeans_iou_run; the buffer launch t added into is summed over the ranks before launch t+1 folds it
        if ((rc = km_launch(wh, n, cent, k, labels, t ? part[(t - 1) % 3] : nullptr, part[t % 3], part[(t + 1) % 3], history,
                            slot0 + t - 1, t + 1 == iters, st)))
            return rc;
        if ((rc = dg_comm_allreduce_i64(c, part[t % 3], (int64_t)pe, st))) return rc;
    }
    return km_finish(part[(iters - 1) % 3], k, cent, acc, history, slot0 + iters - 1, st);
}

extern "C" int dg_kmeans_iou_run_sharded(dg_comm_t *c, const float *wh, int64_t n, float *cent, int64_t k, uint8_t *labels,
                                         int64_t *acc, int64_t *history, int64_t slot0, int64_t iters, void *ws, size_t ws_bytes,
                                         int use_graph, dg_stream_t s)
{
    DG_REQUIRE(c, "kmeans_run_sharded: null communicator");
    DG_REQUIRE(iters >= 0, "kmeans_run_sharded: iters < 0");
    int rc = km_check("kmeans_run_sharded", wh, n, cent, k, labels);
    if (rc) return rc;
    DG_REQUIRE(acc, "kmeans_run_sharded: null acc");
    if ((rc = km_check_ws("kmeans_run_sharded", n, k, ws, ws_bytes))) return rc;
    if (iters == 0) return DG_OK;
    hipStream_t st = (hipStream_t)s;
    if (!use_graph || c->graph_state < 0) return km_sharded_issue(c, wh, n, cent, k, labels, acc, history, slot0, iters, ws, st);

    const uint64_t key[12] = {(uint64_t)(uintptr_t)wh, (uint64_t)n, (uint64_t)(uintptr_t)cent, (uint64_t)k, (uint64_t)(uintptr_t)labels,
                              (uint64_t)(uintptr_t)acc, (uint64_t)(uintptr_t)history, (uint64_t)slot0, (uint64_t)iters,
                              (uint64_t)(uintptr_t)ws, (uint64_t)(uintptr_t)st, 0};
    const bool same = c->graph_state > 0 && memcmp(key, c->graph_key, sizeof(key)) == 0;
    if (same && c->graph_state == 2) {
        if (hipGraphLaunch((hipGraphExec_t)c->graph_exec, st) != hipSuccess) { dg_set_error("kmeans_run_sharded: hipGraphLaunch failed"); return DG_EHIP; }
        return DG_OK;
    }
    if (!same) {
        // first call with these arguments: issued eagerly (RCCL sets up its channels on first use, outside any capture)
        if (c->graph_exec) { (void)hipGraphExecDestroy((hipGraphExec_t)c->graph_exec); c->graph_exec = nullptr; }
        memcpy(c->graph_key, key, sizeof(key));
        c->graph_state = 1;
        return km_sharded_issue(c, wh, n, cent, k, labels, acc, history, slot0, iters, ws, st);
    }
    // second call with the same arguments: capture the chunk, then replay it
    hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c->graph_state = -1;
        dg_set_error("kmeans_run_sharded: stream capture refused (%s), issuing eagerly", hipGetErrorString(e));
        return km_sharded_issue(c, wh, n, cent, k, labels, acc, history, slot0, iters, ws, st);
    }
    rc = km_sharded_issue(c, wh, n, cent, k, labels, acc, history, slot0, iters, ws, st);
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(st, &g);
    hipGraphExec_t ex = nullptr;
    if (rc == DG_OK && e == hipSuccess && g) e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (g) (void)hipGraphDestroy(g);
    if (rc != DG_OK) {
        // a launch or an RCCL call failed WHILE capturing: the communicator may be in an error state, so nothing is re-issued on
        // it -- the error goes to the caller (dg_last_error holds its text), and the handle stays eager from now on
        (void)hipGetLastError();
        if (ex) (void)hipGraphExecDestroy(ex);
        c->graph_state = -1;
        return rc;
    }
    if (e != hipSuccess || !ex) {
        (void)hipGetLastError();
        c->graph_state = -1;
        dg_set_error("kmeans_run_sharded: capture failed (%s), issuing eagerly", hipGetErrorString(e));
        return km_sharded_issue(c, wh, n, cent, k, labels, acc, history, slot0, iters, ws, st);   // nothing ran during the capture
    }
    c->graph_exec = ex;
    c->graph_state = 2;
    if (hipGraphLaunch(ex, st) != hipSuccess) { dg_set_error("kmeans_run_sharded: hipGraphLaunch failed"); return DG_EHIP; }
    return DG_OK;
}

// 0 none, 1 armed (next call with the same arguments captures), 2 replaying a captured chunk, -1 capture failed: eager for good
extern "C" int dg_comm_graph_state(const dg_comm_t *c) { return c ? c->graph_state : 0; }

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

// ================================================================ the prepared fit: Lloyd iterations over CELLS of boxes
// NO COUNTERPART IN THE REFERENCE (like the whole 1-IoU k-means; spec: oracle/oracle.c orc_kmeans_*).
// The label of a box is a function of (w, h) and the centroids alone, so boxes that lie close together in (w, h) almost
// always share it.  dg_kmeans_prepare groups the boxes ONCE per fit by a cell of a log-spaced grid -- the float's exponent
// and its top five mantissa bits, per coordinate: 32 cells per octave over [2^-5, 1), 162 x 162 cells with the catch-all
// cells for everything else -- with a counting sort (per-workgroup LDS histograms: device-scope atomics are far too slow
// on this part), cuts every cell into PIECES of at most 256 boxes and sums each piece's fixed-point w and h.  An iteration
// (k_kc_iter) then looks at pieces: from the cell's bounds it bounds every centroid's IoU over the whole cell from below
// and above (IoU = I / (a + A - I) grows with the overlap I and falls with the box area a); if one centroid's lower bound
// beats every other's upper bound by more than 2^-20 -- an order of magnitude more than the rounding of the fp32 distances
// the specification compares -- every box of the piece gets that label and the piece's precomputed sums are added as
// they are: integer sums, so the totals are bit for bit those of a pass over the boxes.  Only the pieces of ambiguous
// cells (and of the catch-all cells) are walked box by box with the exact assignment of k_kmeans_step.  On config 4 about a
// fifth of the boxes is walked per iteration; the rest costs one table-free decision per 256 boxes.
// Labels live in sorted order inside the workspace (per piece while it is decided as a whole, per box otherwise), which
// is all the `changed` count needs; the caller's `labels` array, the inertia and acc come from one ordinary
// k_kmeans_step at the last iteration of a run (same centroids, same sums), so nothing is ever un-sorted.
#define KC_N 162
#define KC_NC (KC_N * KC_N)
#define KC_G 256
#define KC_PIECE 256
#define KC_WG_PIECES 64
#define KC_MAGIC 0x4b43656c6c733033ull
#define KC_PER_BOX 255                                            // plab: the piece's boxes carry their own labels (slab)
#define KC_UNSET 254                                              // plab: no box of the piece has a label yet (a label no centroid has: every comparison counts as changed)
#define KC_Q_COUNT (1 << 24)                                      // walk-queue entry: the piece is decided, count its changed labels only
struct KcHdr {
    unsigned long long magic;
    const void *wh;
    long long n;
    int k, npieces;
    long long pad[4];
};
struct KcWs {
    long long *part[4];
    KcHdr *hdr;
    float2 *swh;
    uint8_t *slab, *plab;
    int *counts, *tot, *cell_start, *piece_first, *pcell, *pstart, *pcnt;
    long long *psum;
    float4 *pbox;
    long long pmax;
    size_t total;
};
static KcWs kc_layout(int64_t n, int64_t k, void *base)
{
    KcWs w;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p + off; off += (bytes + 255) & ~(size_t)255; return r; };
    const size_t pe = dg_kmeans_part_elems(k);
    for (int i = 0; i < 4; ++i) w.part[i] = (long long *)take(pe * 8);
    w.hdr = (KcHdr *)take(sizeof(KcHdr));
    w.swh = (float2 *)take((size_t)n * 8);
    w.slab = (uint8_t *)take((size_t)n);
    w.pmax = n / KC_PIECE + KC_NC + 1;
    w.plab = (uint8_t *)take((size_t)w.pmax);
    w.counts = (int *)take((size_t)KC_G * KC_NC * 4);
    w.tot = (int *)take((size_t)KC_NC * 4);
    w.cell_start = (int *)take((size_t)(KC_NC + 1) * 4);
    w.piece_first = (int *)take((size_t)(KC_NC + 1) * 4);
    w.pcell = (int *)take((size_t)w.pmax * 4);
    w.pstart = (int *)take((size_t)w.pmax * 4);
    w.pcnt = (int *)take((size_t)w.pmax * 4);
    w.psum = (long long *)take((size_t)w.pmax * 16);
    w.pbox = (float4 *)take((size_t)w.pmax * 16);
    w.total = off;
    return w;
}

__device__ __forceinline__ int kc_cell1(float x)
{
    const unsigned b = __float_as_uint(x);
    if (!(x > 0.0f) || b >= 0x7f800000u) return 0;               // zero, negative, NaN, inf: the catch-all cell
    const int e = (int)(b >> 18) - 3904 + 1;                     // 3904 = (127 - 5) << 5: 2^-5
    return e < 1 ? 0 : (e > KC_N - 1 ? KC_N - 1 : e);
}
__device__ __forceinline__ int kc_cell(float2 b) { return kc_cell1(b.y) * KC_N + kc_cell1(b.x); }
__device__ __forceinline__ float kc_edge(int idx) { return __uint_as_float((unsigned)(idx - 1 + 3904) << 18); }   // lower edge of cell idx in 1..160; idx + 1: upper

// per-workgroup histogram of the cells of a contiguous slice of the boxes (LDS atomics), written wg-major
__global__ __launch_bounds__(1024) void k_kc_hist(const float2 *__restrict__ wh, long long n, int *__restrict__ counts)
{
    extern __shared__ int kc_h[];
    const int tid = threadIdx.x, g = blockIdx.x;
    for (int c = tid; c < KC_NC; c += 1024) kc_h[c] = 0;
    __syncthreads();
    const long long S = (n + KC_G - 1) / KC_G, lo = g * S, hi = lo + S < n ? lo + S : n;
    for (long long i = lo + tid; i < hi; i += 1024) atomicAdd(&kc_h[kc_cell(wh[i])], 1);
    __syncthreads();
    for (int c = tid; c < KC_NC; c += 1024) counts[(size_t)g * KC_NC + c] = kc_h[c];
}
// per cell: the slices' counts become their exclusive prefix over the slices, the total goes to tot[]
__global__ __launch_bounds__(256) void k_kc_colscan(int *__restrict__ counts, int *__restrict__ tot)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= KC_NC) return;
    int run = 0;
    for (int g = 0; g < KC_G; ++g) { const int v = counts[(size_t)g * KC_NC + c]; counts[(size_t)g * KC_NC + c] = run; run += v; }
    tot[c] = run;
}
// one workgroup: where every cell starts in the sorted order, and its first piece.  Everything goes through LDS so that memory
// sees coalesced accesses only: a thread reading its 26 consecutive cells from memory, three times over, took 52 us of
// dependent trips; the same thread writing its 2 x 26 results straight to memory 30 us (every wave store touched 64 lines)
__global__ __launch_bounds__(1024) void k_kc_scan(const int *__restrict__ tot, int *__restrict__ cell_start, int *__restrict__ piece_first,
                                                  KcHdr *__restrict__ hdr, const void *wh, long long n, int k)
{
    extern __shared__ int kc_h[];                                // KC_NC totals, then one exclusive prefix after the other
    __shared__ unsigned long long s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER = (KC_NC + 1023) / 1024;
    for (int c = tid; c < KC_NC; c += 1024) kc_h[c] = tot[c];
    __syncthreads();
    int t[PER];
    unsigned long long ab = 0;                                   // boxes in the low word, pieces in the high one: one scan for both
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = tid * PER + j;
        t[j] = c < KC_NC ? kc_h[c] : 0;
        ab += (unsigned long long)(unsigned)t[j] | ((unsigned long long)(unsigned)((t[j] + KC_PIECE - 1) / KC_PIECE) << 32);
    }
    unsigned long long inc = ab;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    unsigned long long base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { if (w < wave) base += s_w[w]; total += s_w[w]; }
    const unsigned long long ex = base + inc - ab;
    int ra = (int)(unsigned)ex, rb = (int)(unsigned)(ex >> 32);
#pragma unroll
    for (int j = 0; j < PER; ++j) { const int c = tid * PER + j; if (c < KC_NC) kc_h[c] = ra; ra += t[j]; }
    __syncthreads();
    for (int c = tid; c < KC_NC; c += 1024) cell_start[c] = kc_h[c];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) { const int c = tid * PER + j; if (c < KC_NC) kc_h[c] = rb; rb += (t[j] + KC_PIECE - 1) / KC_PIECE; }
    __syncthreads();
    for (int c = tid; c < KC_NC; c += 1024) piece_first[c] = kc_h[c];
    if (tid == 1023) {
        cell_start[KC_NC] = (int)(unsigned)total; piece_first[KC_NC] = (int)(unsigned)(total >> 32);
        hdr->magic = KC_MAGIC; hdr->wh = wh; hdr->n = n; hdr->k = k; hdr->npieces = (int)(unsigned)(total >> 32);
    }
}
__global__ __launch_bounds__(256) void k_kc_pieces(const int *__restrict__ tot, const int *__restrict__ cell_start,
                                                   const int *__restrict__ piece_first, int *__restrict__ pcell, int *__restrict__ pstart,
                                                   int *__restrict__ pcnt, uint8_t *__restrict__ plab, int plab0)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= KC_NC) return;
    const int t = tot[c], s = cell_start[c], p0 = piece_first[c];
    for (int j = 0; j * KC_PIECE < t; ++j) {
        pcell[p0 + j] = c; pstart[p0 + j] = s + j * KC_PIECE; pcnt[p0 + j] = t - j * KC_PIECE < KC_PIECE ? t - j * KC_PIECE : KC_PIECE;
        plab[p0 + j] = (uint8_t)plab0;                            // KC_PER_BOX: the caller's labels, copied into slab; KC_UNSET: none yet
    }
}
// boxes (and the caller's current labels) into cell order: the slice's cursors start where its prefix says
__global__ __launch_bounds__(1024) void k_kc_scatter(const float2 *__restrict__ wh, const uint8_t *__restrict__ labels, long long n,
                                                     const int *__restrict__ counts, const int *__restrict__ cell_start,
                                                     float2 *__restrict__ swh, uint8_t *__restrict__ slab)
{
    extern __shared__ int kc_h[];
    const int tid = threadIdx.x, g = blockIdx.x;
    for (int c = tid; c < KC_NC; c += 1024) kc_h[c] = cell_start[c] + counts[(size_t)g * KC_NC + c];
    __syncthreads();
    const long long S = (n + KC_G - 1) / KC_G, lo = g * S, hi = lo + S < n ? lo + S : n;
    for (long long i = lo + tid; i < hi; i += 1024) {
        const float2 b = wh[i];
        const int at = atomicAdd(&kc_h[kc_cell(b)], 1);
        swh[at] = b;
        if (labels) slab[at] = labels[i];
    }
}
// sixteen lanes per piece, four pieces per wave: the piece's fixed-point sums of w and h (the values k_kmeans_step would add box
// by box) and its bounding box.  (One wave per piece spent its time in 6-step wave reductions of two 64-bit sums and four
// floats per 256 boxes: 34 us for 10 M boxes; four steps here, shared by four pieces.)
__global__ __launch_bounds__(256) void k_kc_piecesums(const float2 *__restrict__ swh, const int *__restrict__ pstart,
                                                      const int *__restrict__ pcnt, const KcHdr *__restrict__ hdr,
                                                      long long *__restrict__ psum, float4 *__restrict__ pbox)
{
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const bool live = p < hdr->npieces;
    const int s = live ? pstart[p] : 0, c = live ? pcnt[p] : 0;
    long long sw = 0, sh = 0;
    float wlo = INFINITY, whi = -INFINITY, hlo = INFINITY, hhi = -INFINITY;
    bool bad = false;
#pragma unroll 4
    for (int j = sub; j < c; j += 16) {
        const float2 b = swh[s + j];
        sw += fx32(b.x); sh += fx32(b.y);
        wlo = fminf(wlo, b.x); whi = fmaxf(whi, b.x); hlo = fminf(hlo, b.y); hhi = fmaxf(hhi, b.y);
        bad |= !(b.x > 0.0f) || !(b.y > 0.0f) || !(b.x < KM_FAST_LIMIT) || !(b.y < KM_FAST_LIMIT);
    }
    unsigned badw = bad ? 1u : 0u;
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) {                            // xor < 16 stays inside the piece's sixteen lanes
        sw += __shfl_xor(sw, d, 64); sh += __shfl_xor(sh, d, 64);
        wlo = fminf(wlo, __shfl_xor(wlo, d, 64)); whi = fmaxf(whi, __shfl_xor(whi, d, 64));
        hlo = fminf(hlo, __shfl_xor(hlo, d, 64)); hhi = fmaxf(hhi, __shfl_xor(hhi, d, 64));
        badw |= __shfl_xor(badw, d, 64);
    }
    if (live && sub == 0) {
        psum[2 * (size_t)p] = sw; psum[2 * (size_t)p + 1] = sh;
        // the piece's own bounding box in (w, h): what the decision bounds the IoUs over; a piece with a box that is not a
        // positive finite number of moderate size is never decided as a whole (whi < wlo says so)
        pbox[p] = badw ? make_float4(1.f, 0.f, 1.f, 0.f) : make_float4(wlo, whi, hlo, hhi);
    }
}

// The decision for one piece: the label all its boxes get, or 255 (walk the boxes).  For a fixed centroid the IoU of a box
// (w, h) rises with w up to the centroid's width and falls beyond it, whatever h is, and the same in h: over the piece's
// bounding box its maximum sits at the point nearest the centroid and its minimum at one of the four corners.  Those five
// values per centroid are evaluated in fp32 (a few 1e-7 off at most); one centroid's minimum must beat every other's
// maximum by 4e-6 -- the fp32 distances the specification compares are then ordered the same way for every box of the
// piece, with room to spare for their own rounding (2^-23) -- or the piece is walked.
__device__ __forceinline__ float kc_iou(float w, float h, float cw, float ch)
{
    const float inter = fminf(w, cw) * fminf(h, ch);
    return inter / ((w * h + cw * ch) - inter);
}
template <int KMAX, bool FULL>
__device__ __forceinline__ int kc_decide(float4 bb, const float (&cw)[KMAX], const float (&ch)[KMAX], int k)
{
    const float wlo = bb.x, whi = bb.y, hlo = bb.z, hhi = bb.w;
    if (!(wlo <= whi)) return 255;
    float best_lo = -1.0f, top_hi = -1.0f, second_hi = -1.0f;
    int best = 0, top = 0;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        if (FULL || c < k) {
            const float qhi = kc_iou(fminf(fmaxf(cw[c], wlo), whi), fminf(fmaxf(ch[c], hlo), hhi), cw[c], ch[c]);
            const float qlo = fminf(fminf(kc_iou(wlo, hlo, cw[c], ch[c]), kc_iou(whi, hlo, cw[c], ch[c])),
                                    fminf(kc_iou(wlo, hhi, cw[c], ch[c]), kc_iou(whi, hhi, cw[c], ch[c])));
            if (qlo > best_lo) { best_lo = qlo; best = c; }
            if (qhi > top_hi) { second_hi = top_hi; top_hi = qhi; top = c; }
            else if (qhi > second_hi) second_hi = qhi;
        }
    }
    const float rival = (top == best) ? second_hi : top_hi;       // the largest maximum among the OTHER centroids
    return (best_lo > rival + 4e-6f) ? best : 255;
}

template <int KMAX, bool FULL>
__global__ __launch_bounds__(KM_BLOCK) void k_kc_iter(const KcHdr *__restrict__ hdr, const float2 *__restrict__ swh,
                                                      uint8_t *__restrict__ slab, uint8_t *__restrict__ plab,
                                                      const int *__restrict__ pcell, const int *__restrict__ pstart,
                                                      const int *__restrict__ pcnt, const long long *__restrict__ psum,
                                                      const float4 *__restrict__ pbox, float *cent, int k, const long long *part_prev, long long *part_out, long long *part_zero,
                                                      long long *__restrict__ history, int64_t slot, int write_cent, int pmax)
{
    constexpr int COLS = 3 * KMAX + 2;
    __shared__ unsigned long long sacc_all[KMAX * 2 * KM_BLOCK];
    __shared__ unsigned scnt_all[KMAX * KM_BLOCK];
    __shared__ float2 scent[16];
    __shared__ long long s_prev[3 * 16 + 2];
    __shared__ int s_queue[KC_WG_PIECES], s_qstart[KC_WG_PIECES], s_qcnt[KC_WG_PIECES], s_qn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pstride = km_stride(k), cols = 3 * k + 2;
    const int P = pmax;                                          // slots past the last piece hold pcnt == 0
    // 64 pieces per workgroup, decided by its first wave (one per lane), interleaved over the grid so that every workgroup
    // gets pieces from all over the (w, h) plane; the pieces that have to be walked are then shared by its four waves
    const int p = (tid < KC_WG_PIECES) ? tid * (int)gridDim.x + (int)blockIdx.x : P;
    int pst = 0, pc = 0;
    long long ps0 = 0, ps1 = 0;
    unsigned prevl = 255;
    float4 bb = make_float4(1.f, 0.f, 1.f, 0.f);
    if (p < P) { bb = pbox[p]; pst = pstart[p]; pc = pcnt[p]; ps0 = psum[2 * (size_t)p]; ps1 = psum[2 * (size_t)p + 1]; prevl = plab[p]; }
    const bool valid = pc > 0;
    long long pv[KM_REPL];
#pragma unroll
    for (int r = 0; r < KM_REPL; ++r) pv[r] = (part_prev && tid < cols) ? part_prev[r * pstride + tid] : 0;
    float2 c_old = make_float2(0.f, 0.f);
    if (tid < k) c_old = make_float2(cent[2 * tid], cent[2 * tid + 1]);
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        sacc_all[(c * 2 + 0) * KM_BLOCK + tid] = 0;
        sacc_all[(c * 2 + 1) * KM_BLOCK + tid] = 0;
        scnt_all[c * KM_BLOCK + tid] = 0;
    }
    if (tid == 0) s_qn = 0;
    if (tid < cols) {
        long long t = 0;
#pragma unroll
        for (int r = 0; r < KM_REPL; ++r) t += pv[r];
        s_prev[tid] = t;
    }
    __syncthreads();
    if (tid < 16) {
        float2 c = c_old;
        if (part_prev && tid < k) {
            const long long cnt = s_prev[2 * k + tid];
            if (cnt > 0) { c.x = km_centroid(s_prev[tid], cnt); c.y = km_centroid(s_prev[k + tid], cnt); }
        }
        scent[tid] = c;
        if (write_cent && blockIdx.x == 0 && part_prev && tid < k) { cent[2 * tid] = c.x; cent[2 * tid + 1] = c.y; }   // idempotent, see k_kmeans_step
    }
    if (blockIdx.x == 0) {
        if (part_zero)
            for (int z = tid; z < KM_REPL * pstride; z += KM_BLOCK) part_zero[z] = 0;
        if (part_prev && history && tid == 0) history[slot] = s_prev[3 * k];
    }
    __syncthreads();
    float cw[KMAX], ch[KMAX];
    bool cent_ok = true;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        cw[c] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(scent[c < 16 ? c : 0].x)));
        ch[c] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(scent[c < 16 ? c : 0].y)));
        if (FULL || c < k) cent_ok = cent_ok && (cw[c] > 0.0f) && (ch[c] > 0.0f) && (cw[c] * ch[c] < 0x1p60f);
    }
    const unsigned gap_min = cent_ok ? KM_GAP_MIN : 0xffffffffu;
    unsigned long long *sacc = sacc_all + tid;
    unsigned *scnt = scnt_all + tid;
    unsigned long long inertia = 0;
    int changed = 0;
    // ---- pieces decided as a whole: the piece's sums go in as they are (biased like the box-by-box sums: KM_MAGIC per box)
    if (valid) {
        const int L = cent_ok ? kc_decide<KMAX, FULL>(bb, cw, ch, k) : 255;
        if (L != 255) {
            atomicAdd(&sacc[(L * 2 + 0) * KM_BLOCK], (unsigned long long)ps0 + (unsigned long long)pc * KM_MAGIC);
            atomicAdd(&sacc[(L * 2 + 1) * KM_BLOCK], (unsigned long long)ps1 + (unsigned long long)pc * KM_MAGIC);
            atomicAdd(&scnt[L * KM_BLOCK], (unsigned)pc);
            if (prevl == 255u) {
                // its boxes were labelled one by one last time: how many of them change is counted by a whole wave below (the
                // queue entry says "count only"); this lane alone over 256 bytes was the longest thing in the launch whenever
                // the centroids still moved -- 24 us per iteration in a fit against 13 for the same launch at the fixed point
                const int at = atomicAdd(&s_qn, 1);
                s_queue[at] = tid; s_qstart[at] = pst; s_qcnt[at] = pc | (L << 16) | KC_Q_COUNT;
            } else if (prevl != (unsigned)L) {
                changed += pc;
            }
            plab[p] = (uint8_t)L;
        } else {
            const int at = atomicAdd(&s_qn, 1);                    // this thread's piece is walked below
            s_queue[at] = tid; s_qstart[at] = pst; s_qcnt[at] = pc | ((int)prevl << 16);
        }
    }
    __syncthreads();
    // ---- the other pieces box by box, one wave per piece, four boxes per lane: k_kmeans_step's exact assignment
    const int qn = s_qn;
    for (int qi = wave; qi < qn; qi += KM_BLOCK / 64) {
        const int owner = s_queue[qi];
        const int qp = owner * (int)gridDim.x + (int)blockIdx.x;
        const int qs = s_qstart[qi], qc = s_qcnt[qi] & 0xffff;
        const unsigned ql = ((unsigned)s_qcnt[qi] >> 16) & 0xffu;
        if (s_qcnt[qi] & KC_Q_COUNT) {                             // decided as a whole (label ql): only the changed count is open
#pragma unroll
            for (int j = 0; j < KC_PIECE / 64; ++j) {
                const int o = lane + 64 * j;
                if (o < qc) changed += (unsigned)slab[qs + o] != ql;
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < KC_PIECE / 64; ++j) {
            const int o = lane + 64 * j;
            if (o < qc) {
                const float2 b = swh[qs + o];
                const unsigned old = (ql == 255u) ? (unsigned)slab[qs + o] : ql;
                int best;
                float bd = 0.f;
                if (!(km_filter<KMAX, FULL>(b.x, b.y, cw, ch, k, gap_min, best) && (fabsf(b.x) + fabsf(b.y)) < KM_FAST_LIMIT))
                    best = km_assign<KMAX, FULL>(b.x, b.y, cw, ch, k, bd);
                if ((fabsf(b.x) + fabsf(b.y)) < KM_FAST_LIMIT) km_accum<true, false>(b.x, b.y, 0.f, best, sacc, scnt, inertia, changed, slab + qs + o, old);
                else km_accum<false, false>(b.x, b.y, 0.f, best, sacc, scnt, inertia, changed, slab + qs + o, old);
            }
        }
        if (lane == 0) plab[qp] = 255;
    }
    // ---- block reduction and the replica row: as k_kmeans_step
    __shared__ unsigned s_changed[KM_BLOCK];
    __shared__ unsigned long long s_cpart[COLS][16];
    s_changed[tid] = (unsigned)changed;
    __syncthreads();
    {
        const int sub = tid & 15;
        for (int col = tid >> 4; col < COLS; col += KM_BLOCK / 16) {
            unsigned long long t = 0;
            if (col < 2 * KMAX) {
                const unsigned long long *src = sacc_all + ((col % KMAX) * 2 + col / KMAX) * KM_BLOCK;
#pragma unroll
                for (int j = 0; j < 16; ++j) t += src[j * 16 + sub];
            } else if (col < 3 * KMAX) {
                const unsigned *src = scnt_all + (col - 2 * KMAX) * KM_BLOCK;
#pragma unroll
                for (int j = 0; j < 16; ++j) t += src[j * 16 + sub];
            } else if (col == 3 * KMAX) {
#pragma unroll
                for (int j = 0; j < 16; ++j) t += s_changed[j * 16 + sub];
            }
            s_cpart[col][sub] = t;
        }
    }
    __syncthreads();
    if (tid < COLS) {
        unsigned long long v = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) v += s_cpart[tid][j];
        const int grp = tid / KMAX, c = tid - grp * KMAX;
        if (grp < 2) {
            unsigned long long cnt = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) cnt += s_cpart[2 * KMAX + c][j];
            v -= cnt * KM_MAGIC;
        }
        const int col = (grp < 3) ? (c < k ? grp * k + c : -1) : 3 * k + c;
        if (col >= 0 && v != 0)
            atomicAdd(reinterpret_cast<unsigned long long *>(part_out) + (blockIdx.x % KM_REPL) * pstride + col, v);
    }
}

// ---- all cell iterations of a run in ONE launch ---------------------------------------------------------------------
// k_kc_iter's work is small (a decision per piece, a twentieth of the boxes walked): about 11 us per launch are launch,
// the piece loads and the fold of the previous sums.  Here the workgroups stay: a thread keeps its piece (bounding box, sums,
// range) in registers across iterations, and an iteration ends in a GRID BARRIER -- every workgroup adds its row into the
// iteration's partial-sums buffer with device-scope atomics, bumps a counter and waits until all have, then folds the
// buffer (loads that bypass the caches) and moves the centroids, exactly as the next launch's prologue would.  The grid is
// at most what the device holds at once (occupancy query), so every workgroup is resident and the barrier cannot wait for
// a workgroup that has not started; the wait is bounded all the same (about 0.2 s): a workgroup that gives up sets
// hdr->error and writes -1 into the history slots it leaves undone, which the host side turns into an error.
// Buffers rotate as in dg_kmeans_iou_run: iteration t adds into part[t % 3]; workgroup 0 zeroes part[(t + 1) % 3] BEFORE it
// arrives at barrier t (that buffer was last read after barrier t - 2, and reaching iteration t means every workgroup has
// passed barrier t - 1).
#define KC_SPIN_LIMIT (1u << 20)
template <int KMAX, bool FULL>
__global__ __launch_bounds__(KM_BLOCK) void k_kc_persist(KcHdr *__restrict__ hdr, const float2 *__restrict__ swh,
                                                         uint8_t *__restrict__ slab, uint8_t *__restrict__ plab,
                                                         const int *__restrict__ pstart, const int *__restrict__ pcnt,
                                                         const long long *__restrict__ psum, const float4 *__restrict__ pbox,
                                                         float *cent, int k, long long *part0, long long *part1, long long *part2,
                                                         long long *__restrict__ history, int64_t slot0, int iters, int pmax,
                                                         unsigned *__restrict__ bar)
{
    constexpr int COLS = 3 * KMAX + 2;
    __shared__ unsigned long long sacc_all[KMAX * 2 * KM_BLOCK];
    __shared__ unsigned scnt_all[KMAX * KM_BLOCK];
    __shared__ float2 scent[16];
    __shared__ long long s_prev[3 * 16 + 2];
    __shared__ int s_queue[KM_BLOCK], s_qstart[KM_BLOCK], s_qcnt[KM_BLOCK], s_qn, s_ok;
    __shared__ unsigned s_changed[KM_BLOCK];
    __shared__ unsigned long long s_cpart[COLS][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pstride = km_stride(k), cols = 3 * k + 2;
    const int G = (int)gridDim.x;
    long long *part[3] = {part0, part1, part2};
    // this thread's piece, for the whole run: slot tid of this workgroup (host: ceil(pmax / G) <= KM_BLOCK)
    const int p = tid * G + (int)blockIdx.x;
    int pst = 0, pc = 0;
    long long ps0 = 0, ps1 = 0;
    unsigned prevl = 255;
    float4 bb = make_float4(1.f, 0.f, 1.f, 0.f);
    if (p < pmax) { bb = pbox[p]; pst = pstart[p]; pc = pcnt[p]; ps0 = psum[2 * (size_t)p]; ps1 = psum[2 * (size_t)p + 1]; prevl = plab[p]; }
    const bool valid = pc > 0;
    if (tid < 16) scent[tid] = (tid < k) ? make_float2(cent[2 * tid], cent[2 * tid + 1]) : make_float2(0.f, 0.f);
    __syncthreads();
    for (int t = 0; t < iters; ++t) {
        float cw[KMAX], ch[KMAX];
        bool cent_ok = true;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) {
            cw[c] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(scent[c < 16 ? c : 0].x)));
            ch[c] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(scent[c < 16 ? c : 0].y)));
            if (FULL || c < k) cent_ok = cent_ok && (cw[c] > 0.0f) && (ch[c] > 0.0f) && (cw[c] * ch[c] < 0x1p60f);
        }
        const unsigned gap_min = cent_ok ? KM_GAP_MIN : 0xffffffffu;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) {
            sacc_all[(c * 2 + 0) * KM_BLOCK + tid] = 0;
            sacc_all[(c * 2 + 1) * KM_BLOCK + tid] = 0;
            scnt_all[c * KM_BLOCK + tid] = 0;
        }
        if (tid == 0) s_qn = 0;
        if (blockIdx.x == 0)
            for (int z = tid; z < KM_REPL * pstride; z += KM_BLOCK) part[(t + 1) % 3][z] = 0;
        __syncthreads();
        unsigned long long *sacc = sacc_all + tid;
        unsigned *scnt = scnt_all + tid;
        unsigned long long inertia = 0;
        int changed = 0;
        if (valid) {
            const int L = cent_ok ? kc_decide<KMAX, FULL>(bb, cw, ch, k) : 255;
            if (L != 255) {
                atomicAdd(&sacc[(L * 2 + 0) * KM_BLOCK], (unsigned long long)ps0 + (unsigned long long)pc * KM_MAGIC);
                atomicAdd(&sacc[(L * 2 + 1) * KM_BLOCK], (unsigned long long)ps1 + (unsigned long long)pc * KM_MAGIC);
                atomicAdd(&scnt[L * KM_BLOCK], (unsigned)pc);
                if (prevl == 255u) {                               // counted by a wave below, as in k_kc_iter
                    const int at = atomicAdd(&s_qn, 1);
                    s_queue[at] = tid; s_qstart[at] = pst; s_qcnt[at] = pc | (L << 16) | KC_Q_COUNT;
                } else if (prevl != (unsigned)L) {
                    changed += pc;
                }
                prevl = (unsigned)L;
            } else {
                const int at = atomicAdd(&s_qn, 1);
                s_queue[at] = tid; s_qstart[at] = pst; s_qcnt[at] = pc | ((int)prevl << 16);
                prevl = 255u;
            }
        }
        __syncthreads();
        const int qn = s_qn;
        for (int qi = wave; qi < qn; qi += KM_BLOCK / 64) {
            const int qs = s_qstart[qi], qc = s_qcnt[qi] & 0xffff;
            const unsigned ql = ((unsigned)s_qcnt[qi] >> 16) & 0xffu;
            if (s_qcnt[qi] & KC_Q_COUNT) {
#pragma unroll
                for (int j = 0; j < KC_PIECE / 64; ++j) {
                    const int o = lane + 64 * j;
                    if (o < qc) changed += (unsigned)slab[qs + o] != ql;
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < KC_PIECE / 64; ++j) {
                const int o = lane + 64 * j;
                if (o < qc) {
                    const float2 b = swh[qs + o];
                    const unsigned old = (ql == 255u) ? (unsigned)slab[qs + o] : ql;
                    int best;
                    float bd = 0.f;
                    if (!(km_filter<KMAX, FULL>(b.x, b.y, cw, ch, k, gap_min, best) && (fabsf(b.x) + fabsf(b.y)) < KM_FAST_LIMIT))
                        best = km_assign<KMAX, FULL>(b.x, b.y, cw, ch, k, bd);
                    if ((fabsf(b.x) + fabsf(b.y)) < KM_FAST_LIMIT) km_accum<true, false>(b.x, b.y, 0.f, best, sacc, scnt, inertia, changed, slab + qs + o, old);
                    else km_accum<false, false>(b.x, b.y, 0.f, best, sacc, scnt, inertia, changed, slab + qs + o, old);
                }
            }
        }
        s_changed[tid] = (unsigned)changed;
        __syncthreads();
        {
            const int sub = tid & 15;
            for (int col = tid >> 4; col < COLS; col += KM_BLOCK / 16) {
                unsigned long long v = 0;
                if (col < 2 * KMAX) {
                    const unsigned long long *src = sacc_all + ((col % KMAX) * 2 + col / KMAX) * KM_BLOCK;
#pragma unroll
                    for (int j = 0; j < 16; ++j) v += src[j * 16 + sub];
                } else if (col < 3 * KMAX) {
                    const unsigned *src = scnt_all + (col - 2 * KMAX) * KM_BLOCK;
#pragma unroll
                    for (int j = 0; j < 16; ++j) v += src[j * 16 + sub];
                } else if (col == 3 * KMAX) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) v += s_changed[j * 16 + sub];
                }
                s_cpart[col][sub] = v;
            }
        }
        __syncthreads();
        if (tid < COLS) {
            unsigned long long v = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) v += s_cpart[tid][j];
            const int grp = tid / KMAX, c = tid - grp * KMAX;
            if (grp < 2) {
                unsigned long long cnt = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) cnt += s_cpart[2 * KMAX + c][j];
                v -= cnt * KM_MAGIC;
            }
            const int col = (grp < 3) ? (c < k ? grp * k + c : -1) : 3 * k + c;
            if (col >= 0 && v != 0)
                atomicAdd(reinterpret_cast<unsigned long long *>(part[t % 3]) + (blockIdx.x % KM_REPL) * pstride + col, v);
        }
        // ---- grid barrier t: every workgroup's row is in part[t % 3] (and workgroup 0's zeroing of the next buffer is out)
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            atomicAdd(bar, 1u);
            const unsigned target = (unsigned)G * (unsigned)(t + 1);
            unsigned spins = 0;
            int ok = 1;
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > KC_SPIN_LIMIT) { ok = 0; break; }
            }
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) {                                             // uniform: give up, loudly
            if (tid == 0) {
                hdr->pad[0] = 1;
                if (blockIdx.x == 0 && history) for (int u = t; u < iters; ++u) history[slot0 + u] = -1;
            }
            break;
        }
        // ---- fold iteration t's sums (coherent loads: other workgroups' atomics went to memory), move the centroids
        if (tid < cols) {
            long long v = 0;
#pragma unroll
            for (int r = 0; r < KM_REPL; ++r) v += __hip_atomic_load(part[t % 3] + r * pstride + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_prev[tid] = v;
        }
        __syncthreads();
        if (tid < k) {
            float2 c = scent[tid];
            const long long cnt = s_prev[2 * k + tid];
            if (cnt > 0) { c.x = km_centroid(s_prev[tid], cnt); c.y = km_centroid(s_prev[k + tid], cnt); }
            scent[tid] = c;
            if (blockIdx.x == 0) { cent[2 * tid] = c.x; cent[2 * tid + 1] = c.y; }
        }
        if (blockIdx.x == 0 && tid == 0 && history) history[slot0 + t] = s_prev[3 * k];
        __syncthreads();
    }
    if (valid) plab[p] = (uint8_t)prevl;                         // the piece's label state for the next launch on this workspace
}

// history[slot] and acc[3k] <- the changed count of the cell pass (the box-order pass that produced acc compares with the
// caller's label array, which the cell iterations do not keep up to date)
__global__ void k_kc_changed(const long long *__restrict__ part, int k, long long *__restrict__ acc, long long *__restrict__ history,
                             int64_t slot)
{
    long long t = 0;
    for (int r = 0; r < KM_REPL; ++r) t += part[r * km_stride(k) + 3 * k];
    if (acc) acc[3 * k] = t;
    if (history) history[slot] = t;
}

extern "C" size_t dg_kmeans_prepared_workspace_bytes(int64_t n, int64_t k)
{
    if (n < 0 || k <= 0 || k > 16 || n >= (1ll << 31)) return 0;
    return kc_layout(n, k, nullptr).total;
}

extern "C" int dg_kmeans_prepare(const float *wh, int64_t n, int64_t k, const uint8_t *labels, void *ws, size_t ws_bytes, dg_stream_t s)
{
    DG_REQUIRE(n >= 1 && n < (1ll << 31) && k >= 1 && k <= 16, "kmeans_prepare: need 1 <= n < 2^31 and 1 <= k <= 16");
    DG_REQUIRE(wh && (((uintptr_t)wh) & 7) == 0, "kmeans_prepare: null or misaligned pointer");
    DG_REQUIRE(ws && (((uintptr_t)ws) & 255) == 0, "kmeans_prepare: workspace null or not 256-byte aligned");
    if (ws_bytes < dg_kmeans_prepared_workspace_bytes(n, k)) { dg_set_error("kmeans_prepare: workspace too small"); return DG_ENOMEM; }
    hipStream_t st = (hipStream_t)s;
    KcWs w = kc_layout(n, k, ws);
    const size_t lds = (size_t)KC_NC * 4;
    if (hipFuncSetAttribute((const void *)k_kc_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_kc_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_kc_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        dg_set_error("kmeans_prepare: cannot raise the LDS limit"); return DG_EHIP;
    }
    if (hipMemsetAsync(w.pcnt, 0, (size_t)w.pmax * 4, st) != hipSuccess) { dg_set_error("kmeans_prepare: memset failed"); return DG_EHIP; }
    hipLaunchKernelGGL(k_kc_hist, dim3(KC_G), dim3(1024), lds, st, (const float2 *)wh, (long long)n, w.counts);
    hipLaunchKernelGGL(k_kc_colscan, dim3((KC_NC + 255) / 256), dim3(256), 0, st, w.counts, w.tot);
    hipLaunchKernelGGL(k_kc_scan, dim3(1), dim3(1024), lds, st, w.tot, w.cell_start, w.piece_first, w.hdr, (const void *)wh, (long long)n, (int)k);
    hipLaunchKernelGGL(k_kc_pieces, dim3((KC_NC + 255) / 256), dim3(256), 0, st, w.tot, w.cell_start, w.piece_first, w.pcell, w.pstart, w.pcnt, w.plab,
                       labels ? KC_PER_BOX : KC_UNSET);
    hipLaunchKernelGGL(k_kc_scatter, dim3(KC_G), dim3(1024), lds, st, (const float2 *)wh, labels, (long long)n, w.counts, w.cell_start, w.swh, w.slab);
    hipLaunchKernelGGL(k_kc_piecesums, dim3((unsigned)((w.pmax + 15) / 16)), dim3(256), 0, st, w.swh, w.pstart, w.pcnt, w.hdr, w.psum, w.pbox);
    DG_CHECK_LAUNCH("kmeans_prepare");
    return DG_OK;
}

static int kc_launch(const KcWs &w, float *cent, int64_t k, const long long *prev, long long *out, long long *zero, int64_t *history,
                     int64_t slot, int write_cent, hipStream_t st)
{
    const unsigned grid = (unsigned)((w.pmax + KC_WG_PIECES - 1) / KC_WG_PIECES);
#define KC_LAUNCH(KM, FULL)                                                                                                       \
    hipLaunchKernelGGL((k_kc_iter<KM, FULL>), dim3(grid), dim3(KM_BLOCK), 0, st, w.hdr, w.swh, w.slab, w.plab, w.pcell, w.pstart, \
                       w.pcnt, w.psum, w.pbox, cent, (int)k, prev, out, zero, (long long *)history, slot, write_cent, (int)w.pmax)
    const int kmax = km_kmax(k);
    if (k == 5) KC_LAUNCH(5, true);
    else if (kmax == 5) KC_LAUNCH(5, false);
    else if (k == 9) KC_LAUNCH(9, true);
    else if (kmax == 9) KC_LAUNCH(9, false);
    else KC_LAUNCH(16, false);
#undef KC_LAUNCH
    DG_CHECK_LAUNCH("k_kc_iter");
    return DG_OK;
}

// dg_kmeans_iou_run on a workspace dg_kmeans_prepare filled for these boxes: same results (cent, labels, acc, history),
// the iterations before the last one over cells.
extern "C" int dg_kmeans_iou_run_prepared(const float *wh, int64_t n, float *cent, int64_t k, uint8_t *labels, int64_t *acc,
                                          int64_t *history, int64_t slot0, int64_t iters, void *ws, size_t ws_bytes, dg_stream_t s)
{
    DG_REQUIRE(iters >= 0, "kmeans_run_prepared: iters < 0");
    // labels == NULL: the cell iterations only -- centroids and history as always, but no pass over the boxes in their own
    // order at the end, hence no labels, no inertia and no acc (which may be NULL too).  For the chunks of a fit that runs
    // until nothing changes: only its last chunk's results are ever looked at, and once an iteration has changed nothing
    // one dg_kmeans_iou_step with the centroids as they stand gives exactly those results.
    const bool light = labels == nullptr;
    static const uint8_t no_labels = 0;
    int rc = km_check("kmeans_run_prepared", wh, n, cent, k, light ? &no_labels : labels);
    if (rc) return rc;
    DG_REQUIRE((acc || light) && n >= 1, "kmeans_run_prepared: null acc or no boxes");
    DG_REQUIRE(ws && (((uintptr_t)ws) & 255) == 0, "kmeans_run_prepared: workspace null or not 256-byte aligned");
    if (ws_bytes < dg_kmeans_prepared_workspace_bytes(n, k)) { dg_set_error("kmeans_run_prepared: workspace too small"); return DG_ENOMEM; }
    if (iters == 0) return DG_OK;
    hipStream_t st = (hipStream_t)s;
    KcWs w = kc_layout(n, k, ws);
    const size_t pe = dg_kmeans_part_elems(k);
    // four buffers: three in rotation for the cell iterations (as dg_kmeans_iou_run), the fourth for the box-order pass
    km_zero(w.part[0], (size_t)((char *)w.part[3] - (char *)w.part[0]) + pe * 8, st);
    // default: one launch per cell iteration (k_kc_iter); DG_KMEANS_PERSIST=1: all of them in one launch (k_kc_persist, a grid
    // barrier per iteration -- measured slower on this part, see DESIGN.md: kept for the measurement and as the structure a
    // peer-mapped exchange between GPUs would need)
    const bool persist_on = getenv("DG_KMEANS_PERSIST") && atoi(getenv("DG_KMEANS_PERSIST")) != 0;
    int64_t t0 = 0;
    if (persist_on && iters >= 2) {
        const int kmax = km_kmax(k);
        int resident;
#define KC_RES(KM, FULL) resident = km_resident_blocks(k_kc_persist<KM, FULL>)
        if (k == 5) KC_RES(5, true); else if (kmax == 5) KC_RES(5, false); else if (k == 9) KC_RES(9, true); else if (kmax == 9) KC_RES(9, false); else KC_RES(16, false);
#undef KC_RES
        int64_t G = (w.pmax + KC_WG_PIECES - 1) / KC_WG_PIECES;
        if (G > resident) G = resident;
        if (getenv("DG_KMEANS_PERSIST_G") && atoi(getenv("DG_KMEANS_PERSIST_G")) > 0 && atoi(getenv("DG_KMEANS_PERSIST_G")) < G) G = atoi(getenv("DG_KMEANS_PERSIST_G"));
        if ((w.pmax + G - 1) / G <= KM_BLOCK) {                       // a piece per thread at most
            unsigned *bar = (unsigned *)&w.hdr->pad[1];
            km_zero(&w.hdr->pad[0], 16, st);
#define KC_PERSIST(KM, FULL)                                                                                                       \
            hipLaunchKernelGGL((k_kc_persist<KM, FULL>), dim3((unsigned)G), dim3(KM_BLOCK), 0, st, w.hdr, w.swh, w.slab, w.plab, w.pstart, \
                               w.pcnt, w.psum, w.pbox, cent, (int)k, w.part[0], w.part[1], w.part[2], (long long *)history, slot0,       \
                               (int)(iters - 1), (int)w.pmax, bar)
            if (k == 5) KC_PERSIST(5, true); else if (kmax == 5) KC_PERSIST(5, false); else if (k == 9) KC_PERSIST(9, true);
            else if (kmax == 9) KC_PERSIST(9, false); else KC_PERSIST(16, false);
#undef KC_PERSIST
            DG_CHECK_LAUNCH("k_kc_persist");
            t0 = iters - 1;                                         // the centroids of the last iteration are in `cent` now
        }
    }
    for (int64_t t = t0; t < iters; ++t) {
        const long long *prev = (t && !t0) ? w.part[(t - 1) % 3] : nullptr;
        if (t + 1 == iters && !light) {
            // the last iteration twice from the same centroids: box order for labels / inertia / acc, cells for `changed`
            if ((rc = km_launch(wh, n, cent, k, labels, (const int64_t *)prev, (int64_t *)w.part[3], nullptr, history, slot0 + t - 1, true, st))) return rc;
            if ((rc = kc_launch(w, cent, k, prev, w.part[t % 3], nullptr, nullptr, 0, 0, st))) return rc;
        } else {
            if ((rc = kc_launch(w, cent, k, prev, w.part[t % 3], t + 1 < iters ? w.part[(t + 1) % 3] : nullptr, history, slot0 + t - 1, 1, st))) return rc;
        }
    }
    if (light)                                                    // fold the last iteration: centroids, its changed count
        return km_finish((const int64_t *)w.part[(iters - 1) % 3], k, cent, nullptr, history, slot0 + iters - 1, st);
    if ((rc = km_finish((const int64_t *)w.part[3], k, cent, acc, nullptr, 0, st))) return rc;
    hipLaunchKernelGGL(k_kc_changed, dim3(1), dim3(1), 0, st, w.part[(iters - 1) % 3], (int)k, (long long *)acc, (long long *)history, slot0 + iters - 1);
    DG_CHECK_LAUNCH("k_kc_changed");
    return DG_OK;
}
