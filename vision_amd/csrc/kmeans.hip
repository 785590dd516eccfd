// kmeans.hip — YOLOv2-style anchor k-means with d = 1 - IoU (gfx950).
// The reference only fixes the sklearn-like API (clustering/cluster_prac.py:38-40); the Lloyd step
// is build-defined and specified in oracle/oracle.c (orc_kmeans_step / orc_kmeans_update):
//   d = 1 - inter/((w*h + cw*ch) - inter), fp32, ties -> lowest index; sums in int64 fixed point
//   (llrint(x * 2^32)) so any sharding / reduction order gives identical bits.
// Step kernel: grid-stride, two boxes per lane per 16-byte load, centroids in SGPRs, per-lane
// partial sums in LDS -> wave shuffle reduce -> one partial row per workgroup (plain stores); the workgroup
// that finishes last folds the rows into acc[3k+2] and, in the device-side loop, updates the centroids.
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
// rarely in doubt: a = inter * rcp(uni) is within 1.5 * 2^-23 (relative) of the exact quotient q.  The cluster index
// replaces the low four mantissa bits (another 15 ulp), and an unsigned max / median-of-three pair keeps the two
// largest keys m1 >= m2.  When m1 - m2 > 256 ulp (>= 2^-16 relative) and m1 >= 2^-5, the exact quotients differ by
// more than q * 2^-17 >= 2^-22, i.e. by several ulps of d = fl(1 - q): that cluster wins the exact comparison
// strictly, whatever the index order, and only its d is evaluated exactly.  Anything else (near ties, tiny overlaps,
// NaN/inf/negative keys, which the unsigned max drags into m1) takes km_assign.  Callers bound |w|+|h| < 2^19 and
// centroid areas < 2^60 so that rcp(uni) stays a normal number.
#define KM_GAP_MIN 256u
#define KM_KEY_LO 0x3D000000u           /* 2^-5 */
#define KM_KEY_HI 0x7F800000u           /* +inf: NaN and sign-bit keys compare above it */
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int KMAX, bool FULL>
__device__ __forceinline__ bool km_filter(float w, float h, const float (&cw)[KMAX], const float (&ch)[KMAX], int k, unsigned gap_min,
                                          int &best)
{
    const float area = w * h;
    unsigned m1 = 0, m2 = 0;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        if (FULL || c < k) {
            const float inter = fminf(w, cw[c]) * fminf(h, ch[c]);
            const float uni = (area + (cw[c] * ch[c])) - inter;
            const unsigned key = (__float_as_uint(inter * __builtin_amdgcn_rcpf(uni)) & ~15u) | (unsigned)c;
            m2 = umed3(m1, m2, key);
            m1 = max(m1, key);
        }
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
                                         unsigned long long &inertia, int &changed, uint8_t *lab)
{
    atomicAdd(&sacc[(best * 2 + 0) * KM_BLOCK], fx32_biased<FAST>(w));
    atomicAdd(&sacc[(best * 2 + 1) * KM_BLOCK], fx32_biased<FAST>(h));
    atomicAdd(&scnt[best * KM_BLOCK], 1u);
    if (INERTIA) inertia += fx32_biased<FAST>(bd);
    changed += (*lab != (uint8_t)best);
    *lab = (uint8_t)best;
}

template <int KMAX, bool FULL, bool INERTIA = true>
__device__ __forceinline__ void km_one(float w, float h, const float (&cw)[KMAX], const float (&ch)[KMAX], int k,
                                       unsigned long long *sacc, unsigned *scnt, unsigned long long &inertia, int &changed,
                                       uint8_t *lab)
{
    float bd;
    const int best = km_assign<KMAX, FULL>(w, h, cw, ch, k, bd);
    if ((fabsf(w) + fabsf(h)) + fabsf(bd) < KM_FAST_LIMIT) km_accum<true, INERTIA>(w, h, bd, best, sacc, scnt, inertia, changed, lab);
    else km_accum<false, INERTIA>(w, h, bd, best, sacc, scnt, inertia, changed, lab);
}

// INERTIA = false (the iterations of dg_kmeans_iou_run before its last): the sum of distances is not formed, so the
// winner's exact distance is not evaluated either; acc[3k+1] is then written as 0.  A Lloyd iteration needs the sums,
// the counts and the changed count; the inertia is reported for the final assignment only (as sklearn does).
// mode: KM_MODE_STEP  plain step: acc <- this shard's sums, centroids untouched;
//       KM_MODE_RUN   single-device loop: the last workgroup also moves the centroids by this launch's sums and
//                     writes history[slot] = this iteration's changed count;
//       KM_MODE_APPLY sharded loop: `acc` arrives holding the ALL-REDUCED sums of the previous iteration; every
//                     workgroup first derives this iteration's centroids from (cent, acc) -- the centroid update is the
//                     prologue of the next step, so an iteration is ONE launch + ONE all-reduce -- and the last
//                     workgroup writes those centroids back to `cent`, history[slot] = the previous iteration's changed
//                     count, and acc <- this shard's sums.  In-place is safe: a workgroup draws its ticket after its
//                     last use of cent / acc, and only the drawer of the last ticket writes them.
#define KM_MODE_STEP 0
#define KM_MODE_RUN 1
#define KM_MODE_APPLY 2
__device__ __forceinline__ float km_centroid(long long sum, long long cnt)      // as k_kmeans_update / orc_kmeans_update
{
    return (float)((double)sum / (double)cnt / 4294967296.0);
}

template <int KMAX, bool FULL, bool INERTIA>
__global__ __launch_bounds__(KM_BLOCK) void k_kmeans_step(const float2 *__restrict__ wh, int64_t n, float *cent, int k,
                                                          uint8_t *__restrict__ labels, long long *partials, unsigned *ticket,
                                                          long long *acc, int mode, long long *__restrict__ history,
                                                          int64_t slot)
{
    constexpr int COLS = 3 * KMAX + 2, CP = (COLS + 1) & ~1;      // a row is padded to whole 16-byte pairs
    __shared__ unsigned long long sacc_all[KMAX * 2 * KM_BLOCK];
    __shared__ unsigned scnt_all[KMAX * KM_BLOCK];
    __shared__ long long red[KM_BLOCK / 64][3 * KMAX + 2];
    __shared__ float2 scent[16];
    __shared__ long long s_prev_changed;
    const int tid = threadIdx.x;
    if (tid < 16) {
        float2 c = (tid < k) ? make_float2(cent[2 * tid], cent[2 * tid + 1]) : make_float2(0.f, 0.f);
        if (mode == KM_MODE_APPLY && tid < k) {
            const long long cnt = acc[2 * k + tid];
            if (cnt > 0) { c.x = km_centroid(acc[tid], cnt); c.y = km_centroid(acc[k + tid], cnt); }
        }
        scent[tid] = c;
        if (tid == 0) s_prev_changed = (mode == KM_MODE_APPLY) ? acc[3 * k] : 0;
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
        sacc_all[(c * 2 + 0) * KM_BLOCK + tid] = 0;
        sacc_all[(c * 2 + 1) * KM_BLOCK + tid] = 0;
        scnt_all[c * KM_BLOCK + tid] = 0;
    }
    __syncthreads();
    const unsigned gap_min = cent_ok ? KM_GAP_MIN : 0xffffffffu;     // a gap nobody reaches: exact path only
    unsigned long long *sacc = sacc_all + tid;     // lane-private column: no barrier needed until the final reduce
    unsigned *scnt = scnt_all + tid;
    unsigned long long inertia = 0;                // biased like the LDS sums: KM_MAGIC per box
    int changed = 0;

    // two boxes per lane per 16-byte load, two loads in flight; an 8-byte-aligned shard start is peeled off first
    const int head = ((reinterpret_cast<uintptr_t>(wh) & 8) && n > 0) ? 1 : 0;
    if (head && blockIdx.x == 0 && tid == 0) {
        const float2 t = wh[0];
        km_one<KMAX, FULL, INERTIA>(t.x, t.y, cw, ch, k, sacc, scnt, inertia, changed, labels);
    }
    wh += head; n -= head;
    const int64_t npair = n >> 1;
    const float4 *wh4 = reinterpret_cast<const float4 *>(wh);
    uint8_t *lab = labels + head;
    const int64_t stride = (int64_t)gridDim.x * KM_BLOCK;
    int64_t i = (int64_t)blockIdx.x * KM_BLOCK + tid;
    bool have = i + stride < npair;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (have) { a = wh4[i]; b = wh4[i + stride]; }
    while (have) {
        // the next pair of loads is issued before this pair is worked on
        const int64_t inext = i + 2 * stride;
        const bool more = inext + stride < npair;
        float4 na = a, nb = b;
        if (more) { na = wh4[inext]; nb = wh4[inext + stride]; }
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
            km_accum<true, INERTIA>(a.x, a.y, d0, b0, sacc, scnt, inertia, changed, lab + 2 * i);
            km_accum<true, INERTIA>(a.z, a.w, d1, b1, sacc, scnt, inertia, changed, lab + 2 * i + 1);
            km_accum<true, INERTIA>(b.x, b.y, d2, b2, sacc, scnt, inertia, changed, lab + 2 * (i + stride));
            km_accum<true, INERTIA>(b.z, b.w, d3, b3, sacc, scnt, inertia, changed, lab + 2 * (i + stride) + 1);
        } else {
            km_accum<false, INERTIA>(a.x, a.y, d0, b0, sacc, scnt, inertia, changed, lab + 2 * i);
            km_accum<false, INERTIA>(a.z, a.w, d1, b1, sacc, scnt, inertia, changed, lab + 2 * i + 1);
            km_accum<false, INERTIA>(b.x, b.y, d2, b2, sacc, scnt, inertia, changed, lab + 2 * (i + stride));
            km_accum<false, INERTIA>(b.z, b.w, d3, b3, sacc, scnt, inertia, changed, lab + 2 * (i + stride) + 1);
        }
        i = inext; a = na; b = nb; have = more;
    }
    for (; i < npair; i += stride) {                     // what the four-box loop leaves: same scheme, two boxes
        const float4 t = wh4[i];
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
            km_accum<true, INERTIA>(t.x, t.y, d0, b0, sacc, scnt, inertia, changed, lab + 2 * i);
            km_accum<true, INERTIA>(t.z, t.w, d1, b1, sacc, scnt, inertia, changed, lab + 2 * i + 1);
        } else {
            km_accum<false, INERTIA>(t.x, t.y, d0, b0, sacc, scnt, inertia, changed, lab + 2 * i);
            km_accum<false, INERTIA>(t.z, t.w, d1, b1, sacc, scnt, inertia, changed, lab + 2 * i + 1);
        }
    }
    if ((n & 1) && blockIdx.x == 0 && tid == 0) {
        const float2 t = wh[n - 1];
        km_one<KMAX, FULL, INERTIA>(t.x, t.y, cw, ch, k, sacc, scnt, inertia, changed, lab + (n - 1));
    }

    // block reduction: wave shuffles, then LDS across the 4 waves
    const int lane = tid & 63, wave = tid >> 6;
    long long nbox = 0;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
        long long a = wave_sum_ll((long long)sacc[(c * 2 + 0) * KM_BLOCK]), b2 = wave_sum_ll((long long)sacc[(c * 2 + 1) * KM_BLOCK]);
        long long c2 = wave_sum_ll((long long)scnt[c * KM_BLOCK]);
        const long long bias = (long long)((unsigned long long)c2 * KM_MAGIC);   // mod 2^64, like the sums
        nbox += c2;
        if (lane == 0) { red[wave][c] = a - bias; red[wave][KMAX + c] = b2 - bias; red[wave][2 * KMAX + c] = c2; }
    }
    {
        long long a = wave_sum_ll((long long)changed), b2 = wave_sum_ll((long long)inertia);
        if (lane == 0) { red[wave][3 * KMAX] = a; red[wave][3 * KMAX + 1] = INERTIA ? b2 - (long long)((unsigned long long)nbox * KM_MAGIC) : 0; }
    }
    __syncthreads();
    if (tid < CP) {
        long long v = 0;
        if (tid < COLS) {
#pragma unroll
            for (int w = 0; w < KM_BLOCK / 64; ++w) v += red[w][tid];
        }
        // Device-scope (write-through) store, completed before the ticket is drawn.  A __threadfence() here costs a
        // sweep of the XCD's whole L2 per workgroup (+24 us per launch when every workgroup does one).
        __hip_atomic_store(&partials[(int64_t)blockIdx.x * CP + tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // The workgroup that draws the last ticket folds all rows (and, inside dg_kmeans_iou_run, moves the centroids):
    // one launch per Lloyd iteration instead of three.  The ticket returns to zero for the next launch.
    __shared__ unsigned s_ticket;
    __syncthreads();
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != gridDim.x - 1) return;
    unsigned long long *s_tot = sacc_all;                  // the accumulators are spent; reuse their LDS
    if (tid < CP) s_tot[tid] = 0;
    __syncthreads();
    {
        // Device-scope loads (sc1): the rows were written from every XCD and must not come from this XCD's L2.
        // Thirty-two are issued before the first is used; as atomic loads the compiler would wait for each in turn.
        // A thread strides by a multiple of the row length, so all its values belong to one column and meet in a register.
        constexpr int ACTIVE = (KM_BLOCK / CP) * CP;
        const int nvals = (int)gridDim.x * CP;
        if (tid < ACTIVE) {
            long long sum = 0;
            for (int base = tid; base < nvals; base += 32 * ACTIVE) {
                long long v[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const int p = base + j * ACTIVE;
                    const long long *ptr = partials + (p < nvals ? p : 0);
                    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(v[j]) : "v"(ptr) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                               "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                             :
                             : "memory");
                asm volatile(""
                             : "+v"(v[16]), "+v"(v[17]), "+v"(v[18]), "+v"(v[19]), "+v"(v[20]), "+v"(v[21]), "+v"(v[22]), "+v"(v[23]),
                               "+v"(v[24]), "+v"(v[25]), "+v"(v[26]), "+v"(v[27]), "+v"(v[28]), "+v"(v[29]), "+v"(v[30]), "+v"(v[31])
                             :
                             : "memory");
#pragma unroll
                for (int j = 0; j < 32; ++j) sum += (base + j * ACTIVE < nvals) ? v[j] : 0;
            }
            atomicAdd(&s_tot[tid % CP], (unsigned long long)sum);
        }
    }
    __syncthreads();
    if (tid < COLS) {
        const long long t = (long long)s_tot[tid];
        const int grp = tid / KMAX, c = tid - grp * KMAX;
        if (grp < 3) { if (c < k) acc[grp * k + c] = t; }
        else acc[3 * k + c] = t;                           // changed, inertia
    }
    if (mode == KM_MODE_RUN) {                             // same arithmetic as k_kmeans_update
        if (tid < k) {
            const long long cnt = (long long)s_tot[2 * KMAX + tid];
            if (cnt > 0) {
                cent[2 * tid] = km_centroid((long long)s_tot[tid], cnt);
                cent[2 * tid + 1] = km_centroid((long long)s_tot[KMAX + tid], cnt);
            }
        }
        if (tid == 0 && history) history[slot] = (long long)s_tot[3 * KMAX];
    } else if (mode == KM_MODE_APPLY) {                    // the centroids this launch derived and used
        if (tid < k) { cent[2 * tid] = scent[tid].x; cent[2 * tid + 1] = scent[tid].y; }
        if (tid == 0 && history) history[slot] = s_prev_changed;
    }
    if (tid == 0) *ticket = 0;
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
    // at least 16 boxes per thread: a small shard (1.25 M boxes) runs 16 us with 305 workgroups and 26 us with 1024,
    // whose prologues and partial rows (folded by the last workgroup) then outweigh the boxes
    int64_t b = dg_cdiv(n, KM_BLOCK * 16);
    return (int)(b < 1 ? 1 : (b > resident ? resident : b));
}

extern "C" size_t dg_kmeans_workspace_bytes(int64_t n, int64_t k)
{
    if (n < 0 || k <= 0 || k > 16) return 0;
    return (size_t)KM_MAXBLOCKS * (3 * 16 + 2) * 8 + 16;     // rows + the ticket
}
static inline unsigned *km_ticket(void *ws) { return (unsigned *)((char *)ws + (size_t)KM_MAXBLOCKS * (3 * 16 + 2) * 8); }

// step (+ fold) as one launch; with cent_out the last workgroup also writes the new centroids and history[slot]
static int km_launch(const float *wh, int64_t n, float *cent, int64_t k, uint8_t *labels, int64_t *acc, int mode,
                     int64_t *history, int64_t slot, bool inertia, void *ws, hipStream_t st)
{
    const int kmax = km_kmax(k);
    long long *partials = (long long *)ws;
#define KM_LAUNCH1(KM, FULL, INERTIA)                                                                                           \
    do {                                                                                                                        \
        static const int resident = km_resident_blocks(k_kmeans_step<KM, FULL, INERTIA>);                                       \
        hipLaunchKernelGGL((k_kmeans_step<KM, FULL, INERTIA>), dim3(km_blocks(n, resident)), dim3(KM_BLOCK), 0, st,             \
                           (const float2 *)wh, n, cent, (int)k, labels, partials, km_ticket(ws), (long long *)acc, mode,        \
                           (long long *)history, slot);                                                                         \
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

static int km_check(const float *wh, int64_t n, const float *cent, int64_t k, const uint8_t *labels, const int64_t *acc, const void *ws,
                    size_t ws_bytes)
{
    DG_REQUIRE(n >= 0 && k >= 1 && k <= 16, "kmeans_step: need n >= 0 and 1 <= k <= 16");
    DG_REQUIRE(cent && acc && ws, "kmeans_step: null pointer");
    DG_REQUIRE(n == 0 || (wh && labels), "kmeans_step: null pointer");
    DG_REQUIRE((((uintptr_t)wh) & 7) == 0 && (((uintptr_t)ws) & 15) == 0, "kmeans_step: misaligned pointer");
    if (ws_bytes < dg_kmeans_workspace_bytes(n, k)) { dg_set_error("kmeans_step: workspace too small"); return DG_ENOMEM; }
    return DG_OK;
}

extern "C" int dg_kmeans_iou_step(const float *wh, int64_t n, const float *cent, int64_t k, uint8_t *labels, int64_t *acc,
                                  void *ws, size_t ws_bytes, dg_stream_t s)
{
    int rc = km_check(wh, n, cent, k, labels, acc, ws, ws_bytes);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)s;
    // the workspace arrives with arbitrary contents
    if (hipMemsetAsync(km_ticket(ws), 0, 16, st) != hipSuccess) { dg_set_error("kmeans_step: memset failed"); return DG_EHIP; }
    return km_launch(wh, n, const_cast<float *>(cent), k, labels, acc, KM_MODE_STEP, nullptr, 0, true, ws, st);
}

// One iteration of the SHARDED Lloyd loop as one launch (KM_MODE_APPLY above).  apply_prev != 0: acc holds the
// all-reduced sums of the previous iteration; the centroids are first moved by them (empty cluster keeps its centroid),
// written back to `cent`, and history[slot] (may be NULL) receives the previous iteration's changed count.  Then the
// step runs and acc is overwritten with this shard's sums: all-reduce it and call again.  After the last iteration,
// dg_kmeans_update applies the last sums.  want_inertia == 0 leaves acc[3k+1] = 0 (a Lloyd iteration does not use it).
// A call with apply_prev != 0 must follow another dg_kmeans_iou_iter call on the same workspace, stream-ordered, with
// nothing else using that workspace in between (the launch ticket in it is cleared by apply_prev == 0 calls only).
extern "C" int dg_kmeans_iou_iter(const float *wh, int64_t n, float *cent, int64_t k, uint8_t *labels, int64_t *acc,
                                  int apply_prev, int want_inertia, int64_t *history, int64_t slot, void *ws,
                                  size_t ws_bytes, dg_stream_t s)
{
    int rc = km_check(wh, n, cent, k, labels, acc, ws, ws_bytes);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)s;
    // the ticket returns to zero at the end of every launch: only the first call of a sequence has to clear it
    if (!apply_prev && hipMemsetAsync(km_ticket(ws), 0, 16, st) != hipSuccess) { dg_set_error("kmeans_iter: memset failed"); return DG_EHIP; }
    return km_launch(wh, n, cent, k, labels, acc, apply_prev ? KM_MODE_APPLY : KM_MODE_STEP, apply_prev ? history : nullptr,
                     slot, want_inertia != 0, ws, st);
}

extern "C" int dg_kmeans_update(const int64_t *acc, int64_t k, float *cent, int64_t *history, int64_t slot, dg_stream_t s)
{
    DG_REQUIRE(k >= 1 && k <= 16 && acc && cent, "kmeans_update: bad argument");
    hipLaunchKernelGGL(k_kmeans_update, dim3(1), dim3(64), 0, (hipStream_t)s, (const long long *)acc, (int)k, cent,
                       (long long *)history, slot);
    DG_CHECK_LAUNCH("k_kmeans_update");
    return DG_OK;
}

// `iters` Lloyd iterations back to back on one device (one launch per iteration, no host round trip);
// history[slot0 + t] = changed count of iteration t.  A converged iteration reproduces the same centroids, so
// running past convergence is idempotent and n_iter_ is read from the history afterwards.
extern "C" int dg_kmeans_iou_run(const float *wh, int64_t n, float *cent, int64_t k, uint8_t *labels, int64_t *acc,
                                 int64_t *history, int64_t slot0, int64_t iters, void *ws, size_t ws_bytes, dg_stream_t s)
{
    DG_REQUIRE(iters >= 0, "kmeans_run: iters < 0");
    int rc = km_check(wh, n, cent, k, labels, acc, ws, ws_bytes);
    if (rc) return rc;
    if (iters == 0) return DG_OK;
    hipStream_t st = (hipStream_t)s;
    if (hipMemsetAsync(km_ticket(ws), 0, 16, st) != hipSuccess) { dg_set_error("kmeans_run: memset failed"); return DG_EHIP; }
    for (int64_t t = 0; t < iters; ++t) {
        rc = km_launch(wh, n, cent, k, labels, acc, KM_MODE_RUN, history, slot0 + t, t + 1 == iters, ws, st);   // inertia: last only
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
