// nms.hip — greedy bitmask NMS for gfx950 (wave64).
// Reference: nms(out_box,out_score,threshold), yolo/yolo_v1/yolo.py:188-203, whose IoU row is
// get_iou (yolo/yolo_v1/modules.py:73-98).  Three phases, all batched over images:
//   sort   : 64-bit keys (~orderable(score) << 32 | index), bitonic in LDS (global passes for
//            segments > 4096), then a gather that lays the sorted boxes out as SoA planes
//            x1[],y1[],x2[],y2[],area[] so the mask phase reads coalesced dwords;
//   mask   : for sorted rows i and columns j>i, bit (i,j) = !(iou(i,j) < thr).  One lane owns one
//            column, the row box is an LDS broadcast and the 64-bit word of a row is the wave's
//            v_cmp result (__ballot); words go out row-major [i][W] in 128-byte pieces;
//   reduce : one workgroup per image walks the 64-row blocks in order; the intra-block chain is
//            resolved from the diagonal word in scalar registers (readlane), the kept rows are
//            OR-ed into per-thread `removed` words.  Implements the reference's `numel()>1`
//            quirk exactly.
#include "common.hpp"
#include <type_traits>
#include <string.h>
#include <math.h>

#define SORT_CHUNK 4096  // elements sorted inside LDS (32 KiB of keys)
#define NMS_CSR_MIN_W 8  // from 8 words (449 boxes) per image on, the reduce runs from the compact mask (k_nms_csr_*)

// per image, 64 bytes: what the pruned path (xbucket_image ...) decided; the dense-mask kernels of an image run iff mode == 1
struct NmsHdr {
    float xmin, scale;       // x-bucket of a coordinate: clamp((x - xmin) * scale, 0, XB_G - 1)
    int32_t mode;            // 0: pruned path (banded pair search in x-bucket order); 1: dense mask
    int32_t fpdone;          // k_nms_resolve_mc converged and wrote the keep list
    int32_t rounds;          // rounds it took (diagnostic)
    int32_t npairs;          // suppressing pairs k_nms_band found (all 64 lists; written by k_nms_resolve_mc)
    int32_t cpb;             // candidate pairs per box, as xbucket_image estimated them
    uint32_t estsum;         // k_nms_estimate: suppressing pairs of 1024 sampled boxes (x n / 1024 = the image's)
    int32_t pad[8];
};

// the 64 list cursors of an image: 128 bytes apart -- device-scope atomics are carried out at the memory side, line by line,
// and 64 hot counters in two lines made k_nms_band 37 us slower than one cursor had
#define LC_STRIDE 32
struct NmsWs {
    int64_t Np, W, SW, Np2;
    u64 *keys;
    float *x1, *y1, *x2, *y2, *area;
    int32_t *order;
    u64 *mask, *rsum;
    // compact form of the mask for 64 < W <= 1024 (see k_nms_csr_*): per 64-row block a contiguous run of
    // (row-in-block << 16 | word, bits) entries for the non-zero words right of the diagonal
    int64_t ecap;            // entries per image the workspace holds
    int32_t *bcnt, *boff, *cursor, *cflag;
    u32 *ekey;
    u64 *eval;
    // pruned path (round 3, 64 < W <= 1024): header per image, the boxes in x-bucket order (bbox = corners, bar = {area, rank}),
    // each box's position in that order, bucket starts, "has a suppressor" bits, diagonal words, two work lists
    int pruned;              // the layout holds the arrays below
    struct NmsHdr *hdr;
    float4 *bbox;
    float2 *bar;
    int32_t *posu, *start;   // position in x-bucket order by the caller's index (found beside the chunk sort), bucket starts
    u64 *diag;
    int64_t lcap;            // entries each of an image's 64 pair lists (and 64 work lists) holds: they share the image's dense mask
    int lsh;                 // a pair's list: higher rank >> lsh
    int32_t *lcnt;           // [batch][64][LC_STRIDE] entries in each list, every counter in a 128-byte line of its own
    u32 *mc;                 // [batch][MC_WORDS] what the workgroups of k_nms_resolve_mc tell each other
    size_t total;
};

// boxes of image `img`: the caller's count, held to [0, nmax] (a count from outside must not become a read outside the image's
// boxes), or nmax when there are no counts
__device__ __forceinline__ int nms_count(const int32_t *__restrict__ counts, int img, int64_t nmax)
{
    if (!counts) return (int)nmax;
    const int c = counts[img];
    return c < 0 ? 0 : (c > (int)nmax ? (int)nmax : c);
}

static inline int64_t next_pow2(int64_t v) { int64_t p = 64; while (p < v) p <<= 1; return p; }

static NmsWs nms_layout(int64_t batch, int64_t nmax, void *base)
{
    NmsWs w;
    w.Np = dg_round_up(nmax > 0 ? nmax : 1, 64);
    w.W = w.Np / 64;
    w.SW = dg_cdiv(w.W, 64);
    w.Np2 = next_pow2(nmax);
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p + off; off += (bytes + 255) & ~(size_t)255; return r; };
    w.keys = (u64 *)take((size_t)batch * w.Np2 * 8);
    w.x1 = (float *)take((size_t)batch * w.Np * 4);
    w.y1 = (float *)take((size_t)batch * w.Np * 4);
    w.x2 = (float *)take((size_t)batch * w.Np * 4);
    w.y2 = (float *)take((size_t)batch * w.Np * 4);
    w.area = (float *)take((size_t)batch * w.Np * 4);
    w.order = (int32_t *)take((size_t)batch * w.Np * 4);
    w.mask = (u64 *)take((size_t)batch * w.Np * w.W * 8);
    w.rsum = (u64 *)take((size_t)batch * w.Np * w.SW * 8);
    w.ecap = (w.W >= NMS_CSR_MIN_W && w.W <= 1024) ? w.Np * 32 : 0;
    w.bcnt = (int32_t *)take((size_t)batch * w.W * 4);
    w.boff = (int32_t *)take((size_t)batch * (w.W + 1) * 4);
    w.cursor = (int32_t *)take((size_t)batch * w.W * 4);
    w.cflag = (int32_t *)take((size_t)batch * 4);
    w.ekey = (u32 *)take((size_t)batch * (w.ecap + 8) * 4);      // + CSR_SLACK
    w.eval = (u64 *)take((size_t)batch * (w.ecap + 8) * 8);
    // (the pruned path's 64 pair lists and 64 work lists per image live in the image's part of `mask`, which that path never
    //  touches: Np * W * 2 u32 words, 128 lists of Np * W / 64 entries)
    w.pruned = (w.W > 64 && w.W <= 1024) ? 1 : 0;
    w.lcap = (w.Np * w.W / 64) & ~(int64_t)3;
    w.lsh = 7;
    while ((64ll << w.lsh) < w.Np) ++w.lsh;
    if (w.pruned) {
        w.hdr = (struct NmsHdr *)take((size_t)batch * 64);
        w.bbox = (float4 *)take((size_t)batch * w.Np * 16);
        w.bar = (float2 *)take((size_t)batch * w.Np * 8);
        w.posu = (int32_t *)take((size_t)batch * w.Np * 4);
        w.start = (int32_t *)take((size_t)batch * (4096 + 1) * 4);       // XB_G + 1
        w.diag = (u64 *)take((size_t)batch * w.Np * 8);
        w.lcnt = (int32_t *)take((size_t)batch * 64 * 32 * 4);           // LC_STRIDE
        w.mc = (u32 *)take((size_t)batch * 4352 * 4);                    // MC_WORDS
    } else {
        w.hdr = nullptr; w.bbox = nullptr; w.bar = nullptr; w.posu = nullptr; w.start = nullptr; w.diag = nullptr;
        w.lcnt = nullptr; w.mc = nullptr;
    }
    w.total = off;
    return w;
}

// make_key (descending score <=> ascending key; ties -> lower index first; NaN first, as torch): common.hpp
__device__ __forceinline__ void gather_sorted(const float4 *__restrict__ boxes, u64 key, bool valid, int64_t p,
                                              float *x1, float *y1, float *x2, float *y2, float *area, int32_t *order)
{
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    int32_t idx = 0;
    if (valid) { idx = (int32_t)(u32)key; b = boxes[idx]; }
    x1[p] = b.x; y1[p] = b.y; x2[p] = b.z; y2[p] = b.w;
    area[p] = dg_area(b.x, b.y, b.z, b.w);
    order[p] = idx;
}

// ---------------------------------------------------------------- sort, segments <= 4096
template <int NP2, int T>
__global__ __launch_bounds__(T) void k_nms_sort_small(const float4 *__restrict__ boxes, const float *__restrict__ scores,
                                                      const int32_t *__restrict__ counts, int64_t nmax, int64_t Np,
                                                      float *x1, float *y1, float *x2, float *y2, float *area, int32_t *order)
{
    __shared__ u64 keys[NP2];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int n = nms_count(counts, img, nmax);
    boxes += (int64_t)img * nmax;
    scores += (int64_t)img * nmax;
    for (int i = tid; i < NP2; i += T) keys[i] = (i < n) ? make_key(scores[i], (u32)i) : ~0ull;
    __syncthreads();
    // A wave's 64 pair indices t touch one 128-key block as long as j <= 64, and the same block in every such step: those
    // steps need no workgroup barrier (LDS operations of one wave complete in order), only the steps that cross waves
    // (j >= 128) and the first wave-local step after one do -- 18 barriers instead of 66 for 2048 keys.
    bool crossed = false;                                        // the previous step crossed waves
    for (int k = 2; k <= NP2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool cross = j >= 128;
            if (cross || crossed) __syncthreads();
            else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int t = tid; t < NP2 / 2; t += T) {
                int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                int l = i | j;
                u64 a = keys[i], b = keys[l];
                bool asc = (i & k) == 0;
                if ((a > b) == asc) { keys[i] = b; keys[l] = a; }
            }
            crossed = cross;
        }
    }
    __syncthreads();
    const int64_t o = (int64_t)img * Np;
    for (int p = tid; p < (int)Np; p += T)
        gather_sorted(boxes, p < NP2 ? keys[p] : 0, p < n, o + p, x1, y1, x2, y2, area, order);
}

// ---------------------------------------------------------------- sort, long segments (global bitonic)
__global__ __launch_bounds__(256) void k_nms_make_keys(const float *__restrict__ scores, const int32_t *__restrict__ counts,
                                                       int64_t nmax, int64_t Np2, u64 *__restrict__ keys)
{
    const int img = blockIdx.y;
    const int n = nms_count(counts, img, nmax);
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < Np2) keys[(int64_t)img * Np2 + i] = (i < n) ? make_key(scores[(int64_t)img * nmax + i], (u32)i) : ~0ull;
}

// sorts/merges one SORT_CHUNK in LDS: stages k = kfirst..klast, and inside stage kfirst only j <= jfirst
__global__ __launch_bounds__(1024) void k_bitonic_lds(u64 *__restrict__ keys, int64_t Np2, int64_t kfirst, int64_t klast, int jfirst)
{
    __shared__ u64 s[SORT_CHUNK];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * SORT_CHUNK;
    u64 *g = keys + (int64_t)blockIdx.y * Np2 + base;
    for (int i = tid; i < SORT_CHUNK; i += 1024) s[i] = g[i];
    __syncthreads();
    for (int64_t k = kfirst; k <= klast; k <<= 1) {
        int j0 = (k == kfirst) ? jfirst : (int)(k >> 1);
        for (int j = j0; j > 0; j >>= 1) {
            for (int t = tid; t < SORT_CHUNK / 2; t += 1024) {
                int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                int l = i | j;
                u64 a = s[i], b = s[l];
                bool asc = ((base + i) & k) == 0;
                if ((a > b) == asc) { s[i] = b; s[l] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < SORT_CHUNK; i += 1024) g[i] = s[i];
}

__global__ __launch_bounds__(256) void k_bitonic_global(u64 *__restrict__ keys, int64_t Np2, int64_t k, int64_t j)
{
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= Np2 / 2) return;
    u64 *g = keys + (int64_t)blockIdx.y * Np2;
    int64_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    int64_t l = i | j;
    u64 a = g[i], b = g[l];
    bool asc = (i & k) == 0;
    if ((a > b) == asc) { g[i] = b; g[l] = a; }
}

// ---------------------------------------------------------------- x-bucket order of one image (pruned path, see k_nms_band)
#define XB_G 4096
__device__ __forceinline__ int xb_bucket(float x, float xmin, float scale)
{
    return (int)fminf(fmaxf((x - xmin) * scale, 0.0f), (float)(XB_G - 1));      // monotone in x; a NaN product (inf * 0) -> 0
}

// One workgroup of 1024 threads per image (n <= 65536: at most 64 boxes per thread): range of the left edges and non-finite
// boxes, bucket histogram in LDS, scan, every box's position in x-bucket order.  (Global-memory atomics are not an option here:
// device-scope atomics are carried out at the memory side on this part, about 8 ns each on one address -- a grid-wide histogram
// of 131 k of them took 25 us.)  The candidate-pair count that decides between this path and the dense mask is estimated from
// 1024 boxes spread over the image.
// It works on the boxes AS THE CALLER GAVE THEM (position by the caller's index, `posu`), not on the score-sorted copy: nothing in
// it depends on the ranks, so it runs as one more workgroup of k_nms_sort_chunks' launch, beside the 2..16 workgroups that sort
// the score keys, instead of as a launch of its own behind k_nms_merge_rank (22-26 us of a 64 k-box call's 185);
// k_nms_merge_rank, which knows both a box's index and its rank, carries the position over (`pos`, by rank).
// s_hist: XB_G ints, s_start: XB_G + 1 (the sort's key buffer, which this workgroup does not use otherwise).
__device__ __forceinline__ void xbucket_image(const float4 *__restrict__ bx, int n, int64_t Np, int force, NmsHdr *__restrict__ h,
                                              int32_t *__restrict__ gs, int32_t *__restrict__ posu, int32_t *__restrict__ lcnt_img,
                                              int *__restrict__ s_hist, int *__restrict__ s_start)
{
    __shared__ float s_mn[16], s_mx[16];
    __shared__ int s_bad[16], s_part[16];
    __shared__ unsigned long long s_cand;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64) lcnt_img[tid * LC_STRIDE] = 0;                  // the image's pair lists start empty
    float mn = INFINITY, mx = -INFINITY;
    int bad = 0;
    // a thread's boxes: tid, tid + 1024, ... (at most 64), sixteen loads in flight
#pragma unroll 1
    for (int k0 = 0; k0 < 64 && tid + 1024 * k0 < n; k0 += 16) {
        float4 b[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = tid + 1024 * (k0 + u);
            b[u] = i < n ? bx[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = tid + 1024 * (k0 + u);
            if (i < n) {                                          // a non-finite coordinate makes the area inf or NaN
                mn = fminf(mn, b[u].x); mx = fmaxf(mx, b[u].x); bad |= !(dg_area(b[u].x, b[u].y, b[u].z, b[u].w) < INFINITY);
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { mn = fminf(mn, __shfl_xor(mn, d, 64)); mx = fmaxf(mx, __shfl_xor(mx, d, 64)); bad |= __shfl_xor(bad, d, 64); }
    if (lane == 0) { s_mn[wave] = mn; s_mx[wave] = mx; s_bad[wave] = bad; }
    for (int t = tid; t < XB_G; t += 1024) s_hist[t] = 0;
    if (tid == 0) s_cand = 0ull;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) { mn = fminf(mn, s_mn[k]); mx = fmaxf(mx, s_mx[k]); bad |= s_bad[k]; }
    if (bad || force == 0 || n <= 64) {                           // uniform
        if (tid == 0) { h->xmin = 0.f; h->scale = 0.f; h->mode = 1; h->fpdone = 0; h->rounds = 0; h->npairs = 0; h->cpb = 0; h->estsum = 0u; }
        return;
    }
    const float range = mx - mn;
    float scale = (float)XB_G / range;
    if (!(range > 0.0f) || !(scale < INFINITY)) scale = 0.0f;    // everything in bucket 0: the candidate count then says "dense"
    // ONE pass of LDS atomics: the add that counts a box also tells its place inside its bucket; bucket (12 bits) and place
    // (20 bits) wait in registers (64 per thread, every index static) until the bucket starts are known.  (A counting pass and
    // a second, returning pass over the boxes read again took 25 us for 64 k boxes, two thirds of it the 131 k atomics.)
    u32 packed[64];
#pragma unroll
    for (int k0 = 0; k0 < 64; k0 += 16) {
        float xv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = tid + 1024 * (k0 + u);
            xv[u] = i < n ? bx[i].x : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = tid + 1024 * (k0 + u);
            u32 pk = 0u;
            if (i < n) {
                const int bk = xb_bucket(xv[u], mn, scale);
                pk = ((u32)bk << 20) | (u32)atomicAdd(&s_hist[bk], 1);           // (a bucket holds at most n <= 65536 < 2^20 boxes)
            }
            packed[k0 + u] = pk;
        }
    }
    __syncthreads();
    {   // exclusive scan of the 4096 counts: four per thread, wave scan, 16 wave totals
        const int c0 = s_hist[4 * tid], c1 = s_hist[4 * tid + 1], c2 = s_hist[4 * tid + 2], c3 = s_hist[4 * tid + 3];
        const int mine = c0 + c1 + c2 + c3;
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int k = 0; k < wave; ++k) base += s_part[k];
        int e = base + incl - mine;
        s_start[4 * tid] = e; gs[4 * tid] = e; e += c0;
        s_start[4 * tid + 1] = e; gs[4 * tid + 1] = e; e += c1;
        s_start[4 * tid + 2] = e; gs[4 * tid + 2] = e; e += c2;
        s_start[4 * tid + 3] = e; gs[4 * tid + 3] = e;
        if (tid == 0) { s_start[XB_G] = n; gs[XB_G] = n; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        const int i = tid + 1024 * k;
        if (i < n) posu[i] = s_start[packed[k] >> 20] + (int)(packed[k] & 0xfffffu);
    }
    {   // one sampled box per thread, spread over the image: the boxes whose left edge may lie inside its x range
        const int i = (int)(((long long)tid * n) >> 10);
        const float4 sb = bx[i];
        const int c = s_start[xb_bucket(fmaxf(sb.x, sb.z), mn, scale) + 1] - s_start[xb_bucket(sb.x, mn, scale)];
        unsigned long long cand = (unsigned long long)(c > 0 ? c : 0);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) cand += __shfl_xor(cand, d, 64);
        if (lane == 0) atomicAdd(&s_cand, cand);
    }
    __syncthreads();
    if (tid == 0) {
        // candidates per box ~ s_cand / 1024; the band kernel tests about 1.8 x as many pairs (a row tile runs to the end of
        // its widest box), the dense mask n / 2 per box
        const bool prune = force == 2 || s_cand * 4ull <= 1024ull * (unsigned long long)n;
        h->xmin = mn; h->scale = scale; h->mode = prune ? 0 : 1; h->fpdone = 0; h->rounds = 0; h->npairs = 0;
        h->cpb = (int32_t)(s_cand >> 10); h->estsum = 0u;
    }
}

// ---------------------------------------------------------------- sort, 4096 < n <= 65536: sorted chunks + rank merge
// Two launches instead of the bitonic network's 17 (for 64 k keys about 100 us, most of it launch-to-launch): every
// SORT_CHUNK of keys is built and sorted in LDS (ascending), then each key finds its final rank directly -- its place in its
// own chunk plus, for every other chunk, the number of keys there that go before it (binary search: 6 steps over the chunk's
// 64 block maxima in LDS, 6 inside one 64-key block in memory; equal keys -- the padding -- go by chunk order) -- and the
// box is gathered straight to that rank.
__global__ __launch_bounds__(1024) void k_nms_sort_chunks(const float *__restrict__ scores, const int32_t *__restrict__ counts,
                                                          int64_t nmax, int64_t Np2, u64 *__restrict__ keys,
                                                          const float4 *__restrict__ boxes, int64_t Np, int force, NmsHdr *__restrict__ hdr,
                                                          int32_t *__restrict__ start, int32_t *__restrict__ posu, int32_t *__restrict__ lcnt)
{
    // Round 3, second form: the keys live in REGISTERS, four consecutive ones per thread (element 4 tid + r).  Of the 78
    // compare-exchange steps of the bitonic network, 23 have their partner in the same thread (j = 1, 2: no data movement),
    // 45 in the same wave (j = 4 .. 128 elements = 1 .. 32 lanes: a cross-lane exchange), and only 10 cross waves
    // (j >= 256: through LDS, two workgroup barriers each).  The LDS version before it made every step a round trip of
    // reads and conditional writes (15 of them with a barrier) and took 31 us for 64 k keys on its 16 workgroups.
    __shared__ __attribute__((aligned(16))) u64 s[SORT_CHUNK + 8];
    const int tid = threadIdx.x, img = blockIdx.y;
    const int n = nms_count(counts, img, nmax);
    if ((int64_t)blockIdx.x * SORT_CHUNK >= Np2) {
        // the launch's extra workgroup (pruned path only): the image's x-bucket order, from the caller's boxes (xbucket_image)
        xbucket_image(boxes + (int64_t)img * nmax, n, Np, force, hdr + img, start + (int64_t)img * (XB_G + 1), posu + (int64_t)img * Np,
                      lcnt + (int64_t)img * 64 * LC_STRIDE, (int *)s, (int *)s + XB_G);
        return;
    }
    const int64_t base = (int64_t)blockIdx.x * SORT_CHUNK;
    u64 *g = keys + (int64_t)img * Np2 + base;
    u64 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = base + 4 * tid + r;
        v[r] = (i < n) ? make_key(scores[(int64_t)img * nmax + i], (u32)i) : ~0ull;
    }
    // element i keeps the smaller key of its pair iff it is the lower element of an ascending pair or the upper of a descending one
    auto keep = [](u64 mine, u64 other, bool takemin) { return ((other < mine) == takemin) ? other : mine; };
    const int e0 = 4 * tid;
#pragma unroll 1
    for (int k = 2; k <= SORT_CHUNK; k <<= 1) {
#pragma unroll 1
        for (int j = k >> 1; j >= 256; j >>= 1) {                   // partners in other waves
            __syncthreads();                                         // (the reads of the step before are done)
            *(uint4 *)&s[e0] = make_uint4((u32)v[0], (u32)(v[0] >> 32), (u32)v[1], (u32)(v[1] >> 32));
            *(uint4 *)&s[e0 + 2] = make_uint4((u32)v[2], (u32)(v[2] >> 32), (u32)v[3], (u32)(v[3] >> 32));
            __syncthreads();
            const int p0 = e0 ^ j;
            const uint4 a = *(const uint4 *)&s[p0], b = *(const uint4 *)&s[p0 + 2];
            const u64 o[4] = {((u64)a.y << 32) | a.x, ((u64)a.w << 32) | a.z, ((u64)b.y << 32) | b.x, ((u64)b.w << 32) | b.z};
            const bool takemin = ((e0 & j) == 0) == ((e0 & k) == 0);   // (j, k >= 256: the same for the thread's four elements)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = keep(v[r], o[r], takemin);
        }
#pragma unroll 1
        for (int j = (k >> 1) < 128 ? (k >> 1) : 128; j >= 4; j >>= 1) {   // partners in other lanes of this wave
            const int d = j >> 2;
            const bool takemin = ((e0 & j) == 0) == ((e0 & k) == 0);   // (j >= 4 and k >= 8: the same for the four elements)
            const int addr = ((tid & 63) ^ d) << 2;                  // the eight exchanges of a step are issued before the first is waited for
            u32 lo[4], hi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lo[r] = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)(u32)v[r]);
                hi[r] = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)(u32)(v[r] >> 32));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = keep(v[r], ((u64)hi[r] << 32) | lo[r], takemin);
        }
        if (k >= 4) {                                                // j = 2: (0, 2) and (1, 3)
            const bool asc = ((e0 & k) == 0);                        // k = 4: bit 2 of the element = bit 0 of tid; in general bits >= 2
            const u64 a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
            const bool s02 = (a0 > a2) == asc, s13 = (a1 > a3) == asc;
            v[0] = s02 ? a2 : a0; v[2] = s02 ? a0 : a2; v[1] = s13 ? a3 : a1; v[3] = s13 ? a1 : a3;
        }
        {                                                            // j = 1: (0, 1) and (2, 3)
            // k = 2: element e ascends iff bit 1 of e is clear: the pair (0, 1) ascends, (2, 3) descends; k >= 4: by the thread
            const bool asc01 = k == 2 ? true : ((e0 & k) == 0), asc23 = k == 2 ? false : ((e0 & k) == 0);
            const u64 a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
            const bool s01 = (a0 > a1) == asc01, s23 = (a2 > a3) == asc23;
            v[0] = s01 ? a1 : a0; v[1] = s01 ? a0 : a1; v[2] = s23 ? a3 : a2; v[3] = s23 ? a2 : a3;
        }
    }
    *(uint4 *)&g[e0] = make_uint4((u32)v[0], (u32)(v[0] >> 32), (u32)v[1], (u32)(v[1] >> 32));
    *(uint4 *)&g[e0 + 2] = make_uint4((u32)v[2], (u32)(v[2] >> 32), (u32)v[3], (u32)(v[3] >> 32));
}

__global__ __launch_bounds__(256) void k_nms_merge_rank(const float4 *__restrict__ boxes, const u64 *__restrict__ keys,
                                                        const int32_t *__restrict__ counts, int64_t nmax, int64_t Np, int64_t Np2,
                                                        float *x1, float *y1, float *x2, float *y2, float *area, int32_t *order,
                                                        const int32_t *__restrict__ posu, const NmsHdr *__restrict__ hdr,
                                                        const int32_t *__restrict__ start, float4 *__restrict__ bbox, float2 *__restrict__ bar)
{
    __shared__ u64 s_max[16][64];                                 // last key of every 64-key block of every chunk
    const int img = blockIdx.y, tid = threadIdx.x;
    const int n = nms_count(counts, img, nmax);
    const u64 *kimg = keys + (int64_t)img * Np2;
    const int nchunks = (int)(Np2 / SORT_CHUNK);                  // 2 .. 16
    for (int t = tid; t < nchunks * 64; t += 256) s_max[t >> 6][t & 63] = kimg[(int64_t)(t >> 6) * SORT_CHUNK + (t & 63) * 64 + 63];
    __syncthreads();
    const int64_t e = (int64_t)blockIdx.x * 256 + tid;
    const int c = (int)(e / SORT_CHUNK);
    const u64 K = kimg[e];
    int rank = (int)(e % SORT_CHUNK);
    int blk[16];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {                             // blocks of chunk cc that lie wholly before K
        int b = 0;
        if (cc < nchunks && cc != c) {
#pragma unroll
            for (int st = 32; st > 0; st >>= 1) {
                const u64 v = s_max[cc][b + st - 1];
                if (cc < c ? v <= K : v < K) b += st;
            }
            if (b == 63) { const u64 v = s_max[cc][63]; if (cc < c ? v <= K : v < K) b = 64; }
        }
        blk[cc] = b;
    }
    int in[16];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) in[cc] = 0;
    // and the keys of the next block, in memory: a 4-ary search -- three pivots per chunk and step, 64 -> 16 -> 4 -> 1, the (at
    // most 45) loads of a step in flight together: three dependent trips to the L2 instead of the six of a binary search (each
    // of them 1.5-2 us with every wave of the launch waiting on one)
#pragma unroll
    for (int q = 16; q > 0; q >>= 2) {
        u64 v[3][16];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int cc = 0; cc < 16; ++cc)
                v[t][cc] = (cc < nchunks && cc != c && blk[cc] < 64) ? kimg[(int64_t)cc * SORT_CHUNK + blk[cc] * 64 + in[cc] + (t + 1) * q - 1] : ~0ull;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc)
            if (cc < nchunks && cc != c && blk[cc] < 64) {
                int cnt = 0;
#pragma unroll
                for (int t = 0; t < 3; ++t) cnt += (cc < c ? v[t][cc] <= K : v[t][cc] < K) ? 1 : 0;    // sorted: the pivots before K are a prefix
                in[cc] += cnt * q;
            }
    }
#pragma unroll
    for (int cc = 0; cc < 16; ++cc)
        if (cc < nchunks && cc != c) rank += blk[cc] * 64 + in[cc];
    // (inside a block the search covers offsets 0 .. 62 + one more step: the block's last key is not before K, or blk would be larger)
    if (rank < Np) {
        // gather_sorted, and for an image of the pruned path (hdr.mode == 0) the box also goes to its place in x-bucket order --
        // found by index beside the chunk sort (xbucket_image); the zero boxes of the padding stay behind the real ones -- as
        // bbox = corners, bar = {area, rank}.  With the rank (< 65536) goes the last 64-box tile of that order that can hold a
        // partner of this box: the one where the bucket of its right edge ends.  (Was a launch of its own, k_nms_xb_scatter.)
        const bool valid = rank < n;
        const int64_t o = (int64_t)img * Np;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        int32_t idx = 0;
        if (valid) { idx = (int32_t)(u32)K; b = boxes[(int64_t)img * nmax + idx]; }
        const float ar = dg_area(b.x, b.y, b.z, b.w);
        x1[o + rank] = b.x; y1[o + rank] = b.y; x2[o + rank] = b.z; y2[o + rank] = b.w;
        area[o + rank] = ar;
        order[o + rank] = idx;
        if (posu && hdr[img].mode == 0) {
            const NmsHdr *h = hdr + img;
            int p = rank, endtile = 0;
            if (valid) {
                p = posu[o + idx];
                endtile = (start[(int64_t)img * (XB_G + 1) + xb_bucket(fmaxf(b.x, b.z), h->xmin, h->scale) + 1] - 1) >> 6;
            }
            bbox[o + p] = b;
            bar[o + p] = make_float2(ar, __int_as_float(rank | (endtile << 16)));
        }
    }
}

__global__ __launch_bounds__(256) void k_nms_gather(const float4 *__restrict__ boxes, const u64 *__restrict__ keys,
                                                    const int32_t *__restrict__ counts, int64_t nmax, int64_t Np, int64_t Np2,
                                                    float *x1, float *y1, float *x2, float *y2, float *area, int32_t *order)
{
    const int img = blockIdx.y;
    const int n = nms_count(counts, img, nmax);
    int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= Np) return;                                     // Np is a multiple of 64: whole waves leave
    gather_sorted(boxes + (int64_t)img * nmax, keys[(int64_t)img * Np2 + p], p < n, (int64_t)img * Np + p, x1, y1, x2, y2, area, order);
}

// ---------------------------------------------------------------- mask
// grid: x = column tile (MTW = 32 words = 2048 columns; 8 words for small launches), y = 64-row block, z = image.
// 256 threads; a wave owns MTW / 4 of the tile's word-columns (32-word tiles: 17 KB of LDS, eight workgroups = 8 waves per
// SIMD on a CU; with 64-word tiles and four workgroups per CU the 64k mask took 1.31 ms instead of 1.16-1.20).  One lane
// owns one column; the row box is an LDS broadcast.
//
// The test !(inter/den < thr) is done without the division.  With p = the float below thr and
// m = (p+thr)/2, a correctly rounded quotient is < thr exactly when the real quotient is below m
// (or equal to m when the tie rounds down to p, i.e. thr's mantissa is odd).
//   fast path (2^-20 <= thr < inf, all areas finite): e1 = fma(-thr,den,inter) and e2 = fma(-p,den,inter)
//   carry the exact signs of inter-thr*den and inter-p*den (single rounding, no underflow in this
//   range).  e1 >= 0 -> suppressed; e2 <= 0 -> kept; only p < q < thr (one ulp wide) needs more, and
//   a wave that sees such a lane redoes that row exactly;
//   exact path: m has 25 significant bits and den 24, so m*den is exact in double and so is the compare.
//   NaNs fall to "suppress" through the negated compare, like the division did.
//
// Besides the mask words, every row gets a summary: bit w of rsum[row][t] says word 64t+w is non-zero.
// The reduce phase walks summaries, so it touches only the non-zero words of the rows it keeps.
// nms_suppressed<TIE_EVEN>(inter, den, mid): common.hpp

// v_min/v_max as such: fminf/fmaxf would first canonicalise every operand (one more VALU op each); with
// quiet NaNs the result is the same, the non-NaN operand
__device__ __forceinline__ float vmin(float a, float b) { float o; asm("v_min_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b)); return o; }
__device__ __forceinline__ float vmax(float a, float b) { float o; asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b)); return o; }

template <bool TIE_EVEN, int MTW>
__global__ __launch_bounds__(256, 4) void k_nms_mask(const float *__restrict__ sx1, const float *__restrict__ sy1,
                                                  const float *__restrict__ sx2, const float *__restrict__ sy2,
                                                  const float *__restrict__ sarea, const int32_t *__restrict__ counts,
                                                  int64_t nmax, int64_t Np, int64_t W, int64_t SW, float thr, float thr_below,
                                                  double mid, int fast_ok, u64 *__restrict__ mask, u64 *__restrict__ rsum,
                                                  const NmsHdr *__restrict__ hdr)
{
    __shared__ float4 rbox[64];
    __shared__ float rarea[64];
    __shared__ u64 tile[64][MTW + 1];
    const int img = blockIdx.z;
    if (hdr && hdr[img].mode != 1) return;                    // the pruned path does this image
    const int n = nms_count(counts, img, nmax);
    const int ib = blockIdx.y;
    const int jt = blockIdx.x, jb0 = jt * MTW;
    if (ib * 64 >= n || (int64_t)jb0 * 64 >= n || jb0 + MTW <= ib) return;   // below the diagonal / outside
    const int64_t o = (int64_t)img * Np;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64) {
        const int64_t i = o + ib * 64 + tid;
        rbox[tid] = make_float4(sx1[i], sy1[i], sx2[i], sy2[i]);
        rarea[tid] = sarea[i];
    }
    __syncthreads();
    const bool rowbad = __ballot(!(rarea[lane] < INFINITY)) != 0ull;
#pragma unroll 1
    for (int q = 0; q < MTW / 4; ++q) {
        const int wi = wave + 4 * q;
        const int jb = jb0 + wi;
        asm volatile("" ::: "memory");                             // keep the row reads inside the loop, not in 320 registers
        if (jb >= ib && jb < W && (int64_t)jb * 64 < n) {          // wave-uniform
            const int64_t j = o + (int64_t)jb * 64 + lane;
            const float cx1 = sx1[j], cy1 = sy1[j], cx2 = sx2[j], cy2 = sy2[j], ca = sarea[j];
            const bool colbad = __ballot(!(ca < INFINITY)) != 0ull;
            bool exact = !(fast_ok && !rowbad && !colbad);
            if (!exact) {
                u32 wlo = 0, whi = 0;
                u64 ambany = 0;
#pragma unroll
                for (int r = 0; r < 64; ++r) {
                    const float4 p = rbox[r];
                    const float ix = vmax(vmin(p.z, cx2) - vmax(p.x, cx1), 0.0f);
                    const float iy = vmax(vmin(p.w, cy2) - vmax(p.y, cy1), 0.0f);
                    const float inter = ix * iy;
                    const float den = ((rarea[r] + ca) - inter) + 1e-7f;
                    const float e1 = __builtin_fmaf(-thr, den, inter), e2 = __builtin_fmaf(-thr_below, den, inter);
                    const u64 bal = __ballot(!(e1 < 0.0f));
                    ambany |= __ballot(e2 > 0.0f) & ~bal;               // p < q < thr somewhere in this row
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(wlo) : "s"((u32)bal), "n"(r));
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(whi) : "s"((u32)(bal >> 32)), "n"(r));
                }
                u64 myword = ((u64)whi << 32) | wlo;
                if (jb == ib) myword &= ~((2ull << lane) - 1ull);   // only j > i
                tile[lane][wi] = myword;
                exact = ambany != 0ull;                              // rare: redo this word with the exact test
            }
            if (exact) {
#pragma unroll 2
                for (int r = 0; r < 64; ++r) {
                    const float4 p = rbox[r];
                    const float ix = fmaxf(fminf(p.z, cx2) - fmaxf(p.x, cx1), 0.0f);
                    const float iy = fmaxf(fminf(p.w, cy2) - fmaxf(p.y, cy1), 0.0f);
                    const float inter = ix * iy;
                    const float den = ((rarea[r] + ca) - inter) + 1e-7f;
                    u64 bal = __ballot(nms_suppressed<TIE_EVEN>(inter, den, mid));   // yolo.py:201 keeps iou < thr
                    if (jb == ib) bal &= ~((2ull << r) - 1ull);
                    if (lane == 0) tile[r][wi] = bal;
                }
            }
        } else {
            tile[lane][wi] = 0ull;
        }
    }
    __syncthreads();
    if (tid < 64) {
        u32 sbits = 0;
#pragma unroll 8
        for (int w = 0; w < MTW; ++w) sbits |= (u32)(tile[tid][w] != 0ull) << w;
        // this tile's part of the row's summary word (little endian: 64 / MTW parts of MTW bits); the last tile that has
        // columns also zeroes the parts after its own (those tiles return at the top, or are not in the grid)
        typedef typename std::conditional<MTW == 32, u32, unsigned char>::type part_t;
        constexpr int PER = 64 / MTW;
        part_t *ps = reinterpret_cast<part_t *>(rsum + (o + (int64_t)ib * 64 + tid) * SW + jt / PER);
        const int pi = jt % PER;
        ps[pi] = (part_t)sbits;
        if ((int64_t)(jb0 + MTW) * 64 >= n)
            for (int z = pi + 1; z < PER; ++z) ps[z] = 0;
    }
    // 64 rows x MTW words out, MTW * 8 contiguous bytes per row, 64 / MTW rows per wave pass
    u64 *mo = mask + (o + (int64_t)ib * 64) * W + jb0;
    constexpr int RPP = 64 / MTW;
    const int col = lane % MTW, sub = lane / MTW;
    if (jb0 + col < W) {
#pragma unroll
        for (int p = 0; p < MTW / 4; ++p) {
            const int row = (wave + 4 * p) * RPP + sub;
            mo[(int64_t)row * W + col] = tile[row][col];
        }
    }
}

// ---------------------------------------------------------------- reduce
__device__ __forceinline__ u64 uniform64(u64 v)
{
    u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 readlane64(u64 v, int l)
{
    u32 lo = __builtin_amdgcn_readlane((u32)v, l), hi = __builtin_amdgcn_readlane((u32)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}

// the serial chain over 64-row blocks straight from the dense mask: the body of k_nms_reduce, also what
// k_nms_resolve_small falls back to (every pointer already at the image)
template <int T, int WPT>
__device__ __forceinline__ void nms_reduce_serial(const u64 *__restrict__ mask, const int32_t *__restrict__ order, int n, int64_t W,
                                                  int quirk, int64_t *__restrict__ keep, int32_t *__restrict__ keep_count_img)
{
    __shared__ u64 s_remv, s_kept, s_curb;
    const int tid = threadIdx.x;

    u64 remv[WPT], prev[WPT];
#pragma unroll
    for (int q = 0; q < WPT; ++q) {
        int64_t base = ((int64_t)tid + (int64_t)q * T) * 64;
        remv[q] = (base >= n) ? ~0ull : ((n - base >= 64) ? 0ull : (~0ull << (n - base)));
        prev[q] = remv[q];
    }
    int total = 0, last_ib = -1, last_r = 0;
    u64 last_curb = 0;
    const int nb = (n + 63) >> 6;
    for (int ib = 0; ib < nb; ++ib) {
        {   // owner of word ib publishes it
            const int own = ib % T, oq = ib / T;
#pragma unroll
            for (int q = 0; q < WPT; ++q)
                if (q == oq && tid == own) s_remv = remv[q];
        }
        __syncthreads();
        if (tid < 64) {
            const u64 diag = mask[((int64_t)ib * 64 + tid) * W + ib];
            u64 cur = uniform64(s_remv), kept = 0, curb = cur;
            u64 alive = ~cur;
            while (alive) {
                int r = __builtin_ctzll(alive);
                kept |= 1ull << r;
                curb = cur;
                cur |= readlane64(diag, r);
                alive = ~cur & ~((2ull << r) - 1ull);
            }
            if ((kept >> tid) & 1ull) {
                int pos = total + __builtin_popcountll(kept & ((1ull << tid) - 1ull));
                keep[pos] = (int64_t)order[ib * 64 + tid];
            }
            if (tid == 0) { s_kept = kept; s_curb = curb; }
        }
        __syncthreads();
        const u64 kept = uniform64(s_kept);
        if (kept) {
            const int rl = 63 - __builtin_clzll(kept);
            last_ib = ib; last_r = rl; last_curb = uniform64(s_curb);
            total += __builtin_popcountll(kept);
            const u64 others = kept & ~(1ull << rl);
#pragma unroll
            for (int q = 0; q < WPT; ++q) {
                const int64_t w = (int64_t)tid + (int64_t)q * T;
                if (w > ib && w < W) {
                    const u64 *rowbase = mask + (int64_t)ib * 64 * W + w;
                    u64 acc = 0, k = others;
                    while (k) {
                        u64 v0, v1 = 0, v2 = 0, v3 = 0;
                        int r = __builtin_ctzll(k); k &= k - 1; v0 = rowbase[(int64_t)r * W];
                        if (k) { r = __builtin_ctzll(k); k &= k - 1; v1 = rowbase[(int64_t)r * W]; }
                        if (k) { r = __builtin_ctzll(k); k &= k - 1; v2 = rowbase[(int64_t)r * W]; }
                        if (k) { r = __builtin_ctzll(k); k &= k - 1; v3 = rowbase[(int64_t)r * W]; }
                        acc |= (v0 | v1) | (v2 | v3);
                    }
                    u64 vl = rowbase[(int64_t)rl * W];
                    remv[q] |= acc;
                    prev[q] = remv[q];
                    remv[q] |= vl;
                }
            }
        }
    }
    if (quirk && total > 0) {
        // yolo.py:196 `while ids_sorted.numel()>1`: the last kept box km is dropped iff nothing ranked
        // below it was still alive when it became the head, i.e. before its own row was applied.
        int surv = 0;
        const u64 above = (last_r == 63) ? 0ull : (~0ull << (last_r + 1));
        if (tid == 0) surv = ((~last_curb) & above) != 0;
#pragma unroll
        for (int q = 0; q < WPT; ++q) {
            const int64_t w = (int64_t)tid + (int64_t)q * T;
            if (w > last_ib && w < W) surv |= (~prev[q]) != 0ull;
        }
        surv = __syncthreads_or(surv);
        if (!surv) total -= 1;
    }
    if (tid == 0) *keep_count_img = total;
}

template <int T, int WPT>
__global__ __launch_bounds__(T) void k_nms_reduce(const u64 *__restrict__ mask, const int32_t *__restrict__ order,
                                                  const int32_t *__restrict__ counts, int64_t nmax, int64_t Np, int64_t W,
                                                  int quirk, int64_t *__restrict__ keep, int32_t *__restrict__ keep_count)
{
    const int img = blockIdx.x;
    const int n = nms_count(counts, img, nmax);
    nms_reduce_serial<T, WPT>(mask + (int64_t)img * Np * W, order + (int64_t)img * Np, n, W, quirk, keep + (int64_t)img * nmax,
                              keep_count + img);
}

// ---------------------------------------------------------------- fixed-point keep list from the dense mask (8 <= W <= 64)
// Images of 449 .. 4096 boxes (one 2000-proposal image, config 1; the 256 x 2000 batch, config 3).  The serial chain over
// 64-row blocks paid a compaction launch (k_nms_csr_build_small, 13 us) and about 0.7 us per block (25-50 us for one
// 2000-box image); here ONE launch gathers the non-zero mask words of the image into LDS -- (row << 16 | word, bits), found
// through the row summaries, every word's load in flight at once -- and then decides the boxes in ROUNDS like
// k_nms_resolve_mc (see there), all entries in registers, the four bit sets in LDS: a round is two barriers.  More entries
// than that holds (8192: a 2000-box image at thr 0.3) and the rounds run with the words streamed from the mask instead (a
// thread owns rows; 87 -> 54 us against the serial chain); no end after max_rounds, and the serial chain over the dense mask
// finishes the image.
#define RS_CAP 8192
#define RS_PER (RS_CAP / 1024)
__global__ __launch_bounds__(1024) void k_nms_resolve_small(const u64 *__restrict__ mask, const u64 *__restrict__ rsum,
                                                            const int32_t *__restrict__ order, const int32_t *__restrict__ counts,
                                                            int64_t nmax, int64_t Np, int64_t W, int max_rounds, int quirk,
                                                            int64_t *__restrict__ keep, int32_t *__restrict__ keep_count)
{
    __shared__ u32 e_key[RS_CAP];                                // (row << 16 | word) of the non-zero words, in any order
    __shared__ u64 K[64], R[64], R2[64], Bk[64];
    __shared__ int s_cnt, s_last, s_off[64];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = nms_count(counts, img, nmax);
    const int nb = (n + 63) >> 6;
    mask += (int64_t)img * Np * W;
    rsum += (int64_t)img * Np;                                    // one summary word per row (W <= 64)
    order += (int64_t)img * Np;
    keep += (int64_t)img * nmax;
    u64 valid = 0;
    if (tid < 64) {
        const int64_t vb = (int64_t)tid * 64;
        valid = (vb >= n) ? 0ull : ((n - vb >= 64) ? ~0ull : ((1ull << (n - vb)) - 1ull));
        K[tid] = 0ull; R[tid] = 0ull; R2[tid] = 0ull; Bk[tid] = 0ull;
    }
    if (tid == 0) { s_cnt = 0; s_last = -1; }
    // the `order` of the boxes this lane may have to write at the end (words wave, wave + 16, ...): fetched now
    int ord[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int w = wave + 16 * u; ord[u] = (w < nb) ? order[w * 64 + lane] : 0; }
    u64 bits[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = tid + 1024 * k;
        bits[k] = (row < n) ? rsum[row] : 0ull;
    }
    __syncthreads();
    bool over = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = tid + 1024 * k;
        u64 b = (row < n) ? (bits[k] & ~((1ull << (row >> 6)) - 1ull)) : 0ull;     // words left of the diagonal were never written
        const int c = __builtin_popcountll(b);
        int at = c ? atomicAdd(&s_cnt, c) : 0;
        while (b) {
            const int w = __builtin_ctzll(b);
            b &= b - 1;
            if (at < RS_CAP) e_key[at] = ((u32)row << 16) | (u32)w;
            ++at;
        }
    }
    __syncthreads();
    const int E = s_cnt;
    over = E > RS_CAP;
    u32 key[RS_PER];
    u64 val[RS_PER];
    u32 alive = 0;
    if (!over) {
#pragma unroll
        for (int q = 0; q < RS_PER; ++q) {                        // every mask word of the image that matters, all loads together
            const int e = tid + 1024 * q;
            key[q] = e < E ? e_key[e] : 0u;
            val[q] = e < E ? mask[(int64_t)(key[q] >> 16) * W + (key[q] & 0xffffu)] : 0ull;
            if (e < E) alive |= 1u << q;
        }
    }
    int rounds = 0;
    bool bail = false;
    if (over) {
        // More non-zero words than the registers hold (a 2000-box image at thr 0.3): the same rounds with the words STREAMED --
        // a thread owns the rows tid, tid + 1024, ... and loads their non-zero words from the mask (L2-resident, eight loads
        // in flight) every time it needs them: once per round while the row is undecided, once more when it is kept.
        u32 ralive = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = tid + 1024 * k;
            bits[k] = (row < n) ? (bits[k] & ~((1ull << (row >> 6)) - 1ull)) : 0ull;
            if (bits[k]) ralive |= 1u << k;
        }
        auto for_words = [&](int row, u64 b, auto &&fn) {       // fn(word index, mask word) for every set bit of b
            const u64 *mrow = mask + (int64_t)row * W;
            while (b) {
                int wi[8];
                u64 v[8];
                int c = 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) { wi[u] = 0; if (b) { wi[u] = __builtin_ctzll(b); b &= b - 1; c = u + 1; } }
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (u < c) ? mrow[wi[u]] : 0ull;
#pragma unroll
                for (int u = 0; u < 8; ++u) if (u < c) fn(wi[u], v[u]);
            }
        };
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((ralive >> k) & 1u)
                for_words(tid + 1024 * k, bits[k], [&](int w, u64 v) { if (v) atomicOr((unsigned long long *)&Bk[w], (unsigned long long)v); });
        __syncthreads();
        if (tid < 64) { K[tid] = valid & ~Bk[tid]; Bk[tid] = 0ull; }
        for (;;) {
            if (!__syncthreads_or(ralive != 0u)) break;
            if (rounds >= max_rounds) { bail = true; break; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!((ralive >> k) & 1u)) continue;
                const int row = tid + 1024 * k;
                const u64 rb = 1ull << (row & 63);
                if (K[row >> 6] & rb) {
                    for_words(row, bits[k], [&](int w, u64 v) {
                        const u64 old = atomicOr((unsigned long long *)&R[w], (unsigned long long)v);
                        if (old & v) atomicOr((unsigned long long *)&R2[w], (unsigned long long)(old & v));
                    });
                    ralive &= ~(1u << k);
                } else if (R[row >> 6] & rb) {
                    ralive &= ~(1u << k);
                } else {
                    for_words(row, bits[k], [&](int w, u64 v) { if (v) atomicOr((unsigned long long *)&Bk[w], (unsigned long long)v); });
                }
            }
            __syncthreads();
            if (tid < 64) { K[tid] |= valid & ~(K[tid] | R[tid] | Bk[tid]); Bk[tid] = 0ull; }
            ++rounds;
        }
    }
    if (!over) {
        // round 0: boxes nothing suppresses are kept
#pragma unroll
        for (int q = 0; q < RS_PER; ++q)
            if ((alive >> q) & 1u) atomicOr((unsigned long long *)&Bk[key[q] & 0xffffu], (unsigned long long)val[q]);
        __syncthreads();
        if (tid < 64) { K[tid] = valid & ~Bk[tid]; Bk[tid] = 0ull; }
        for (;;) {
            if (!__syncthreads_or(alive != 0u)) break;
            if (rounds >= max_rounds) { bail = true; break; }
#pragma unroll
            for (int q = 0; q < RS_PER; ++q) {
                if (!((alive >> q) & 1u)) continue;
                const u32 row = key[q] >> 16, w = key[q] & 0xffffu;
                const u64 rb = 1ull << (row & 63);
                if (K[row >> 6] & rb) {
                    const u64 old = atomicOr((unsigned long long *)&R[w], (unsigned long long)val[q]);
                    if (old & val[q]) atomicOr((unsigned long long *)&R2[w], (unsigned long long)(old & val[q]));
                    alive &= ~(1u << q);
                } else if (R[row >> 6] & rb) {
                    alive &= ~(1u << q);
                } else {
                    atomicOr((unsigned long long *)&Bk[w], (unsigned long long)val[q]);
                }
            }
            __syncthreads();
            if (tid < 64) { K[tid] |= valid & ~(K[tid] | R[tid] | Bk[tid]); Bk[tid] = 0ull; }
            ++rounds;
        }
    }
    if (bail) {                                                  // uniform
        __syncthreads();
        nms_reduce_serial<1024, 1>(mask, order, n, W, quirk, keep, keep_count + img);
        return;
    }
    // ---- the kept set is final: count, quirk, keep list in rank order
    const u64 kw = (tid < 64) ? K[tid] : 0ull;
    const int mine = __builtin_popcountll(kw);
    int incl = mine;
    if (wave == 0) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
        s_off[lane] = incl - mine;
        if (kw) atomicMax(&s_last, tid * 64 + 63 - __builtin_clzll(kw));
        if (lane == 63) s_cnt = incl;
    }
    __syncthreads();
    const int total = s_cnt, km = s_last;
    int drop = 0;
    if (quirk && total > 0) {                                    // uniform
        // yolo.py:196 `while ids_sorted.numel()>1`: the last kept box goes iff none of its own targets had only it as kept
        // suppressor (every other lower-ranked box was already removed by the kept boxes before it)
        if (over) {
            if (km >= 0 && (km & 1023) == tid) {                   // the thread that owns the last kept box's row
                const int k = km >> 10;
                const u64 *mrow = mask + (int64_t)km * W;
                u64 b = bits[0];
                if (k == 1) b = bits[1]; else if (k == 2) b = bits[2]; else if (k == 3) b = bits[3];
                while (b) { const int w = __builtin_ctzll(b); b &= b - 1; atomicOr((unsigned long long *)&Bk[w], (unsigned long long)mrow[w]); }
            }
        } else {
#pragma unroll
            for (int q = 0; q < RS_PER; ++q)
                if ((int)(key[q] >> 16) == km && tid + 1024 * q < E) atomicOr((unsigned long long *)&Bk[key[q] & 0xffffu], (unsigned long long)val[q]);
        }
        __syncthreads();
        drop = !__syncthreads_or(tid < 64 && (Bk[tid] & ~R2[tid]) != 0ull);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int w = wave + 16 * u;
        if (w < nb) {
            u64 k = K[w];
            if (drop && w == (km >> 6)) k &= ~(1ull << (km & 63));
            if ((k >> lane) & 1ull) keep[s_off[w] + __builtin_popcountll(k & ((1ull << lane) - 1ull))] = (int64_t)ord[u];
        }
    }
    if (tid == 0) keep_count[img] = total - drop;
}

// ---------------------------------------------------------------- reduce for DENSE masks (8 <= W <= 1024, compact form overflowed)
// Round 1's kernel here (k_nms_reduce_sum: row summaries, only the non-zero words of kept rows) paid one HBM round trip
// per 64-row block: the words of the rows it keeps can be fetched only once the block is resolved (2.43 ms for 64k dense
// boxes at thr 0.5; this kernel 0.92 ms).  But the next blocks need only the NEAR words of those rows.  So the chain runs
// over SUPER-BLOCKS of 8 blocks (512 rows), in two concurrent roles:
//   band:     the 512 x 16 words [rows of super-block S] x [word columns of S and S+1] depend on nothing: they are fetched a
//             super-block ahead (8 words per thread in registers) and live in LDS (64 KB), with the rows' `order`;
//   resolver: wave 0 alone resolves the 8 blocks of S from the band -- diagonal word by v_readlane chain, hopping over the
//             rows that are alive and suppress something in their own block; the kept rows' near words are OR-ed into
//             `removed` with LDS atomics, which one wave sees in order: no barrier inside a super-block;
//   far side: meanwhile the other 15 waves fetch the FAR words (>= (S+1)*8, first needed by S+1) of everything S-1 kept:
//             thread = word column, 16 independent loads in flight, one HBM round trip per 512 rows instead of one per 64.
// One barrier pair per super-block; whichever side is slower sets its time (here the resolver: ~0.55 us per block, about
// 300 instructions of a lone wave).  `removed1` (everything but the last kept box, for the quirk of yolo.py:196) is kept
// by the far side alone: the last kept row of a super-block stays pending until a later super-block keeps something.
// A serial version of the same idea (resolve, barrier, far, barrier; band one super-block wide) took 1.17 ms.
#define BAND_BLOCKS 8
#define PIPE_NEAR (2 * BAND_BLOCKS)
#define PIPE_ROWS (64 * BAND_BLOCKS + 1)   // band[word][row], one row of padding: the resolver reads 64 consecutive rows of a
                                          // word (conflict-free), the fill writes 16 words of 4 rows per wave (banks 2j + 2i)
__global__ __launch_bounds__(1024) void k_nms_reduce_band(const u64 *__restrict__ mask, const int32_t *__restrict__ order,
                                                          const int32_t *__restrict__ counts, int64_t nmax, int64_t Np, int64_t W,
                                                          int quirk, int64_t *__restrict__ keep, int32_t *__restrict__ keep_count,
                                                          const int32_t *__restrict__ cflag, const NmsHdr *__restrict__ hdr)
{
    if (hdr && hdr[blockIdx.x].mode != 1) return;             // the pruned path does this image
    if (cflag && cflag[blockIdx.x]) return;                   // k_nms_reduce_csr does this image
    extern __shared__ u64 nms_lds[];
    u64 *removed = nms_lds, *removed1 = nms_lds + W, *band = nms_lds + 2 * W;      // band: [PIPE_NEAR][PIPE_ROWS]
    __shared__ int klist[2][64 * BAND_BLOCKS];
    __shared__ int s_order[64 * BAND_BLOCKS];                 // `order` of the super-block's rows: a global load in the resolver's
                                                              // path would stall its in-order wave for a memory round trip per block
    __shared__ int s_nk[2], s_total, s_last_ib, s_last_r;
    __shared__ u64 s_last_curb;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int n = nms_count(counts, img, nmax);
    mask += (int64_t)img * Np * W;
    order += (int64_t)img * Np;
    keep += (int64_t)img * nmax;
    const int nb = (n + 63) >> 6, nsb = (nb + BAND_BLOCKS - 1) / BAND_BLOCKS;
    auto init_word = [&](int w) -> u64 {
        const int64_t base = (int64_t)w * 64;
        return (base >= n) ? ~0ull : ((n - base >= 64) ? 0ull : (~0ull << (n - base)));
    };
    for (int w = tid; w < (int)W; w += 1024) { removed[w] = init_word(w); removed1[w] = init_word(w); }
    auto band_fetch = [&](int S, u64 (&r)[8]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = tid + 1024 * k, i = e >> 4, j = e & 15;
            const int row = S * 64 * BAND_BLOCKS + i, w = S * BAND_BLOCKS + j;
            r[k] = (S < nsb && row < nb * 64 && w < nb && w >= (row >> 6)) ? mask[(int64_t)row * W + w] : 0ull;
        }
    };
    auto order_fetch = [&](int S) -> int {
        const int row = S * 64 * BAND_BLOCKS + tid;
        return (tid < 64 * BAND_BLOCKS && row < nb * 64) ? order[row] : 0;
    };
    u64 bn[8];
    int on = order_fetch(0);
    if (tid < 64 * BAND_BLOCKS) s_order[tid] = on;
    band_fetch(0, bn);
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int e = tid + 1024 * k; band[(e & 15) * PIPE_ROWS + (e >> 4)] = bn[k]; }
    if (tid == 0) { s_total = 0; s_last_ib = -1; s_last_r = 0; s_last_curb = 0; s_nk[0] = 0; s_nk[1] = 0; }
    __syncthreads();
    int pend = -1;                                            // last kept row of the super-blocks the far side has seen
    for (int S = 0; S <= nsb; ++S) {
        band_fetch(S + 1, bn);                                // in flight while both sides work
        on = order_fetch(S + 1);
        if (tid < 64) {
            if (S < nsb) {
                int total = s_total, nk_sb = 0, last_ib = s_last_ib, last_r = s_last_r;
                u64 last_curb = s_last_curb;
                u64 dg[BAND_BLOCKS];                          // the eight diagonal words up front: nothing in the chain waits for them
#pragma unroll
                for (int bb = 0; bb < BAND_BLOCKS; ++bb) dg[bb] = band[bb * PIPE_ROWS + bb * 64 + lane];
#pragma unroll
                for (int bb = 0; bb < BAND_BLOCKS; ++bb) {
                    const int ib = S * BAND_BLOCKS + bb;
                    if (ib < nb) {                            // uniform
                        u64 cur = uniform64(removed[ib]);
                        if (cur != ~0ull) {
                            const u64 diag = dg[bb];
                            u64 curb = cur;
                            int lastact = -1;
                            u64 act = ~cur & uniform64(__ballot(diag != 0ull));
                            while (act) {
                                const int r = __builtin_ctzll(act);
                                curb = cur; lastact = r;
                                cur |= readlane64(diag, r);
                                act &= act - 1;
                                act &= ~cur;
                            }
                            const u64 kept = ~cur;
                            const int nk = __builtin_popcountll(kept);
                            if (nk) {
                                const bool mine = (kept >> lane) & 1ull;
                                const int rank = __builtin_popcountll(kept & ((1ull << lane) - 1ull));
                                if (mine) {
                                    keep[total + rank] = (int64_t)s_order[bb * 64 + lane];
                                    klist[S & 1][nk_sb + rank] = ib * 64 + lane;
                                }
                                // near words of the kept rows into `removed`.  Several lanes OR-ing into ONE LDS word serialise
                                // (a ds_or_b64 per word with every kept lane active cost more than the rest of the block), so
                                // with few kept rows the roles turn: lane = word column, a loop over the kept rows, one atomic
                                // instruction whose lanes all hit different words
                                if (nk <= 12) {
                                    u64 acc = 0, km = kept;
                                    const bool wlane = lane > bb && lane < PIPE_NEAR;
                                    while (km) {                 // four independent reads per pass (a repeated row ORs the same bits)
                                        const int r0 = __builtin_ctzll(km); km &= km - 1;
                                        const int r1 = km ? __builtin_ctzll(km) : r0; km &= km - 1;
                                        const int r2 = km ? __builtin_ctzll(km) : r0; km &= km - 1;
                                        const int r3 = km ? __builtin_ctzll(km) : r0; km &= km - 1;
                                        if (wlane) {
                                            const u64 *bw = band + lane * PIPE_ROWS + bb * 64;
                                            const u64 v0 = bw[r0], v1 = bw[r1], v2 = bw[r2], v3 = bw[r3];
                                            acc |= (v0 | v1) | (v2 | v3);
                                        }
                                    }
                                    if (wlane && acc) atomicOr((unsigned long long *)&removed[S * BAND_BLOCKS + lane], (unsigned long long)acc);
                                } else if (mine) {
#pragma unroll
                                    for (int j = bb + 1; j < PIPE_NEAR; ++j) {
                                        const u64 wj = band[j * PIPE_ROWS + bb * 64 + lane];
                                        if (wj) atomicOr((unsigned long long *)&removed[S * BAND_BLOCKS + j], (unsigned long long)wj);
                                    }
                                }
                                const int rl = 63 - __builtin_clzll(kept);
                                last_ib = ib; last_r = rl; last_curb = (rl == lastact) ? curb : cur;
                                total += nk; nk_sb += nk;
                            }
                        }
                    }
                }
                if (tid == 0) { s_nk[S & 1] = nk_sb; s_total = total; s_last_ib = last_ib; s_last_r = last_r; s_last_curb = last_curb; }
            }
        } else if (S >= 1) {
            // what super-block S-1 kept: its far words (its band reached word (S+1)*8 - 1) go into `removed`; every word
            // right of a row's own block also goes into `removed1`, except for the super-block's last kept row, which
            // stays pending until a later super-block keeps something
            const int nk_prev = s_nk[(S - 1) & 1];
            const int *kl = klist[(S - 1) & 1];
            if (nk_prev) {
                const int far0 = (S + 1) * BAND_BLOCKS;
                for (int w = (S - 1) * BAND_BLOCKS + (tid - 64); w < nb; w += 1024 - 64) {
                    u64 v1 = 0;
                    int k = 0;
                    for (; k + 16 <= nk_prev - 1; k += 16) {   // 16 independent loads in flight
                        u64 v[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) { const int row = kl[k + u]; v[u] = (w > (row >> 6)) ? mask[(int64_t)row * W + w] : 0ull; }
#pragma unroll
                        for (int u = 0; u < 16; ++u) v1 |= v[u];
                    }
                    for (; k + 4 <= nk_prev - 1; k += 4) {
                        u64 v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int row = kl[k + u]; v[u] = (w > (row >> 6)) ? mask[(int64_t)row * W + w] : 0ull; }
                        v1 |= (v[0] | v[1]) | (v[2] | v[3]);
                    }
                    for (; k < nk_prev - 1; ++k) { const int row = kl[k]; if (w > (row >> 6)) v1 |= mask[(int64_t)row * W + w]; }
                    const int lrow = kl[nk_prev - 1];
                    const u64 vl = (w > (lrow >> 6)) ? mask[(int64_t)lrow * W + w] : 0ull;
                    const u64 vp = (pend >= 0) ? mask[(int64_t)pend * W + w] : 0ull;       // w >= (S-1)*8 > pend's block
                    if (w >= far0 && (v1 | vl)) atomicOr((unsigned long long *)&removed[w], (unsigned long long)(v1 | vl));
                    removed1[w] |= v1 | vp;
                }
            }
        }
        if (S >= 1 && s_nk[(S - 1) & 1]) pend = klist[(S - 1) & 1][s_nk[(S - 1) & 1] - 1];
        __syncthreads();                                       // resolve(S) and far(S-1) done: the band is free
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int e = tid + 1024 * k; band[(e & 15) * PIPE_ROWS + (e >> 4)] = bn[k]; }
        if (tid < 64 * BAND_BLOCKS) s_order[tid] = on;
        __syncthreads();
    }
    int total = s_total;
    if (quirk && total > 0) {
        // yolo.py:196 `while ids_sorted.numel()>1`: the last kept box is dropped iff nothing ranked below it was still alive
        // before its own row was applied (removed1 = every kept row but that one)
        const int last_ib = s_last_ib, last_r = s_last_r;
        const u64 above = (last_r == 63) ? 0ull : (~0ull << (last_r + 1));
        int surv = (tid == 0) ? (((~s_last_curb) & above) != 0ull) : 0;
        for (int w = last_ib + 1 + tid; w < (int)W; w += 1024) surv |= (~removed1[w]) != 0ull;
        surv = __syncthreads_or(surv);
        if (!surv) total -= 1;
    }
    if (tid == 0) keep_count[img] = total;
}

// ---------------------------------------------------------------- compact mask + LDS-fed reduce (64 < W <= 1024)
// A reduce that walks the mask itself pays one HBM round trip per 64-row block (about 2 us): the words it needs are only known once the
// block is resolved.  Here the non-zero words right of the diagonal are first gathered, per block, into one contiguous
// run of (row-in-block << 16 | word, bits) entries -- a few MB in all -- which does not depend on what is kept.  The
// reduce fetches each block's run, diagonal words and order 8 blocks ahead into registers (one entry per thread) and
// publishes them into an LDS double buffer one block ahead, so the serial chain touches LDS only.  A set too dense for the entry
// capacity (32 per row on average) keeps the summary-driven kernel: cflag says which of the two runs.
#define CSR_LDS_ENTRIES 1024     // entries of a block's run the reduce holds in LDS (one per thread); the rest is read from memory
#define CSR_AHEAD 8              // blocks fetched ahead into registers
#define CSR_SLACK 8              // readable entries past the last run
__device__ __forceinline__ u64 csr_live_bits(u64 sumword, int s, int ib)   // summary bits of words > ib
{
    const int d = ib - s * 64;
    if (d < 0) return sumword;
    if (d >= 63) return 0ull;
    return sumword & ~((2ull << d) - 1ull);
}

// one wave = the 64 rows of one block: bcnt[block] = number of entries
__global__ __launch_bounds__(256) void k_nms_csr_count(const u64 *__restrict__ rsum, const int32_t *__restrict__ counts, int64_t nmax,
                                                       int64_t Np, int64_t W, int64_t SW, int32_t *__restrict__ bcnt,
                                                       const NmsHdr *__restrict__ hdr)
{
    const int img = blockIdx.y;
    if (hdr && hdr[img].mode != 1) return;
    const int n = nms_count(counts, img, nmax);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // row; Np is a multiple of 64
    if (i >= Np) return;
    const int ib = (int)(i >> 6);
    const int nb = (n + 63) >> 6, SWn = (nb + 63) >> 6;
    int c = 0;
    if (i < n)
        for (int s = ib >> 6; s < SWn; ++s) c += __builtin_popcountll(csr_live_bits(rsum[((int64_t)img * Np + i) * SW + s], s, ib));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) bcnt[(int64_t)img * W + ib] = c;
}

// one workgroup per image: exclusive scan of the block counts; cflag = the entries fit
__global__ __launch_bounds__(1024) void k_nms_csr_scan(const int32_t *__restrict__ bcnt, const int32_t *__restrict__ counts, int64_t nmax,
                                                       int64_t W, int64_t ecap, int32_t *__restrict__ boff, int32_t *__restrict__ cursor,
                                                       int32_t *__restrict__ cflag, const NmsHdr *__restrict__ hdr)
{
    __shared__ int64_t sc[1024];
    const int img = blockIdx.x, tid = threadIdx.x;
    if (hdr && hdr[img].mode != 1) return;
    const int n = nms_count(counts, img, nmax);
    const int nb = (n + 63) >> 6;                                    // <= 1024 here
    const int64_t v = (tid < nb) ? bcnt[(int64_t)img * W + tid] : 0;
    sc[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int64_t a = (tid >= o) ? sc[tid - o] : 0;
        __syncthreads();
        sc[tid] += a;
        __syncthreads();
    }
    const int64_t total = sc[1023];
    const bool fits = total <= ecap;
    if (tid < nb) {
        const int32_t o = fits ? (int32_t)(sc[tid] - v) : 0;
        boff[(int64_t)img * (W + 1) + tid] = o;
        cursor[(int64_t)img * W + tid] = o;
    }
    if (tid == 0) {
        boff[(int64_t)img * (W + 1) + nb] = fits ? (int32_t)total : 0;
        cflag[img] = fits ? 1 : 0;                                   // 0: too dense for the compact form, k_nms_reduce_band
    }
}

// thread = (row, summary word): copies its non-zero words right of the diagonal into the block's run
__global__ __launch_bounds__(256) void k_nms_csr_fill(const u64 *__restrict__ mask, const u64 *__restrict__ rsum,
                                                      const int32_t *__restrict__ counts, int64_t nmax, int64_t Np, int64_t W,
                                                      int64_t SW, int64_t ecap, const int32_t *__restrict__ cflag,
                                                      int32_t *__restrict__ cursor, u32 *__restrict__ ekey, u64 *__restrict__ eval,
                                                      const NmsHdr *__restrict__ hdr)
{
    const int img = blockIdx.y;
    if (hdr && hdr[img].mode != 1) return;
    if (cflag[img] != 1) return;
    const int n = nms_count(counts, img, nmax);
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = t / SW;
    const int s = (int)(t - i * SW);
    if (i >= n) return;
    const int ib = (int)(i >> 6), r = (int)(i & 63);
    const int SWn = (((n + 63) >> 6) + 63) >> 6;                 // summary words the mask phase wrote for this image
    if (s < (ib >> 6) || s >= SWn) return;
    u64 bits = csr_live_bits(rsum[((int64_t)img * Np + i) * SW + s], s, ib);
    const int c = __builtin_popcountll(bits);
    if (!c) return;
    int64_t pos = (int64_t)img * (ecap + CSR_SLACK) + atomicAdd(&cursor[(int64_t)img * W + ib], c);
    // (the 64 threads of a wave are 64/SW consecutive rows: when SW divides 64 they share one block counter, but the
    //  per-thread atomic is kept simple; k_nms_csr_fill_wave below is the aggregated form used when that holds)
    const u64 *mrow = mask + ((int64_t)img * Np + i) * W;
    while (bits) {
        const int w = s * 64 + __builtin_ctzll(bits);
        bits &= bits - 1;
        ekey[pos] = ((u32)r << 16) | (u32)w;
        eval[pos] = mrow[w];
        ++pos;
    }
}

// same, for SW a power of two <= 64: a wave covers 64/SW whole rows of ONE block, so it reserves its entries with a
// single atomic (lane prefix sums by shuffles) instead of one atomic per thread on the same counter
__global__ __launch_bounds__(256) void k_nms_csr_fill_wave(const u64 *__restrict__ mask, const u64 *__restrict__ rsum,
                                                           const int32_t *__restrict__ counts, int64_t nmax, int64_t Np, int64_t W,
                                                           int64_t SW, int sw_shift, int64_t ecap, const int32_t *__restrict__ cflag,
                                                           int32_t *__restrict__ cursor, u32 *__restrict__ ekey, u64 *__restrict__ eval,
                                                           const NmsHdr *__restrict__ hdr)
{
    const int img = blockIdx.y;
    if (hdr && hdr[img].mode != 1) return;
    if (!cflag[img]) return;
    const int n = nms_count(counts, img, nmax);
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = t >> sw_shift;
    const int s = (int)(t & (SW - 1));
    const int64_t i0 = (t - lane) >> sw_shift;                   // first row of this wave (uniform)
    if (i0 >= n) return;
    const int ib = (int)(i0 >> 6), r = (int)(i & 63);
    u64 bits = 0;
    const int SWn = (((n + 63) >> 6) + 63) >> 6;                 // summary words the mask phase wrote for this image
    if (i < n && s >= (ib >> 6) && s < SWn) bits = csr_live_bits(rsum[((int64_t)img * Np + i) * SW + s], s, ib);
    const int c = __builtin_popcountll(bits);
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    const int wave_total = __shfl(incl, 63, 64);
    if (wave_total == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(&cursor[(int64_t)img * W + ib], wave_total);
    base = __shfl(base, 0, 64);
    int64_t pos = (int64_t)img * (ecap + CSR_SLACK) + base + (incl - c);
    const u64 *mrow = mask + ((int64_t)img * Np + i) * W;
    while (bits) {
        const int w = s * 64 + __builtin_ctzll(bits);
        bits &= bits - 1;
        ekey[pos] = ((u32)r << 16) | (u32)w;
        eval[pos] = mrow[w];
        ++pos;
    }
}

// count + scan + fill in ONE launch for images of at most 64 blocks (W <= 64: one summary word per row, at most four rows per
// thread): a wave's 64 consecutive rows are one block, so block counts need no atomics and a block's run is filled in row
// order from the wave's prefix sums.  One 1024-thread workgroup per image (three launches of a few microseconds each
// otherwise: a third of a single 2000-box image's NMS); with few images gridDim.y workgroups share an image's blocks for the
// fill (each counts and scans for itself: 16 KB of summaries), since one workgroup alone fills a dense image's run slowly.
__global__ __launch_bounds__(1024) void k_nms_csr_build_small(const u64 *__restrict__ mask, const u64 *__restrict__ rsum,
                                                              const int32_t *__restrict__ counts, int64_t nmax, int64_t Np, int64_t W,
                                                              int64_t ecap_fit, int64_t ecap, int32_t *__restrict__ boff,
                                                              int32_t *__restrict__ cflag, u32 *__restrict__ ekey, u64 *__restrict__ eval)
{
    __shared__ int s_cnt[64], s_off[65];
    __shared__ int s_fits;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int n = nms_count(counts, img, nmax);
    const int nb = (n + 63) >> 6;                                // <= 64
    u64 bits[4];
    int cnt[4], incl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = tid + 1024 * k, ib = row >> 6;
        bits[k] = (row < n) ? csr_live_bits(rsum[((int64_t)img * Np + row) * 1], 0, ib) : 0ull;
        cnt[k] = __builtin_popcountll(bits[k]);
        int in = cnt[k];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(in, d, 64);
            if (lane >= d) in += up;
        }
        incl[k] = in;
        if (lane == 63 && ib < 64) s_cnt[ib] = in;               // every block is visited by exactly one wave
    }
    __syncthreads();
    if (tid < 64) {
        const int v = (tid < nb) ? s_cnt[tid] : 0;
        int in = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(in, d, 64);
            if (lane >= d) in += up;
        }
        const int total = __shfl(in, 63, 64);
        const bool fits = (int64_t)total <= ecap_fit;
        if (tid < nb) s_off[tid] = in - v;
        if (tid == 0) s_fits = fits;
        if (blockIdx.y == 0) {                                   // the parts count and scan alike; part 0 publishes
            if (tid < nb) boff[(int64_t)img * (W + 1) + tid] = fits ? in - v : 0;
            if (tid == 0) { boff[(int64_t)img * (W + 1) + nb] = fits ? total : 0; cflag[img] = fits ? 1 : 0; }
        }
    }
    __syncthreads();
    if (!s_fits) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = tid + 1024 * k, ib = row >> 6;
        if (row >= n || !bits[k] || (ib % (int)gridDim.y) != (int)blockIdx.y) continue;     // this part's blocks
        int64_t pos = (int64_t)img * (ecap + CSR_SLACK) + s_off[ib] + (incl[k] - cnt[k]);
        const u64 *mrow = mask + ((int64_t)img * Np + row) * W;
        u64 b = bits[k];
        while (b) {
            const int w = __builtin_ctzll(b);
            b &= b - 1;
            ekey[pos] = ((u32)(row & 63) << 16) | (u32)w;
            eval[pos] = mrow[w];
            ++pos;
        }
    }
}

// SEG (the pruned path's fallback, when k_nms_resolve_mc gave up): a prologue first groups k_nms_band's flat pair list
// (`ekey` = (lower rank << 16 | higher rank) entries, hdr.npairs of them) into per-block runs -- keys into the image's unused
// dense-mask memory, values into `eval` -- and builds the diagonal words in diag[]; entries of a block's own word column
// are skipped by the chain (the diagonal words carry them).
template <bool SEG>
__global__ __launch_bounds__(1024) void k_nms_reduce_csr(const u64 *__restrict__ mask, const int32_t *__restrict__ order,
                                                         const int32_t *__restrict__ counts, int64_t nmax, int64_t Np, int64_t W,
                                                         int64_t ecap, const int32_t *__restrict__ cflag, const int32_t *__restrict__ boff,
                                                         const u32 *__restrict__ ekey, const u64 *__restrict__ eval, int quirk,
                                                         int64_t *__restrict__ keep, int32_t *__restrict__ keep_count,
                                                         const NmsHdr *__restrict__ hdr, const u64 *__restrict__ diag,
                                                         const int32_t *__restrict__ lcnt, int64_t lcap)
{
    __shared__ u64 remv[1024];
    __shared__ __attribute__((aligned(16))) u64 s_val[2][CSR_LDS_ENTRIES];
    __shared__ __attribute__((aligned(16))) u32 s_key[2][CSR_LDS_ENTRIES];
    __shared__ u64 s_diag[2][64];
    __shared__ int32_t s_order[2][64];
    __shared__ int s_len[2], s_new[2];
    __shared__ int32_t s_boff[1025];                          // block offsets: read every iteration, keep them out of the memory path
    __shared__ int32_t s_blen[1024];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (SEG) {
        if (!(hdr[img].mode == 0 && hdr[img].fpdone == 0)) return;   // only what the fixed-point kernel left undone
    } else {
        if (hdr && hdr[img].mode != 1) return;                // the pruned path does this image
        if (cflag[img] != 1) return;                          // too dense: k_nms_reduce_band does this image
    }
    const int n = nms_count(counts, img, nmax);
    mask += (int64_t)img * Np * W;
    if (SEG) diag += (int64_t)img * Np;
    order += (int64_t)img * Np;
    keep += (int64_t)img * nmax;
    if (!SEG) boff += (int64_t)img * (W + 1);
    ekey += (int64_t)img * (ecap + CSR_SLACK);
    eval += (int64_t)img * (ecap + CSR_SLACK);
    const int nb = (n + 63) >> 6;
    for (int w = tid; w < (int)W; w += 1024) {
        const int64_t base = (int64_t)w * 64;
        remv[w] = (base >= n) ? ~0ull : ((n - base >= 64) ? 0ull : (~0ull << (n - base)));
    }
    if (SEG) {
        // group the flat pair list by the lower rank's block: count, scan, scatter (one workgroup; this is the rare path)
        __shared__ int32_t s_cur[1024];
        __shared__ int s_wsum[16];
        u32 *rkey = const_cast<u32 *>(ekey);                  // (at most ecap pairs come here: k_nms_resolve_mc made sure)
        u64 *rval = const_cast<u64 *>(eval), *dg = const_cast<u64 *>(diag);
        const u32 *lists = (const u32 *)mask;                 // the image's 64 pair lists (k_nms_band)
        lcnt += img * 64 * LC_STRIDE;
        s_blen[tid] = 0;
        for (int i = tid; i < nb * 64; i += 1024) dg[i] = 0ull;
        __syncthreads();
        for (int g = 0; g < 64; ++g) {
            const int E = lcnt[g * LC_STRIDE];
            const u32 *lg = lists + (int64_t)g * lcap;
            for (int e = tid; e < E; e += 1024) atomicAdd(&s_blen[lg[e] >> 22], 1);
        }
        __syncthreads();
        {
            const int v = s_blen[tid];
            int incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
            if (lane == 63) s_wsum[wave] = incl;
            __syncthreads();
            int base = 0;
            for (int k = 0; k < wave; ++k) base += s_wsum[k];
            s_boff[tid] = base + incl - v;
            s_cur[tid] = base + incl - v;
        }
        __syncthreads();
        for (int g = 0; g < 64; ++g) {
            const int E = lcnt[g * LC_STRIDE];
            const u32 *lg = lists + (int64_t)g * lcap;
            for (int e = tid; e < E; e += 1024) {
                const u32 key = lg[e], lo = key >> 16, hi = key & 0xffffu;
                const int at = atomicAdd(&s_cur[lo >> 6], 1);
                rkey[at] = ((lo & 63u) << 16) | (hi >> 6);
                rval[at] = 1ull << (hi & 63u);
                if ((hi >> 6) == (lo >> 6)) atomicOr((unsigned long long *)&dg[lo], 1ull << (hi & 63u));
            }
        }
        __syncthreads();                                      // the runs were written by this workgroup: visible to it from here on
    } else {
        for (int w = tid; w < nb; w += 1024) { const int o = boff[w]; s_boff[w] = o; s_blen[w] = boff[w + 1] - o; }
    }
    if (tid < 2) s_new[tid] = 0;
    __syncthreads();

    // Every thread carries ONE entry of each of the next CSR_AHEAD blocks in registers (wave 0 also their diagonal words
    // and order): register slot b % CSR_AHEAD holds block b from CSR_AHEAD iterations before it is needed until it is
    // published into the LDS double buffer, one iteration before.  The block loop is unrolled by CSR_AHEAD so that the
    // slots are plain registers.  Only the lanes that have an entry load one (a run is a few hundred entries).
    u32 hkey[CSR_AHEAD];
    u64 hval[CSR_AHEAD], hdiag[CSR_AHEAD];
    int32_t hord[CSR_AHEAD];
    int hlen[CSR_AHEAD];
#pragma unroll
    for (int j = 0; j < CSR_AHEAD; ++j) { hkey[j] = 0; hval[j] = 0; hdiag[j] = 0; hord[j] = 0; hlen[j] = 0; }
    auto fetch = [&](int j, int b) {
        if (b >= nb) return;
        const int o = s_boff[b], len = s_blen[b];
        hlen[j] = len;
        hkey[j] = (tid < len) ? ekey[o + tid] : 0u;
        hval[j] = (tid < len) ? eval[o + tid] : 0ull;
        if (wave == 0) { hdiag[j] = SEG ? diag[b * 64 + lane] : mask[((int64_t)b * 64 + lane) * W + b]; hord[j] = order[b * 64 + lane]; }
    };
    auto publish = [&](int j, int slot) {
        s_key[slot][tid] = hkey[j];
        s_val[slot][tid] = hval[j];
        if (wave == 0) { s_diag[slot][lane] = hdiag[j]; s_order[slot][lane] = hord[j]; }
        if (tid == 0) s_len[slot] = hlen[j];
    };
#pragma unroll
    for (int j = 0; j < CSR_AHEAD; ++j) fetch(j, j);
    if (nb > 0) { publish(0, 0); fetch(0, CSR_AHEAD); }
    __syncthreads();

    int total = 0, last_r = 0, last_new = 0, any_kept = 0;
    u64 last_curb = 0;
    for (int ib0 = 0; ib0 < nb; ib0 += CSR_AHEAD) {
#pragma unroll
      for (int jj = 0; jj < CSR_AHEAD; ++jj) {
        const int ib = ib0 + jj;
        if (ib >= nb) break;
        const int slot = ib & 1;
        u64 cur = uniform64(remv[ib]);
        const u64 diag = s_diag[slot][lane];
        u64 curb = cur;
        int lastact = -1;
        if (cur != ~0ull) {
            u64 m = uniform64(__ballot(diag != 0ull));
            while (m) {
                const int r = __builtin_ctzll(m);
                m &= m - 1;
                if (!((cur >> r) & 1ull)) { curb = cur; lastact = r; cur |= readlane64(diag, r); }
            }
        }
        const u64 kept = ~cur;
        const int nk = __builtin_popcountll(kept);
        const int rl = nk ? 63 - __builtin_clzll(kept) : 0;
        const int len = s_len[slot];
        if (tid == 0) s_new[slot] = 0;
        if (nk) {
            if (wave == 0 && ((kept >> lane) & 1ull))
                keep[total + __builtin_popcountll(kept & ((1ull << lane) - 1ull))] = (int64_t)s_order[slot][lane];
            // pass 1: every kept row but the last one
            const u64 others = kept & ~(1ull << rl);
            if (tid < len) {
                const u32 k = s_key[slot][tid];
                if (((others >> (k >> 16)) & 1ull) && (!SEG || (int)(k & 0xffffu) != ib))
                    atomicOr((unsigned long long *)&remv[k & 0xffffu], (unsigned long long)s_val[slot][tid]);
            }
            if (len > CSR_LDS_ENTRIES) {                    // a long run: its tail straight from memory
                const int o = s_boff[ib];
                for (int e = CSR_LDS_ENTRIES + tid; e < len; e += 1024) {
                    const u32 k = ekey[o + e];
                    if (((others >> (k >> 16)) & 1ull) && (!SEG || (int)(k & 0xffffu) != ib))
                        atomicOr((unsigned long long *)&remv[k & 0xffffu], (unsigned long long)eval[o + e]);
                }
            }
        }
        __syncthreads();
        if (nk) {
            // pass 2: the last kept row, counting the bits it is the first to set (the numel()>1 quirk needs that)
            int fresh = 0;
            if (tid < len) {
                const u32 k = s_key[slot][tid];
                if ((int)(k >> 16) == rl && (!SEG || (int)(k & 0xffffu) != ib)) {
                    const u64 v = s_val[slot][tid];
                    const u64 old = atomicOr((unsigned long long *)&remv[k & 0xffffu], (unsigned long long)v);
                    fresh = __builtin_popcountll(v & ~old);
                }
            }
            if (len > CSR_LDS_ENTRIES) {
                const int o = s_boff[ib];
                for (int e = CSR_LDS_ENTRIES + tid; e < len; e += 1024) {
                    const u32 k = ekey[o + e];
                    if ((int)(k >> 16) == rl && (!SEG || (int)(k & 0xffffu) != ib)) {
                        const u64 v = eval[o + e];
                        const u64 old = atomicOr((unsigned long long *)&remv[k & 0xffffu], (unsigned long long)v);
                        fresh += __builtin_popcountll(v & ~old);
                    }
                }
            }
            if (__ballot(fresh != 0)) {
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) fresh += __shfl_xor(fresh, d, 64);
                if (lane == 0) atomicAdd(&s_new[slot], fresh);
            }
        }
        publish((jj + 1) % CSR_AHEAD, (ib + 1) & 1);            // block ib+1; (jj+1) % AHEAD == (ib+1) % AHEAD: ib0 is a multiple
        fetch((jj + 1) % CSR_AHEAD, ib + 1 + CSR_AHEAD);
        __syncthreads();
        if (nk) {
            any_kept = 1; last_r = rl; last_curb = (rl == lastact) ? curb : cur;
            last_new = s_new[slot];
            total += nk;
        }
      }
    }
    if (quirk && any_kept) {
        // yolo.py:196 `while ids_sorted.numel()>1`: the last kept box is dropped iff nothing ranked below it was still
        // alive before its own row was applied: no such bit in its own block, none set first by its row further right
        const u64 above = (last_r == 63) ? 0ull : (~0ull << (last_r + 1));
        if ((((~last_curb) & above) == 0ull) && last_new == 0) total -= 1;
    }
    if (tid == 0) keep_count[img] = total;
}

// ---------------------------------------------------------------- pruned path (64 < W <= 1024): no N x N mask
// A pair whose boxes do not overlap has IoU 0 < thr (thr > 0, finite areas), so only overlapping pairs can suppress, and a set
// of small boxes has few of them.  The score-sorted boxes are counting-sorted by the x-bucket of their left edge (XB_G buckets
// over [min x1, max x1]: the bucket function is monotone, ties inside a bucket in any order).  In that order box p can only
// overlap a later box q if x1_q < x2_p, i.e. q lies before the end of bucket(x2_p): the pair search is the mask kernel's
// 64-row x 64-column pass over a BAND of column tiles instead of the whole triangle (config 5's sparse set: 31 tiles per
// row tile instead of 513 on average).  Every suppressing pair is found exactly once (position order is a total order) and
// emitted as a compact entry (lower rank's row in its 64-row block << 16 | word of the higher rank, one bit) into the lower
// rank's block segment -- the form k_nms_reduce_csr consumes -- so the 512 MiB mask is neither written nor read.
//   xbucket_image : one workgroup per image, launched as an extra workgroup of k_nms_sort_chunks (it needs no ranks): min/max of
//                   x1, non-finite boxes (-> dense path), histogram, scan, each box's position by the caller's index; the
//                   estimated number of candidate pairs decides between this path and the dense mask (hdr.mode);
//   k_nms_merge_rank: besides the score-sorted copy, the boxes into x-bucket order (bbox = corners, bar = {area, rank});
//   k_nms_band    : the pair search;
//   k_nms_resolve_mc: the greedy keep list as a fixed point instead of a serial chain (below).
#define BAND_STAGE 256
#ifndef BAND_ROWS
#define BAND_ROWS 8      // rows of a tile per pass of k_nms_band's row loop
#endif
// (XB_G, xb_bucket and the x-bucket order itself, xbucket_image: above, beside the chunk sort it runs with)
// the mask phase on its own (dg_nms_phase_mask called again on a sorted workspace, as bench.py times it): the pair list starts empty
__global__ void k_nms_reset_pairs(NmsHdr *__restrict__ hdr, int32_t *__restrict__ lcnt, int batch)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < batch * 64) lcnt[i * LC_STRIDE] = 0;
    if (i < batch) { hdr[i].npairs = 0; hdr[i].estsum = 0u; }
}

// How many suppressing pairs will the band search find?  Known only once the threshold is: 1024 boxes spread over the x-bucket
// order, each against 64 of its candidates (all of them if it has no more), scaled up.  A dense set at a low threshold has
// hundreds of pairs per box -- writing and resolving them costs more than the dense mask does (27 M pairs for the dense
// 64 k set at 0.3: 5 ms against 1.9) -- so above NMS_EST_MAX / 1024 pairs per box k_nms_band leaves the image to the dense path.
#define NMS_EST_MAX (128u * 1024u)
__global__ __launch_bounds__(256) void k_nms_estimate(const float4 *__restrict__ bbox, const float2 *__restrict__ bar,
                                                      NmsHdr *__restrict__ hdr, const int32_t *__restrict__ counts, int64_t nmax,
                                                      int64_t Np, float thr)
{
    const int img = blockIdx.y;
    NmsHdr *h = hdr + img;
    if (h->mode != 0) return;
    const int n = nms_count(counts, img, nmax);
    const int64_t o = (int64_t)img * Np;
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);      // 1024 waves per image, one sample each
    u32 sum = 0;
    {
        const int p = (int)(((long long)gw * n) >> 10);
        const float4 b = bbox[o + p];
        const float2 ar = bar[o + p];
        int e = (int)(((u32)__float_as_int(ar.y) >> 16) + 1) * 64;
        e = e < n ? e : n;
        const int cand = e - (p + 1);
        if (cand <= 0) return;                                     // wave-uniform
        const int q = p + 1 + (cand <= 64 ? lane : (int)(((long long)lane * cand) >> 6));
        bool hit = false;
        if (cand > 64 || lane < cand) {
            const float4 c = bbox[o + q];
            const float ca = bar[o + q].x;
            const float ix = fmaxf(fminf(b.z, c.z) - fmaxf(b.x, c.x), 0.0f), iy = fmaxf(fminf(b.w, c.w) - fmaxf(b.y, c.y), 0.0f);
            const float inter = ix * iy, den = ((ar.x + ca) - inter) + 1e-7f;
            hit = !(inter < thr * den);
        }
        const u32 c = (u32)__builtin_popcountll(__ballot(hit));
        sum += cand <= 64 ? c : (u32)(((unsigned long long)c * (unsigned)cand) >> 6);
    }
    if (lane == 0 && sum) atomicAdd(&h->estsum, sum);
}

__global__ void k_nms_force_dense(NmsHdr *__restrict__ hdr, int batch)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < batch) { hdr[i].mode = 1; hdr[i].fpdone = 0; }
}

// grid: x = part of the band (waves of all parts stride the column tiles together), y = row tile (64 boxes in x-bucket
// order), z = image.  The 64-row pass is k_nms_mask's (same division-free test, same exact redo); what differs is where
// the bits go.  (Measured and not kept: leaving a row's pass after `inter` when none of the 64 column boxes overlaps the row
// box -- most rows of a sparse set -- made the kernel slower, 108 -> 120 us at 64 k boxes: the branch splits the unrolled
// pass; two column tiles per pass: 112 -> 134 us.)
template <bool TIE_EVEN>
__global__ __launch_bounds__(256, 4) void k_nms_band(const float4 *__restrict__ bbox, const float2 *__restrict__ bar,
                                                     const int32_t *__restrict__ start, NmsHdr *__restrict__ hdr,
                                                     const int32_t *__restrict__ counts, int64_t nmax, int64_t Np,
                                                     int64_t lcap, int lsh, float thr, float thr_below, double mid,
                                                     u64 *__restrict__ maskws, int64_t W, int32_t *__restrict__ lcnt, int force,
                                                     u32 *__restrict__ mc)
{
    __shared__ float4 rbox[64];
    __shared__ float rarea[64];
    __shared__ int rrank[64];
    __shared__ int s_jrow[64], s_rpos[64], s_jtmp[64];             // per row slot: last column tile of its band, its place in the tile
    __shared__ int s_jend;
    __shared__ u32 s_stage[4][BAND_STAGE];                         // a wave's pairs wait here: one bump of the list cursors per flush
    __shared__ int s_ocnt[4][64], s_obase[4][64];
    const int img = blockIdx.z;
    NmsHdr *h = hdr + img;
    if (blockIdx.x == 0 && blockIdx.y == 0)                        // what k_nms_resolve_mc's workgroups tell each other starts at zero
        for (int i = threadIdx.x; i < 4352; i += 256) mc[(int64_t)img * 4352 + i] = 0u;     // MC_WORDS
    if (h->mode != 0) return;
    if (force != 2 && h->estsum > NMS_EST_MAX) {                   // too many pairs to list (k_nms_estimate): the dense path does this image
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *(volatile int32_t *)&h->mode = 1;
        return;
    }
    const int n = nms_count(counts, img, nmax);
    const int t = blockIdx.y;
    if (t * 64 >= n) return;
    const int64_t o = (int64_t)img * Np;
    u32 *lists = (u32 *)(maskws + (int64_t)img * Np * W);           // list g: lists + g * lcap
    lcnt += img * 64 * LC_STRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64) {
        const float4 b = bbox[o + t * 64 + tid];
        const float2 ar = bar[o + t * 64 + tid];
        int jend = (t * 64 + tid < n) ? (int)((u32)__float_as_int(ar.y) >> 16) : -1;  // last column tile that can hold a partner
        // With the threshold known the band is narrower than the box: a partner q further right (x1_q >= x1_p) meets p in
        // ix <= x2_p - x1_q, and IoU <= inter / max(A_p, A_q) <= (ix * h_p) / (w_p * h_p) = ix / w_p, so a pair over thr has
        // x1_q <= x1_p + (1 - thr) * w_p: (1 - thr) of the candidates (half of them at 0.5, 0.3 at 0.7).  Float side: the
        // reference's quotient exceeds ix / w_p by a few 2^-24 at most (every rounding in inter, the areas and the
        // denominator, whose + 1e-7 only lowers it) -- covered by the 2e-4 added to (1 - thr) and the limit rounded up;
        // boxes so thin that their area is not a normal number keep the full band (a box with w <= 0 or h <= 0 meets nobody:
        // inter = 0).  Partners inside p's own bucket may start left of it: the limit's bucket includes that bucket.
        if (t * 64 + tid < n) {
            const float w = b.z - b.x, hh = b.w - b.y;
            if (w > 1e-10f && hh > 1e-10f && w < 1e10f && hh < 1e10f) {
                float xl = __builtin_fmaf(w, (1.0f - thr) + 2e-4f, b.x);
                xl += fabsf(xl) * 0x1p-21f + 1e-37f;
                xl = fminf(xl, b.z);
                const int e2 = (start[(int64_t)img * (XB_G + 1) + xb_bucket(xl, h->xmin, h->scale) + 1] - 1) >> 6;
                jend = e2 < jend ? e2 : jend;
            }
        }
        // The rows of the tile in the order of their bands, longest first (their order inside the tile is free: a row carries
        // its rank and its place): column tile j then needs the first R_j row slots only, those whose band reaches it -- a
        // tile's bands differ by the boxes' widths, and running all 64 rows to the end of the widest cost 1.8 x the pairs.
        s_jtmp[tid] = jend;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (one wave wrote them, the same wave reads them all)
        int slot = 0;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) { const int jk = s_jtmp[k]; slot += (jk > jend || (jk == jend && k < tid)) ? 1 : 0; }
        rbox[slot] = b; rarea[slot] = ar.x; rrank[slot] = __float_as_int(ar.y) & 0xffff;
        s_jrow[slot] = jend; s_rpos[slot] = tid;
        int jmax = jend;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) jmax = max(jmax, __shfl_xor(jmax, d, 64));
        if (tid == 0) s_jend = jmax > t ? jmax : t;
    }
    __syncthreads();
    const int jend = s_jend;
    const int myjrow = s_jrow[lane], myrpos = s_rpos[lane];
    const float2 *cbar = bar + o;
    u32 *stage = s_stage[wave];
    int staged = 0;                                                // wave-uniform
    // the staged pairs go to the list of their higher rank's owner (64 lists: k_nms_resolve_mc gives each to one workgroup):
    // counted per list in LDS, one bump of each touched list's cursor, then every pair to its slot
    int *oc = s_ocnt[wave], *ob = s_obase[wave];
    auto flush = [&]() {
        if (staged == 0) return;
        oc[lane] = 0;
        u32 key[4];
        int loc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane + 64 * q;
            key[q] = 0u; loc[q] = 0;
            if (i < staged) { key[q] = stage[i]; loc[q] = atomicAdd(&oc[(key[q] & 0xffffu) >> lsh], 1); }
        }
        const int c = oc[lane];
        int b = 0;
        if (c) {
            b = atomicAdd(&lcnt[lane * LC_STRIDE], c);
            if ((int64_t)b + c > lcap) *(volatile int32_t *)&h->mode = 1;        // a list is full: the dense path redoes this image
        }
        ob[lane] = b;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane + 64 * q;
            if (i < staged) {
                const int own = (int)((key[q] & 0xffffu) >> lsh);
                const int64_t at = (int64_t)ob[own] + loc[q];
                if (at < lcap) lists[(int64_t)own * lcap + at] = key[q];
            }
        }
        staged = 0;
    };
#pragma unroll 1
    for (int j = t + (int)blockIdx.x * 4 + wave; j <= jend; j += 4 * (int)gridDim.x) {       // wave-uniform
        asm volatile("" ::: "memory");
        const float4 cb = bbox[o + (int64_t)j * 64 + lane];
        const float ca = cbar[(int64_t)j * 64 + lane].x;
        const float cx1 = cb.x, cy1 = cb.y, cx2 = cb.z, cy2 = cb.w;
        u32 wlo = 0, whi = 0;
        u64 ambany = 0;
        // row slots whose band reaches this column tile (the tile's own diagonal tile: all of them)
        const int R = (j == t) ? 64 : __builtin_popcountll(__ballot(myjrow >= j));
#pragma unroll 1
        for (int r0 = 0; r0 < R; r0 += BAND_ROWS) {
#pragma unroll
            for (int u = 0; u < BAND_ROWS; ++u) {
                const int r = r0 + u;                              // (slots past R in the last eight: tested too, harmless)
                const float4 p = rbox[r];
                const float ix = vmax(vmin(p.z, cx2) - vmax(p.x, cx1), 0.0f);
                const float iy = vmax(vmin(p.w, cy2) - vmax(p.y, cy1), 0.0f);
                const float inter = ix * iy;
                const float den = ((rarea[r] + ca) - inter) + 1e-7f;
                const float e1 = __builtin_fmaf(-thr, den, inter), e2 = __builtin_fmaf(-thr_below, den, inter);
                const u64 bal = __ballot(!(e1 < 0.0f));
                ambany |= __ballot(e2 > 0.0f) & ~bal;
                // (lane select through m0: an instruction takes one SGPR beside it; nothing else in this kernel uses m0, and the
                // compiler, which reserves it, only notes that the clobber names a reserved register)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
                asm("s_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %4, m0"
                    : "+v"(wlo), "+v"(whi) : "s"((u32)bal), "s"(r), "s"((u32)(bal >> 32)) : "m0");
#pragma clang diagnostic pop
            }
        }
        u64 myword = ((u64)whi << 32) | wlo;                       // lane = row slot, bit = column of this tile
        if (ambany) {                                             // rare: a quotient within one ulp of thr somewhere
            u64 w2 = 0;
#pragma unroll 2
            for (int r = 0; r < ((R + BAND_ROWS - 1) & ~(BAND_ROWS - 1)); ++r) {
                const float4 p = rbox[r];
                const float ix = fmaxf(fminf(p.z, cx2) - fmaxf(p.x, cx1), 0.0f);
                const float iy = fmaxf(fminf(p.w, cy2) - fmaxf(p.y, cy1), 0.0f);
                const float inter = ix * iy;
                const float den = ((rarea[r] + ca) - inter) + 1e-7f;
                const u64 bal = __ballot(nms_suppressed<TIE_EVEN>(inter, den, mid));
                if (lane == r) w2 = bal;
            }
            myword = w2;
        }
        if (j == t) myword &= ~((2ull << myrpos) - 1ull);          // later positions only: every pair once
        const int cnt = __builtin_popcountll(myword);
        if (__ballot(cnt != 0)) {                                  // wave-uniform
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
            const int total = __shfl(incl, 63, 64);
            const int rr = rrank[lane];
            auto key_of = [&](int c) {
                const int cr = __float_as_int(cbar[(int64_t)j * 64 + c].y) & 0xffff;
                const u32 lo = (u32)(rr < cr ? rr : cr), hi = (u32)(rr < cr ? cr : rr);
                return (lo << 16) | hi;                            // ranks < 65536 (W <= 1024)
            };
            if (total <= BAND_STAGE) {
                if (staged + total > BAND_STAGE) flush();
                int at = staged + incl - cnt;
                while (myword) {
                    const int c = __builtin_ctzll(myword);
                    myword &= myword - 1;
                    stage[at++] = key_of(c);
                }
                staged += total;
            } else {                                               // a tile of near-duplicates: four pairs per row at a time
                flush();
                while (__ballot(myword != 0ull)) {
                    const int pc = __builtin_popcountll(myword), c4 = pc < 4 ? pc : 4;
                    int in2 = c4;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(in2, d, 64); if (lane >= d) in2 += up; }
                    int at = in2 - c4;
                    for (int q = 0; q < c4; ++q) {
                        const int c = __builtin_ctzll(myword);
                        myword &= myword - 1;
                        stage[at++] = key_of(c);
                    }
                    staged = __shfl(in2, 63, 64);
                    flush();
                }
            }
        }
    }
    flush();
}

// The keep list as a FIXED POINT instead of a chain over 64-row blocks.  A box is kept iff no kept box of higher rank
// suppresses it; with the suppressing pairs as a list of entries (source rank, target rank) that unfolds in rounds:
//   an entry whose source is KEPT marks its target REMOVED and dies; one whose source is REMOVED dies; one whose source is
//   still undecided blocks its target for this round; after the pass every undecided box that nothing blocks is KEPT.
// The highest-ranked undecided box is decided in every round, so it ends; with scores independent of geometry the number of
// rounds grows like log^2 n (parallel greedy MIS), about ten for config 5.
// MC_G = 64 workgroups per image.  Workgroup g owns the targets g << lsh .. and their pair list (k_nms_band put every pair
// into its higher rank's list): the targets' KEPT / REMOVED / R2 / blocked bits live in ITS LDS and only it touches them; what
// a round needs from the others is the state of the SOURCES, a snapshot of all KEPT and REMOVED words that every workgroup
// publishes (its own words) before the round's grid barrier and reads (all words) after it.  A snapshot that is one round old
// is harmless: KEPT / REMOVED bits are final once set, and a source that still looks undecided only keeps its entry alive for
// another round.  One barrier per round (a round costs about 10 us: four dependent trips to memory): a 64-bit word per round
// that every workgroup adds (its live entries << 8 | 1) to and spins on until 64 have arrived -- bounded: a barrier nobody
// releases ends the kernel with keep_count = -1 instead of hanging it.  All 64 workgroups of an image are dispatched before any
// of the next image, so the ones that spin never wait for one that has not started.  Everything the workgroups tell each
// other goes through agent-scope atomic loads and stores (the XCDs' L2s are not coherent with each other for plain
// accesses); each publisher waits for its stores before the workgroup's arrival is counted.  (One workgroup with every bit
// set in its LDS and no barrier was measured too: 141 us for config 5's 103 k pairs against 92 us, 610 against 125 for
// 625 k pairs; only below about 10 k pairs was it the faster one, 28 against 47 us.)
// Thread t owns the 16-byte chunks t, t + 1024, ... of its workgroup's list (four entries each); the entries that stay live
// it writes back to the same chunks of the work list, in order (in place from the second pass on: the write index never
// passes the read index), so no cursor, no atomics on memory, coalesced accesses and 32 entries in flight per thread.  The
// sources' states of 16 entries are read together (one LDS round trip instead of 16).
// The reference's `numel()>1` quirk (yolo.py:196) drops the last kept box iff every lower-ranked box was already removed by
// OTHER kept boxes, i.e. iff no removed box has it as its ONLY kept suppressor: R2 = targets marked by at least two kept
// sources, single[t] = the first kept source that marked t; the box is dropped iff the highest single[] among the removed
// targets outside R2 is not the last kept box.
// After `max_rounds` it gives up and k_nms_reduce_csr<SEG> walks the pairs serially instead (a chain of n boxes each
// suppressing the next would need n rounds) -- unless the pairs are more than that kernel's lists hold: then the rounds
// simply go on (n of them at the very worst).
#define MC_G 64
#define MC_K 0            // 2048 words: kept bits
#define MC_R 2048         // 2048 words: removed bits
#define MC_SLOT 4096      // 3 x u64: round r's arrivals (low 8 bits) and live entries (the rest) in slot r % 3
#define MC_KCNT 4104      // 64: kept boxes of each workgroup's targets
#define MC_KMAX 4168      // 64: highest kept rank + 1
#define MC_SNGL 4232      // 64: highest sole suppressor + 1
#define MC_WORDS 4352
#define MC_SPIN_LIMIT (1u << 21)
__device__ __forceinline__ u32 mc_ld(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mc_st(u32 *p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 nms_valid32(int w, int n)
{
    const int vb = w * 32;
    return (vb >= n) ? 0u : ((n - vb >= 32) ? ~0u : ((1u << (n - vb)) - 1u));
}

struct NmsFpShared {
    __attribute__((aligned(16))) u32 KR[4096];   // the sources' states: KR[2w] = kept bits of boxes 32w .. 32w+31, KR[2w+1] = removed bits
    u32 oK[32], oR[32], oR2[32], oBk[32];         // the targets this workgroup owns (1 << lsh <= 1024 of them)
    unsigned short single[1024];                  // first kept suppressor of each owned target
    int off[32];
    int cnt[16], ok, sngl;
    unsigned long long live;
};

__global__ __launch_bounds__(1024) void k_nms_resolve_mc(NmsHdr *__restrict__ hdr, const int32_t *__restrict__ counts, int64_t nmax,
                                                         int64_t Np, int64_t W, int64_t lcap, int lsh, int64_t ecap,
                                                         const int32_t *__restrict__ lcnt, const int32_t *__restrict__ order,
                                                         u64 *__restrict__ maskws, u32 *__restrict__ mc, int max_rounds, int quirk,
                                                         int64_t *__restrict__ keep, int32_t *__restrict__ keep_count)
{
    __shared__ NmsFpShared S;
    const int img = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    NmsHdr *h = hdr + img;
    lcnt += img * 64 * LC_STRIDE;
    int Etot = lcnt[lane * LC_STRIDE];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) Etot += __shfl_xor(Etot, d, 64);
    if (h->mode != 0) {                                          // dense mask; if a pair list overflowed, say how many pairs there were
        if (g == 0 && tid == 0) h->npairs = Etot;
        return;
    }
    const int n = nms_count(counts, img, nmax);
    order += (int64_t)img * Np;
    keep += (int64_t)img * nmax;
    u32 *mcs = mc + (int64_t)img * MC_WORDS;
    const u32 *list = (const u32 *)(maskws + (int64_t)img * Np * W) + (int64_t)g * lcap;       // 64 pair lists, then the 64 work lists
    u32 *wlist = (u32 *)(maskws + (int64_t)img * Np * W) + (int64_t)(MC_G + g) * lcap;
    u32 *wmine = wlist + 4 * tid;
    const int E = lcnt[g * LC_STRIDE];                           // <= lcap (mode would be 1 otherwise)
    const int own0 = g << lsh, ow0 = own0 >> 5, nown = (1 << lsh) >> 5;
    if (tid < 32) { S.oK[tid] = 0u; S.oR[tid] = 0u; S.oR2[tid] = 0u; S.oBk[tid] = 0u; }
    __syncthreads();
    // round 0: boxes nothing suppresses are kept
    for (int64_t c0 = 0; c0 * 4 < E; c0 += 8 * 1024) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t c = c0 + u * 1024 + tid;
            v[u] = (c * 4 < E) ? *(const uint4 *)(list + 4 * c) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t e = (c0 + u * 1024 + tid) * 4;
            if (e < E) atomicOr(&S.oBk[((v[u].x & 0xffffu) >> 5) - ow0], 1u << (v[u].x & 31u));
            if (e + 1 < E) atomicOr(&S.oBk[((v[u].y & 0xffffu) >> 5) - ow0], 1u << (v[u].y & 31u));
            if (e + 2 < E) atomicOr(&S.oBk[((v[u].z & 0xffffu) >> 5) - ow0], 1u << (v[u].z & 31u));
            if (e + 3 < E) atomicOr(&S.oBk[((v[u].w & 0xffffu) >> 5) - ow0], 1u << (v[u].w & 31u));
        }
    }
    __syncthreads();
    if (tid < 32) { S.oK[tid] = (tid < nown ? nms_valid32(ow0 + tid, n) : 0u) & ~S.oBk[tid]; S.oBk[tid] = 0u; }
    const u32 *src_list = list;
    int mycnt = 0;
    for (int64_t c = tid; c * 4 < E; c += 1024) mycnt += (E - c * 4 >= 4) ? 4 : (int)(E - c * 4);
    int rounds = 0;
    bool bail = false, hung = false;
    for (;;) {
        __syncthreads();
        {
            int c = mycnt;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
            if (lane == 0) S.cnt[wave] = c;
            if (tid == 0) S.sngl = 0;
        }
        __syncthreads();
        // ---- publish: this workgroup's words, its kept count / highest kept rank / highest sole suppressor
        {
            int v = 0;                                             // one owned target per thread
            if (tid < (1 << lsh) && ((S.oR[tid >> 5] & ~S.oR2[tid >> 5]) >> (tid & 31)) & 1u) v = (int)S.single[tid] + 1;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
            if (lane == 0 && v) atomicMax(&S.sngl, v);
        }
        __syncthreads();
        if (tid < nown) { mc_st(&mcs[MC_K + ow0 + tid], S.oK[tid]); mc_st(&mcs[MC_R + ow0 + tid], S.oR[tid]); }
        if (wave == 1) {
            const u32 kw = lane < nown ? S.oK[lane] : 0u;
            int kc = __builtin_popcount(kw), km = kw ? (ow0 + lane) * 32 + 32 - __builtin_clz(kw) : 0;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { kc += __shfl_xor(kc, d, 64); km = max(km, __shfl_xor(km, d, 64)); }
            if (lane == 0) {
                mc_st(&mcs[MC_KCNT + g], (u32)kc); mc_st(&mcs[MC_KMAX + g], (u32)km); mc_st(&mcs[MC_SNGL + g], (u32)S.sngl);
                if (g == 0)                                        // next round's slot: last read two barriers ago
                    __hip_atomic_store((unsigned long long *)&mcs[MC_SLOT + 2 * ((rounds + 1) % 3)], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // what this thread published has arrived
        __syncthreads();
        // ---- grid barrier `rounds`
        if (tid == 0) {
            unsigned long long mine = 1ull;
#pragma unroll
            for (int k = 0; k < 16; ++k) mine += (unsigned long long)S.cnt[k] << 8;
            unsigned long long *slot = (unsigned long long *)&mcs[MC_SLOT + 2 * (rounds % 3)];
            atomicAdd(slot, mine);
            u32 spins = 0;
            int ok = 1;
            unsigned long long v;
            while (((v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffull) < (unsigned long long)MC_G) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > MC_SPIN_LIMIT) { ok = 0; break; }
            }
            S.ok = ok; S.live = v >> 8;
        }
        __syncthreads();
        if (!S.ok) { hung = true; break; }
        if (S.live == 0ull) break;                                 // every box is decided
        if (rounds >= max_rounds && Etot <= ecap) { bail = true; break; }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int w = tid + 1024 * q;
            if (w < 2 * (int)W) { S.KR[2 * w] = mc_ld(&mcs[MC_K + w]); S.KR[2 * w + 1] = mc_ld(&mcs[MC_R + w]); }
        }
        __syncthreads();
        int out = 0;
        const int nch = (mycnt + 3) >> 2;
        for (int c0 = 0; c0 < nch; c0 += 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = (c0 + u < nch) ? *(const uint4 *)(src_list + 4 * (tid + 1024 * (c0 + u))) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                u32 key[16];
                uint2 kr[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint4 vv = v[hh * 4 + (i >> 2)];
                    key[i] = (i & 3) == 0 ? vv.x : ((i & 3) == 1 ? vv.y : ((i & 3) == 2 ? vv.z : vv.w));
                    kr[i] = *(const uint2 *)&S.KR[2 * (key[i] >> 21)];
                }
                u32 old[16];                                     // the returning atomics of all 16 issued before the first is waited for
                u32 applied = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    old[i] = 0u;
                    if ((c0 + hh * 4) * 4 + i >= mycnt) continue;
                    const u32 src = key[i] >> 16, dst = (key[i] & 0xffffu) - (u32)own0;
                    const u32 sb = 1u << (src & 31), db = 1u << (dst & 31);
                    if (kr[i].x & sb) {
                        old[i] = atomicOr(&S.oR[dst >> 5], db);
                        applied |= 1u << i;
                    } else if (!(kr[i].y & sb)) {
                        atomicOr(&S.oBk[dst >> 5], db);
                        wmine[((out >> 2) << 12) + (out & 3)] = key[i];
                        ++out;
                    }
                }
                if (applied) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        if (!((applied >> i) & 1u)) continue;
                        const u32 dst = (key[i] & 0xffffu) - (u32)own0, db = 1u << (dst & 31);
                        if (old[i] & db) atomicOr(&S.oR2[dst >> 5], db);
                        else S.single[dst] = (unsigned short)(key[i] >> 16);
                    }
                }
            }
        }
        mycnt = out;
        src_list = wlist;
        __syncthreads();
        if (tid < 32) {                                            // what nothing blocks is kept
            S.oK[tid] |= (tid < nown ? nms_valid32(ow0 + tid, n) : 0u) & ~(S.oK[tid] | S.oR[tid] | S.oBk[tid]);
            S.oBk[tid] = 0u;
        }
        ++rounds;
    }
    if (hung) {                                                   // uniform within the workgroup; the others time out as well
        if (tid == 0) {
            if (Etot <= ecap) { h->fpdone = 0; }                  // the serial chain does this image
            else { h->fpdone = 1; keep_count[img] = -1; }
            h->rounds = -1; h->npairs = Etot;
        }
        return;
    }
    if (bail) {                                                   // uniform over the grid: k_nms_reduce_csr<SEG> does this image from scratch
        if (g == 0 && tid == 0) { h->fpdone = 0; h->rounds = rounds; h->npairs = Etot; }
        return;
    }
    // ---- the kept set is final and every workgroup's counts are out: keep list in rank order, each workgroup its targets
    if (wave == 0) {
        int kc = (int)mc_ld(&mcs[MC_KCNT + lane]), km = (int)mc_ld(&mcs[MC_KMAX + lane]), sg = (int)mc_ld(&mcs[MC_SNGL + lane]);
        int incl = kc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { km = max(km, __shfl_xor(km, d, 64)); sg = max(sg, __shfl_xor(sg, d, 64)); }
        const int total = __shfl(incl, 63, 64), mybase = __shfl(incl - kc, g, 64);
        const u32 kw = lane < nown ? S.oK[lane] : 0u;            // word offsets of this workgroup's kept bits
        int wi = __builtin_popcount(kw);
        const int wc = wi;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(wi, d, 64); if (lane >= d) wi += up; }
        if (lane < 32) S.off[lane] = mybase + wi - wc;
        if (lane == 0) { S.cnt[0] = total; S.cnt[1] = km; S.cnt[2] = sg; }
    }
    __syncthreads();
    const int total = S.cnt[0], km = S.cnt[1] - 1;                 // km: the last kept box (-1: none)
    const int drop = (quirk && total > 0 && S.cnt[2] - 1 != km) ? 1 : 0;
    if (tid < (1 << lsh)) {
        const int t = own0 + tid;
        const u32 kw = S.oK[tid >> 5];
        if (((kw >> (tid & 31)) & 1u) && !(drop && t == km))
            keep[S.off[tid >> 5] + __builtin_popcount(kw & ((1u << (tid & 31)) - 1u))] = (int64_t)order[t];
    }
    if (g == 0 && tid == 0) { keep_count[img] = total - drop; h->fpdone = 1; h->rounds = rounds; h->npairs = Etot; }
}

// The same fixed point WITHOUT the per-round grid barrier (round 3, second form; the default).  Nothing in the rounds above
// needs the workgroups in step: KEPT / REMOVED bits are final once set and a source that still looks undecided only keeps its
// entry alive, so a workgroup may pass over its list whenever the sources' words it can see have changed, against whatever
// the others have published by then.  Per pass: publish the own words (stores, not waited for), read all words (one trip),
// the list pass; no arrival counter, no spin, no second trip for the snapshot -- two dependent trips to memory instead of
// four, and a workgroup whose list is empty is finished whatever the others still do.  One grid-wide wait remains, at the
// end: the keep list needs every workgroup's kept count.  A pass is only made when the snapshot differs from the last one
// (own words included); otherwise the workgroup sleeps and polls again.  `max_rounds` counts list passes of one workgroup;
// the first to exceed it (a ladder of boxes each suppressing the next) raises the image's bail flag and every workgroup
// leaves the image to the serial chain, as before.  Which of the two an image takes can depend on timing; the keep list
// does not.
#define MC_ARR (MC_SLOT + 0)        // workgroups that are finished (or have given up)
#define MC_BAIL (MC_SLOT + 1)       // 1: a workgroup ran out of rounds, 2: one stopped seeing progress
#define MC_PASSES (MC_SLOT + 2)     // most list passes any workgroup made
#define MC_IDLE_LIMIT (1 << 17)
#define MC_LDS_LIST 8192       // entries of a pair list that k_nms_resolve_async keeps in LDS (32 KB)
// One pass of a thread over its part of a pair list (k_nms_resolve_async): the 16-byte chunks tid, tid + 1024, ... of `lsrc`,
// NCH of them in flight; survivors go back to the same chunks of `ldst` in order (in place: the write index never passes the
// read index).  Sparse lists leave a thread one or two entries after the first passes, and the 32-entry batch -- 32 LDS
// look-ups and as many skip tests whatever the count -- was most of a pass (stamps: 2.4 us of 4.6); a wave whose threads all
// hold at most eight entries takes the 8-entry batch.
template <int NCH>
__device__ __forceinline__ int fp_list_pass(const u32 *lsrc, u32 *ldst, int mycnt, int tid, NmsFpShared &S, u32 own0)
{
    constexpr int NE = NCH * 4 > 16 ? 16 : NCH * 4;               // entries whose look-ups and returning atomics are in flight together
    int out = 0;
    const int nch = (mycnt + 3) >> 2;
    for (int c0 = 0; c0 < nch; c0 += NCH) {
        uint4 v[NCH];
#pragma unroll
        for (int u = 0; u < NCH; ++u)
            v[u] = (c0 + u < nch) ? *(const uint4 *)(lsrc + 4 * (tid + 1024 * (c0 + u))) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int hh = 0; hh < NCH * 4 / NE; ++hh) {
            u32 key[NE];
            uint2 kr[NE];
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const uint4 vv = v[hh * (NE / 4) + (i >> 2)];
                key[i] = (i & 3) == 0 ? vv.x : ((i & 3) == 1 ? vv.y : ((i & 3) == 2 ? vv.z : vv.w));
                kr[i] = *(const uint2 *)&S.KR[2 * (key[i] >> 21)];
            }
            u32 old[NE];                                         // the returning atomics of all of them issued before the first is waited for
            u32 applied = 0;
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                old[i] = 0u;
                if ((c0 + hh * (NE / 4)) * 4 + i >= mycnt) continue;
                const u32 src = key[i] >> 16, dst = (key[i] & 0xffffu) - own0;
                const u32 sb = 1u << (src & 31), db = 1u << (dst & 31);
                if (kr[i].x & sb) {
                    old[i] = atomicOr(&S.oR[dst >> 5], db);
                    applied |= 1u << i;
                } else if (!(kr[i].y & sb)) {
                    atomicOr(&S.oBk[dst >> 5], db);
                    ldst[((out >> 2) << 12) + (out & 3)] = key[i];
                    ++out;
                }
            }
            if (applied) {
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    if (!((applied >> i) & 1u)) continue;
                    const u32 dst = (key[i] & 0xffffu) - own0, db = 1u << (dst & 31);
                    if (old[i] & db) atomicOr(&S.oR2[dst >> 5], db);
                    else S.single[dst] = (unsigned short)(key[i] >> 16);
                }
            }
        }
    }
    return out;
}

// Diagnostic build (-DDG_NMS_STAMPS, tools/dbg_nms_stamps.py; never the product): where a workgroup of k_nms_resolve_async
// spends its cycles -- per phase of a pass, then per phase of the ending -- for image 0 of the last launch.
#ifdef DG_NMS_STAMPS
__device__ unsigned long long nms_dbg[64][8];
extern "C" int dg_debug_nms_stamps(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(nms_dbg), sizeof(nms_dbg)) == hipSuccess ? 0 : -4; }
#define FPSTAMP_DECL unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter(), tstart = tprev
#define FPSTAMP(k) do { if (tid == 0) { const unsigned long long now = __builtin_readcyclecounter(); acc[k] += now - tprev; tprev = now; } } while (0)
#define FPSTAMP_LOOP_END do { if (tid == 0) { acc[7] = passes; acc[5] = __builtin_readcyclecounter() - tstart; acc[6] = acc[0]; acc[0] = acc[1] = acc[2] = acc[3] = 0; tprev = __builtin_readcyclecounter(); } } while (0)
#define FPSTAMP_OUT do { if (tid == 0 && img == 0) { acc[4] = __builtin_readcyclecounter() - tstart; for (int k = 0; k < 8; ++k) nms_dbg[g][k] = acc[k]; } } while (0)
#else
#define FPSTAMP_DECL do { } while (0)
#define FPSTAMP(k) do { } while (0)
#define FPSTAMP_LOOP_END do { } while (0)
#define FPSTAMP_OUT do { } while (0)
#endif
__global__ __launch_bounds__(1024) void k_nms_resolve_async(NmsHdr *__restrict__ hdr, const int32_t *__restrict__ counts, int64_t nmax,
                                                            int64_t Np, int64_t W, int64_t lcap, int lsh, int64_t ecap,
                                                            const int32_t *__restrict__ lcnt, const int32_t *__restrict__ order,
                                                            u64 *__restrict__ maskws, u32 *__restrict__ mc, int max_rounds, int quirk,
                                                            int64_t *__restrict__ keep, int32_t *__restrict__ keep_count)
{
    __shared__ NmsFpShared S;
    __shared__ __attribute__((aligned(16))) u32 s_list[MC_LDS_LIST];   // a list of at most this many entries lives here, not in memory
    const int img = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    NmsHdr *h = hdr + img;
    lcnt += img * 64 * LC_STRIDE;
    int Etot = lcnt[lane * LC_STRIDE];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) Etot += __shfl_xor(Etot, d, 64);
    if (h->mode != 0) {                                          // dense mask; if a pair list overflowed, say how many pairs there were
        if (g == 0 && tid == 0) h->npairs = Etot;
        return;
    }
    const int n = nms_count(counts, img, nmax);
    order += (int64_t)img * Np;
    keep += (int64_t)img * nmax;
    u32 *mcs = mc + (int64_t)img * MC_WORDS;
    const u32 *list = (const u32 *)(maskws + (int64_t)img * Np * W) + (int64_t)g * lcap;       // 64 pair lists, then the 64 work lists
    u32 *wlist = (u32 *)(maskws + (int64_t)img * Np * W) + (int64_t)(MC_G + g) * lcap;
    u32 *wmine = wlist + 4 * tid;
    const int E = lcnt[g * LC_STRIDE];                           // <= lcap (mode would be 1 otherwise)
    const int own0 = g << lsh, ow0 = own0 >> 5, nown = (1 << lsh) >> 5;
    const bool inlds = E <= MC_LDS_LIST;                         // (uniform) no store to memory, and so no wait for one, in a pass
    if (tid < 32) { S.oK[tid] = 0u; S.oR[tid] = 0u; S.oR2[tid] = 0u; S.oBk[tid] = 0u; }
    __syncthreads();
    // round 0: boxes nothing suppresses are kept
    for (int64_t c0 = 0; c0 * 4 < E; c0 += 8 * 1024) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t c = c0 + u * 1024 + tid;
            v[u] = (c * 4 < E) ? *(const uint4 *)(list + 4 * c) : make_uint4(0, 0, 0, 0);
            if (inlds && c * 4 < E) *(uint4 *)(s_list + 4 * c) = v[u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t e = (c0 + u * 1024 + tid) * 4;
            if (e < E) atomicOr(&S.oBk[((v[u].x & 0xffffu) >> 5) - ow0], 1u << (v[u].x & 31u));
            if (e + 1 < E) atomicOr(&S.oBk[((v[u].y & 0xffffu) >> 5) - ow0], 1u << (v[u].y & 31u));
            if (e + 2 < E) atomicOr(&S.oBk[((v[u].z & 0xffffu) >> 5) - ow0], 1u << (v[u].z & 31u));
            if (e + 3 < E) atomicOr(&S.oBk[((v[u].w & 0xffffu) >> 5) - ow0], 1u << (v[u].w & 31u));
        }
    }
    __syncthreads();
    if (tid < 32) { S.oK[tid] = (tid < nown ? nms_valid32(ow0 + tid, n) : 0u) & ~S.oBk[tid]; S.oBk[tid] = 0u; }
    const u32 *src_list = list;
    int mycnt = 0;
    for (int64_t c = tid; c * 4 < E; c += 1024) mycnt += (E - c * 4 >= 4) ? 4 : (int)(E - c * 4);
    FPSTAMP_DECL;
    int passes = 0, idle = 0;
    bool bail = false, stuck = false, first = true;
    for (;;) {
        __syncthreads();                                           // the own words of the last pass are complete
        if (tid < nown) { mc_st(&mcs[MC_K + ow0 + tid], S.oK[tid]); mc_st(&mcs[MC_R + ow0 + tid], S.oR[tid]); }
        const int anyleft = __syncthreads_or(mycnt != 0);
        FPSTAMP(0);
        if (anyleft == 0) break;              // the list is empty: every owned target is decided
        if (passes >= max_rounds && Etot <= ecap) { bail = true; break; }
        // what the others have published by now; the own words as they are here
        int ch = first ? 1 : 0;
        {
            // the thread's four words and the bail flag: five agent-scope loads issued together, ONE wait (the compiler waits
            // for every atomic load in turn, and with the own-word branches between them that was four trips a pass)
            const int w0 = tid, w1 = tid + 1024;
            const bool v1 = w1 < 2 * (int)W;                       // (w0 < 2 W always: W > 64)
            const u32 *pk0 = &mcs[MC_K + w0], *pr0 = &mcs[MC_R + w0], *pk1 = &mcs[MC_K + (v1 ? w1 : w0)], *pr1 = &mcs[MC_R + (v1 ? w1 : w0)];
            const u32 *pb = &mcs[MC_BAIL];
            u32 k0, r0, k1, r1, bl;
            asm volatile("global_load_dword %0, %5, off sc1\n\tglobal_load_dword %1, %6, off sc1\n\tglobal_load_dword %2, %7, off sc1\n\t"
                         "global_load_dword %3, %8, off sc1\n\tglobal_load_dword %4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(k0), "=&v"(r0), "=&v"(k1), "=&v"(r1), "=&v"(bl)
                         : "v"(pk0), "v"(pr0), "v"(pk1), "v"(pr1), "v"(pb) : "memory");
            if (w0 >= ow0 && w0 < ow0 + nown) { k0 = S.oK[w0 - ow0]; r0 = S.oR[w0 - ow0]; }      // the own words as they are here
            if (w1 >= ow0 && w1 < ow0 + nown) { k1 = S.oK[w1 - ow0]; r1 = S.oR[w1 - ow0]; }
            ch |= (k0 != S.KR[2 * w0]) | (r0 != S.KR[2 * w0 + 1]);
            S.KR[2 * w0] = k0; S.KR[2 * w0 + 1] = r0;
            if (v1) { ch |= (k1 != S.KR[2 * w1]) | (r1 != S.KR[2 * w1 + 1]); S.KR[2 * w1] = k1; S.KR[2 * w1 + 1] = r1; }
            if (tid == 0) S.sngl = (int)bl;                        // somebody has given up: so does everybody
        }
        const int changed = __syncthreads_or(ch);
        FPSTAMP(1);
        if (S.sngl) { bail = true; break; }
        if (changed == 0) {                                        // nothing new: no pass
            if (++idle > MC_IDLE_LIMIT) { stuck = true; break; }
            __builtin_amdgcn_s_sleep(16);
            FPSTAMP(2);
            continue;
        }
        first = false; idle = 0;
        // (a wave's threads hold about equally many entries: which batch a wave takes is decided per wave)
        const bool few = __ballot(mycnt > 8) == 0ull;
        int out;
        if (inlds) out = few ? fp_list_pass<2>(s_list, s_list + 4 * tid, mycnt, tid, S, (u32)own0) : fp_list_pass<8>(s_list, s_list + 4 * tid, mycnt, tid, S, (u32)own0);
        else out = few ? fp_list_pass<2>(src_list, wmine, mycnt, tid, S, (u32)own0) : fp_list_pass<8>(src_list, wmine, mycnt, tid, S, (u32)own0);
        mycnt = out;
        src_list = wlist;
        __syncthreads();
        FPSTAMP(3);
        if (tid < 32) {                                            // what nothing blocks is kept
            S.oK[tid] |= (tid < nown ? nms_valid32(ow0 + tid, n) : 0u) & ~(S.oK[tid] | S.oR[tid] | S.oBk[tid]);
            S.oBk[tid] = 0u;
        }
        ++passes;
    }
    FPSTAMP_LOOP_END;
    // ---- finished (the final words went out at the top of the last turn), out of rounds, or stuck: tell the others, wait for all
    if (tid == 0) S.sngl = 0;
    __syncthreads();
    if (!bail && !stuck) {
        int v = 0;                                                 // one owned target per thread: its sole suppressor
        if (tid < (1 << lsh) && ((S.oR[tid >> 5] & ~S.oR2[tid >> 5]) >> (tid & 31)) & 1u) v = (int)S.single[tid] + 1;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
        if (lane == 0 && v) atomicMax(&S.sngl, v);
    }
    __syncthreads();
    FPSTAMP(0);
    if (wave == 1) {
        const u32 kw = lane < nown ? S.oK[lane] : 0u;
        int kc = __builtin_popcount(kw), km = kw ? (ow0 + lane) * 32 + 32 - __builtin_clz(kw) : 0;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { kc += __shfl_xor(kc, d, 64); km = max(km, __shfl_xor(km, d, 64)); }
        if (lane == 0) {
            // kept count (<= 1024), highest kept rank + 1 and highest sole suppressor + 1 (<= 65536 each) in ONE 64-bit word per
            // workgroup (over the MC_KCNT / MC_KMAX areas): one trip for whoever reads them, not three
            __hip_atomic_store((unsigned long long *)&mcs[MC_KCNT + 2 * g],
                               (unsigned long long)kc | ((unsigned long long)km << 16) | ((unsigned long long)S.sngl << 40),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (bail || stuck) atomicMax(&mcs[MC_BAIL], stuck ? 2u : 1u);   // (what the others poll while they still pass)
            atomicMax(&mcs[MC_PASSES], (u32)passes);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // what this thread published has arrived
    __syncthreads();
    FPSTAMP(1);
    if (tid == 0) {
        // arrivals in the low byte, those that ran out of rounds and those that were stuck in the next two: the word the wait
        // spins on says everything
        atomicAdd(&mcs[MC_ARR], 1u | (bail ? 1u << 8 : 0u) | (stuck ? 1u << 16 : 0u));
        u32 spins = 0, v;
        int ok = 1;
        while (((v = mc_ld(&mcs[MC_ARR])) & 0xffu) < (u32)MC_G) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > MC_SPIN_LIMIT) { ok = 0; break; }
        }
        S.ok = (ok && (v >> 16) == 0u) ? 1 : 0;
        S.live = (unsigned long long)((v >> 8) & 0xffu);
    }
    __syncthreads();
    FPSTAMP(2);
    if (!S.ok) {                                                  // a workgroup never arrived, or one stopped seeing progress
        if (g == 0 && tid == 0) {
            if (Etot <= ecap) { h->fpdone = 0; }                  // the serial chain does this image
            else { h->fpdone = 1; keep_count[img] = -1; }
            h->rounds = -1; h->npairs = Etot;
        }
        return;
    }
    if (S.live != 0ull) {                                         // uniform over the grid: k_nms_reduce_csr<SEG> does this image from scratch
        if (g == 0 && tid == 0) { h->fpdone = 0; h->rounds = (int)mc_ld(&mcs[MC_PASSES]); h->npairs = Etot; }
        return;
    }
    __syncthreads();
    // ---- the kept set is final and every workgroup's counts are out: keep list in rank order, each workgroup its targets
    if (wave == 0) {
        const unsigned long long pk = __hip_atomic_load((const unsigned long long *)&mcs[MC_KCNT + 2 * lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int kc = (int)(pk & 0xffffull), km = (int)((pk >> 16) & 0xffffffull), sg = (int)(pk >> 40);
        int incl = kc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { km = max(km, __shfl_xor(km, d, 64)); sg = max(sg, __shfl_xor(sg, d, 64)); }
        const int total = __shfl(incl, 63, 64), mybase = __shfl(incl - kc, g, 64);
        const u32 kw = lane < nown ? S.oK[lane] : 0u;            // word offsets of this workgroup's kept bits
        int wi = __builtin_popcount(kw);
        const int wc = wi;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(wi, d, 64); if (lane >= d) wi += up; }
        if (lane < 32) S.off[lane] = mybase + wi - wc;
        if (lane == 0) { S.cnt[0] = total; S.cnt[1] = km; S.cnt[2] = sg; }
    }
    __syncthreads();
    FPSTAMP(3);
    const int total = S.cnt[0], km = S.cnt[1] - 1;                 // km: the last kept box (-1: none)
    const int drop = (quirk && total > 0 && S.cnt[2] - 1 != km) ? 1 : 0;
    if (tid < (1 << lsh)) {
        const int t = own0 + tid;
        const u32 kw = S.oK[tid >> 5];
        if (((kw >> (tid & 31)) & 1u) && !(drop && t == km))
            keep[S.off[tid >> 5] + __builtin_popcount(kw & ((1u << (tid & 31)) - 1u))] = (int64_t)order[t];
    }
    if (g == 0 && tid == 0) { keep_count[img] = total - drop; h->fpdone = 1; h->rounds = (int)mc_ld(&mcs[MC_PASSES]); h->npairs = Etot; }
    FPSTAMP_OUT;
}

// ---------------------------------------------------------------- host
extern "C" size_t dg_nms_workspace_bytes(int64_t batch, int64_t nmax)
{
    if (batch <= 0 || nmax < 0) return 0;
    return nms_layout(batch, nmax, nullptr).total;
}

// DG_NMS_PRUNE: 0 = never take the pruned path, 1 (default) = where the candidate count says so, 2 = wherever it applies
// (64 < W <= 1024, finite boxes, 2^-20 <= thr < inf), whatever the candidate count (tests).  DG_NMS_FP_ROUNDS: rounds after
// which k_nms_resolve_mc hands over to the serial chain (default 128; 1 in tests sends every image there).
static int nms_prune_mode() { const char *e = getenv("DG_NMS_PRUNE"); return e ? atoi(e) : 1; }
static int nms_fp_rounds() { const char *e = getenv("DG_NMS_FP_ROUNDS"); return e ? atoi(e) : 128; }

static int nms_check(const char *fn, int64_t batch, int64_t nmax, void *ws, size_t ws_bytes, NmsWs *out)
{
    DG_REQUIRE(batch >= 0 && nmax >= 0, "%s: negative size", fn);
    DG_REQUIRE(nmax <= (1 << 20), "%s: nmax %lld > 1048576 unsupported", fn, (long long)nmax);
    DG_REQUIRE(batch <= 65535, "%s: batch > 65535", fn);
    if (batch == 0 || nmax == 0) { out->total = 0; return DG_OK; }
    DG_REQUIRE(ws && (((uintptr_t)ws) & 255) == 0, "%s: workspace null or not 256-byte aligned", fn);
    *out = nms_layout(batch, nmax, ws);
    if (ws_bytes < out->total) { dg_set_error("%s: workspace %zu < %zu bytes", fn, ws_bytes, out->total); return DG_ENOMEM; }
    return DG_OK;
}

template <int NP2>
static void launch_sort_small(const NmsWs &w, const float *boxes, const float *scores, const int32_t *counts,
                              int64_t batch, int64_t nmax, hipStream_t st)
{
    constexpr int T = (NP2 / 2 > 1024) ? 1024 : (NP2 / 2 < 64 ? 64 : NP2 / 2);
    hipLaunchKernelGGL((k_nms_sort_small<NP2, T>), dim3((unsigned)batch), dim3(T), 0, st, (const float4 *)boxes, scores, counts,
                       nmax, w.Np, w.x1, w.y1, w.x2, w.y2, w.area, w.order);
}

extern "C" int dg_nms_phase_sort(const float *boxes, const float *scores, const int32_t *counts, int64_t batch, int64_t nmax,
                                 void *ws, size_t ws_bytes, dg_stream_t s)
{
    NmsWs w;
    int rc = nms_check("nms_sort", batch, nmax, ws, ws_bytes, &w);
    if (rc || batch == 0 || nmax == 0) return rc;
    DG_REQUIRE(boxes && scores, "nms_sort: null pointer");
    DG_REQUIRE((((uintptr_t)boxes) & 15) == 0, "nms_sort: boxes must be 16-byte aligned");
    hipStream_t st = (hipStream_t)s;
    const bool prune = w.pruned && nms_prune_mode() != 0;
    switch (w.Np2) {
    case 64: launch_sort_small<64>(w, boxes, scores, counts, batch, nmax, st); break;
    case 128: launch_sort_small<128>(w, boxes, scores, counts, batch, nmax, st); break;
    case 256: launch_sort_small<256>(w, boxes, scores, counts, batch, nmax, st); break;
case 512: launch_sort_small<512>(w, boxes, scores, counts, batch, nmax, st); break;
    case 1024: launch_sort_small<1024>(w, boxes, scores, counts, batch, nmax, st); break;
    case 2048: launch_sort_small<2048>(w, boxes, scores, counts, batch, nmax, st); break;
    case 4096: launch_sort_small<4096>(w, boxes, scores, counts, batch, nmax, st); break;
    case 8192: case 16384: case 32768: case 65536:
        // (pruned path: one more workgroup per image finds the x-bucket order beside the sort, see xbucket_image)
        hipLaunchKernelGGL(k_nms_sort_chunks, dim3((unsigned)(w.Np2 / SORT_CHUNK) + (prune ? 1u : 0u), (unsigned)batch), dim3(1024), 0, st, scores,
                           counts, nmax, w.Np2, w.keys, (const float4 *)boxes, w.Np, nms_prune_mode(), w.hdr, w.start, w.posu, w.lcnt);
        hipLaunchKernelGGL(k_nms_merge_rank, dim3((unsigned)(w.Np2 / 256), (unsigned)batch), dim3(256), 0, st, (const float4 *)boxes, w.keys,
                           counts, nmax, w.Np, w.Np2, w.x1, w.y1, w.x2, w.y2, w.area, w.order, prune ? w.posu : nullptr, w.hdr, w.start, w.bbox, w.bar);
        break;
    default: {
        const int64_t Np2 = w.Np2;
        hipLaunchKernelGGL(k_nms_make_keys, dim3((unsigned)(Np2 / 256), (unsigned)batch), dim3(256), 0, st, scores, counts, nmax, Np2, w.keys);
        dim3 gl((unsigned)(Np2 / SORT_CHUNK), (unsigned)batch), gg((unsigned)(Np2 / 2 / 256), (unsigned)batch);
        hipLaunchKernelGGL(k_bitonic_lds, gl, dim3(1024), 0, st, w.keys, Np2, (int64_t)2, (int64_t)SORT_CHUNK, 1);
        for (int64_t k = 2 * SORT_CHUNK; k <= Np2; k <<= 1) {
            for (int64_t j = k >> 1; j >= SORT_CHUNK; j >>= 1)
                hipLaunchKernelGGL(k_bitonic_global, gg, dim3(256), 0, st, w.keys, Np2, k, j);
            hipLaunchKernelGGL(k_bitonic_lds, gl, dim3(1024), 0, st, w.keys, Np2, k, k, SORT_CHUNK / 2);
        }
        hipLaunchKernelGGL(k_nms_gather, dim3((unsigned)dg_cdiv(w.Np, 256), (unsigned)batch), dim3(256), 0, st, (const float4 *)boxes,
                           w.keys, counts, nmax, w.Np, Np2, w.x1, w.y1, w.x2, w.y2, w.area, w.order);
    }
    }
    DG_CHECK_LAUNCH("nms sort");
    return DG_OK;
}

static int nms_phase_mask(const int32_t *counts, int64_t batch, int64_t nmax, float threshold, void *ws, size_t ws_bytes, dg_stream_t s,
                          bool standalone)
{
    NmsWs w;
    int rc = nms_check("nms_mask", batch, nmax, ws, ws_bytes, &w);
    if (rc || batch == 0 || nmax == 0) return rc;
    // midpoint between thr and the float below it, and which way a quotient exactly on it rounds (see k_nms_mask)
    double mid;
    const bool tie_even = dg_nms_threshold_mid(threshold, &mid);
    const int fast_ok = threshold >= 0x1p-20f && threshold < INFINITY;
    const float below = fast_ok ? nextafterf(threshold, -INFINITY) : threshold;
    // 32-word tiles; small launches (one image of a few thousand boxes) take 8-word tiles so that more than a handful of
    // workgroups exist: 2000 boxes = 32 row blocks x 1 tile would keep 32 of 256 CUs busy
    const bool small = dg_cdiv(w.W, 32) * w.W * batch < 512;
    const NmsHdr *hdr = nullptr;
    if (w.pruned && nms_prune_mode() != 0) {
        // the pruned pair search for the images xbucket_image chose (hdr.mode == 0); the dense mask below runs for the others
        hdr = w.hdr;
        if (standalone) hipLaunchKernelGGL(k_nms_reset_pairs, dim3((unsigned)dg_cdiv(batch * 64, 256)), dim3(256), 0, (hipStream_t)s, w.hdr, w.lcnt, (int)batch);
        if (!fast_ok) {
            hipLaunchKernelGGL(k_nms_force_dense, dim3((unsigned)dg_cdiv(batch, 256)), dim3(256), 0, (hipStream_t)s, w.hdr, (int)batch);
        } else {
            unsigned parts = (unsigned)(w.W >= 1024 ? 2 : (w.W >= 512 ? 4 : 8));
            if (getenv("DG_NMS_BAND_PARTS")) parts = (unsigned)atoi(getenv("DG_NMS_BAND_PARTS"));
            const dim3 g(parts, (unsigned)w.W, (unsigned)batch);
            const int force = nms_prune_mode();
            if (force != 2)
                hipLaunchKernelGGL(k_nms_estimate, dim3(256, (unsigned)batch), dim3(256), 0, (hipStream_t)s, w.bbox, w.bar, w.hdr, counts, nmax,
                                   w.Np, threshold);
            if (tie_even)
                hipLaunchKernelGGL((k_nms_band<true>), g, dim3(256), 0, (hipStream_t)s, w.bbox, w.bar, w.start, w.hdr, counts, nmax, w.Np,
                                   w.lcap, w.lsh, threshold, below, mid, w.mask, w.W, w.lcnt, force, w.mc);
            else
                hipLaunchKernelGGL((k_nms_band<false>), g, dim3(256), 0, (hipStream_t)s, w.bbox, w.bar, w.start, w.hdr, counts, nmax, w.Np,
                                   w.lcap, w.lsh, threshold, below, mid, w.mask, w.W, w.lcnt, force, w.mc);
        }
        DG_CHECK_LAUNCH("k_nms_band");
    }
#define MASK_LAUNCH(TE, MTW)                                                                                                     \
    hipLaunchKernelGGL((k_nms_mask<TE, MTW>), dim3((unsigned)dg_cdiv(w.W, MTW), (unsigned)w.W, (unsigned)batch), dim3(256), 0,   \
                       (hipStream_t)s, w.x1, w.y1, w.x2, w.y2, w.area, counts, nmax, w.Np, w.W, w.SW, threshold, below, mid,     \
                       fast_ok, w.mask, w.rsum, hdr)
    if (tie_even) { if (small) MASK_LAUNCH(true, 8); else MASK_LAUNCH(true, 32); }
    else { if (small) MASK_LAUNCH(false, 8); else MASK_LAUNCH(false, 32); }
#undef MASK_LAUNCH
    DG_CHECK_LAUNCH("k_nms_mask");
    return DG_OK;
}

extern "C" int dg_nms_phase_mask(const int32_t *counts, int64_t batch, int64_t nmax, float threshold,
                                 void *ws, size_t ws_bytes, dg_stream_t s)
{
    return nms_phase_mask(counts, batch, nmax, threshold, ws, ws_bytes, s, true);
}

static int nms_phase_reduce(const int32_t *counts, int64_t batch, int64_t nmax, int quirk_drop_last,
                            int64_t *keep, int32_t *keep_count, void *ws, size_t ws_bytes, dg_stream_t s, bool standalone)
{
    NmsWs w;
    int rc = nms_check("nms_reduce", batch, nmax, ws, ws_bytes, &w);
    if (rc || batch == 0) return rc;
    DG_REQUIRE(keep_count, "nms_reduce: null keep_count");
    hipStream_t st = (hipStream_t)s;
    if (nmax == 0) { hipError_t e = hipMemsetAsync(keep_count, 0, (size_t)batch * 4, st); return e == hipSuccess ? DG_OK : DG_EHIP; }
    DG_REQUIRE(keep, "nms_reduce: null keep");
#define RED(T, WPT) hipLaunchKernelGGL((k_nms_reduce<T, WPT>), dim3((unsigned)batch), dim3(T), 0, st, w.mask, w.order, counts, \
                                       nmax, w.Np, w.W, quirk_drop_last, keep, keep_count)
    const NmsHdr *hdr = nullptr;
    if (w.pruned && nms_prune_mode() != 0) {
        hdr = w.hdr;
        // (the mask phase's k_nms_band zeroed w.mc; the reduce phase run again on its own -- bench.py times it so -- does it itself)
        if (standalone && hipMemsetAsync(w.mc, 0, (size_t)batch * MC_WORDS * 4, st) != hipSuccess) { dg_set_error("nms_reduce: memset failed"); return DG_EHIP; }
        // DG_NMS_FP_SYNC=1: the fixed point with a grid barrier per round (k_nms_resolve_mc) instead of the free-running one
        if (getenv("DG_NMS_FP_SYNC") && atoi(getenv("DG_NMS_FP_SYNC")) == 1)
            hipLaunchKernelGGL(k_nms_resolve_mc, dim3(MC_G, (unsigned)batch), dim3(1024), 0, st, w.hdr, counts, nmax, w.Np, w.W, w.lcap, w.lsh,
                               w.ecap, w.lcnt, w.order, w.mask, w.mc, nms_fp_rounds(), quirk_drop_last, keep, keep_count);
        else
            hipLaunchKernelGGL(k_nms_resolve_async, dim3(MC_G, (unsigned)batch), dim3(1024), 0, st, w.hdr, counts, nmax, w.Np, w.W, w.lcap, w.lsh,
                               w.ecap, w.lcnt, w.order, w.mask, w.mc, nms_fp_rounds(), quirk_drop_last, keep, keep_count);
        hipLaunchKernelGGL(k_nms_reduce_csr<true>, dim3((unsigned)batch), dim3(1024), 0, st, w.mask, w.order, counts, nmax, w.Np,
                           w.W, w.ecap, (const int32_t *)nullptr, (const int32_t *)nullptr, w.ekey, w.eval, quirk_drop_last, keep, keep_count,
                           w.hdr, w.diag, w.lcnt, w.lcap);
        DG_CHECK_LAUNCH("k_nms_resolve_mc");
    }
    if (w.W < NMS_CSR_MIN_W) RED(64, 1);
    else if (w.W <= 1024) {
        // compact the mask, then the LDS-fed reduce; an image whose entries do not fit falls to the summary-driven kernel
        // DG_NMS_FORCE_FALLBACK=1 (tests): no image "fits" the compact form, every one takes the dense-mask reduce
        const bool force_fb = getenv("DG_NMS_FORCE_FALLBACK") && atoi(getenv("DG_NMS_FORCE_FALLBACK")) == 1;
        const int64_t ecap_fit = force_fb ? (int64_t)-1 : w.ecap;
        if (w.W <= 64 && !force_fb) {
            // one launch: the non-zero mask words gathered into LDS and the boxes decided in rounds (k_nms_resolve_small)
            const int fb = getenv("DG_NMS_FORCE_FALLBACK") ? atoi(getenv("DG_NMS_FORCE_FALLBACK")) : 0;   // 2: its serial chain
            hipLaunchKernelGGL(k_nms_resolve_small, dim3((unsigned)batch), dim3(1024), 0, st, w.mask, w.rsum, w.order, counts, nmax,
                               w.Np, w.W, fb == 2 ? 0 : nms_fp_rounds(), quirk_drop_last, keep, keep_count);
            DG_CHECK_LAUNCH("k_nms_resolve_small");
            return DG_OK;
        }
        if (w.W <= 64) {
            hipLaunchKernelGGL(k_nms_csr_build_small, dim3((unsigned)batch, batch < 32 ? 8u : 1u), dim3(1024), 0, st, w.mask, w.rsum, counts, nmax, w.Np, w.W,
                               ecap_fit, w.ecap, w.boff, w.cflag, w.ekey, w.eval);
        } else {
            hipLaunchKernelGGL(k_nms_csr_count, dim3((unsigned)(w.Np / 256 + (w.Np % 256 != 0)), (unsigned)batch), dim3(256), 0, st, w.rsum,
                               counts, nmax, w.Np, w.W, w.SW, w.bcnt, hdr);
            hipLaunchKernelGGL(k_nms_csr_scan, dim3((unsigned)batch), dim3(1024), 0, st, w.bcnt, counts, nmax, w.W, ecap_fit, w.boff, w.cursor,
                               w.cflag, hdr);
            if ((w.SW & (w.SW - 1)) == 0 && w.SW <= 64) {
                int sh = 0;
                while ((1ll << sh) < w.SW) ++sh;
                hipLaunchKernelGGL(k_nms_csr_fill_wave, dim3((unsigned)dg_cdiv(w.Np * w.SW, 256), (unsigned)batch), dim3(256), 0, st, w.mask,
                                   w.rsum, counts, nmax, w.Np, w.W, w.SW, sh, w.ecap, w.cflag, w.cursor, w.ekey, w.eval, hdr);
            } else {
                hipLaunchKernelGGL(k_nms_csr_fill, dim3((unsigned)dg_cdiv(w.Np * w.SW, 256), (unsigned)batch), dim3(256), 0, st, w.mask, w.rsum,
                                   counts, nmax, w.Np, w.W, w.SW, w.ecap, w.cflag, w.cursor, w.ekey, w.eval, hdr);
            }
        }
        hipLaunchKernelGGL(k_nms_reduce_csr<false>, dim3((unsigned)batch), dim3(1024), 0, st, w.mask, w.order, counts, nmax, w.Np, w.W, w.ecap,
                           w.cflag, w.boff, w.ekey, w.eval, quirk_drop_last, keep, keep_count, hdr, (const u64 *)nullptr,
                           (const int32_t *)nullptr, (int64_t)0);
        // images whose compact form overflowed (dense masks); returns at once for the others
        {
            const size_t lds = (size_t)w.W * 16 + (size_t)PIPE_ROWS * PIPE_NEAR * 8;      // removed, removed1, 64 KB band
            // per device, not a stream operation (no allocation, legal during capture); set on every call: a process may drive several devices
            const hipError_t attr = hipFuncSetAttribute((const void *)k_nms_reduce_band, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (attr != hipSuccess) { dg_set_error("nms_reduce: cannot raise the LDS limit of k_nms_reduce_band: %s", hipGetErrorString(attr)); return DG_EHIP; }
            hipLaunchKernelGGL(k_nms_reduce_band, dim3((unsigned)batch), dim3(1024), lds, st, w.mask, w.order, counts, nmax,
                               w.Np, w.W, quirk_drop_last, keep, keep_count, w.cflag, hdr);
        }
    }
    else if (w.W <= 4096) RED(1024, 4);
    else RED(1024, 16);
#undef RED
    DG_CHECK_LAUNCH("k_nms_reduce");
    return DG_OK;
}

extern "C" int dg_nms_phase_reduce(const int32_t *counts, int64_t batch, int64_t nmax, int quirk_drop_last,
                                   int64_t *keep, int32_t *keep_count, void *ws, size_t ws_bytes, dg_stream_t s)
{
    return nms_phase_reduce(counts, batch, nmax, quirk_drop_last, keep, keep_count, ws, ws_bytes, s, true);
}

extern "C" int dg_nms_batched_f32(const float *boxes, const float *scores, const int32_t *counts, int64_t batch, int64_t nmax,
                                  float threshold, int quirk_drop_last, int64_t *keep, int32_t *keep_count,
                                  void *ws, size_t ws_bytes, dg_stream_t s)
{
    int rc = dg_nms_phase_sort(boxes, scores, counts, batch, nmax, ws, ws_bytes, s);
    if (rc) return rc;
    rc = nms_phase_mask(counts, batch, nmax, threshold, ws, ws_bytes, s, false);
    if (rc) return rc;
    return nms_phase_reduce(counts, batch, nmax, quirk_drop_last, keep, keep_count, ws, ws_bytes, s, false);
}

// diagnostic: which path image `img` of the last call on this workspace took (synchronous; tests and bench.py use it to
// make sure a timing or a parity claim is about the path it names)
extern "C" int dg_nms_path_info(const void *ws, size_t ws_bytes, int64_t batch, int64_t nmax, int64_t img, int32_t *out6_host)
{
    int32_t *out4_host = out6_host;
    DG_REQUIRE(out4_host && img >= 0 && img < batch, "nms_path_info: bad argument");
    NmsWs w;
    int rc = nms_check("nms_path_info", batch, nmax, (void *)ws, ws_bytes, &w);
    if (rc) return rc;
    out4_host[0] = 1; out4_host[1] = 0; out4_host[2] = 0; out4_host[3] = 0;        // dense mask, no fixed point
    out6_host[4] = 0; out6_host[5] = 0;
    if (batch == 0 || nmax == 0 || !w.pruned || nms_prune_mode() == 0) return DG_OK;
    NmsHdr h;
    if (hipMemcpy(&h, w.hdr + img, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { dg_set_error("nms_path_info: copy failed"); return DG_EHIP; }
    out4_host[0] = h.mode; out4_host[1] = h.fpdone; out4_host[2] = h.rounds; out4_host[3] = h.npairs;
    out6_host[4] = h.cpb; out6_host[5] = (int32_t)(h.estsum >> 10);               // candidates per box, estimated pairs per box
    return DG_OK;
}

// ---------------------------------------------------------------- allocation-free variant
// dg_nms_batched_f32 with its workspace inside the library: a file-scope device array, one copy per GPU, held until the
// process ends.  For callers that cannot allocate (HIP graph capture, tight serving loops).  `scratch` is only a host-side
// stand-in for the array: its device address is looked up once per device with hipGetSymbolAddress and that pointer is
// what the kernels get.  Every call on a GPU shares that GPU's array (calls overlapping on one GPU race); no phase
// relies on its previous contents (the mask / summary / run words a reduce reads are all written by the same call).
static __device__ __align__(256) unsigned char scratch[DG_NMS_NOALLOC_SCRATCH_BYTES];

#include <mutex>
#define NMS_MAX_DEVICES 64
static void *g_scratch_ptr[NMS_MAX_DEVICES];      // device address of `scratch` on each device (0 = not looked up yet)
static std::mutex g_scratch_mu;

static int nms_scratch_ptr(void **out)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NMS_MAX_DEVICES) {
        dg_set_error("nms_noalloc: no current device (or device index >= %d)", NMS_MAX_DEVICES);
        return DG_EHIP;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            dg_set_error("nms_noalloc: hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
            return DG_EHIP;
        }
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return DG_OK;
}

extern "C" size_t dg_nms_noalloc_max_bytes(void) { return (size_t)DG_NMS_NOALLOC_SCRATCH_BYTES; }

extern "C" int dg_nms_batched_noalloc_f32(const float *boxes, const float *scores, const int32_t *counts, int64_t batch,
                                          int64_t nmax, float threshold, int quirk_drop_last, int64_t *keep,
                                          int32_t *keep_count, dg_stream_t s)
{
    DG_REQUIRE(batch >= 0 && nmax >= 0, "nms_noalloc: negative size");
    const size_t need = (batch > 0 && nmax > 0) ? dg_nms_workspace_bytes(batch, nmax) : 0;
    if (need > (size_t)DG_NMS_NOALLOC_SCRATCH_BYTES) {
        dg_set_error("nms_noalloc: batch %lld x nmax %lld needs %zu bytes of scratch, the built-in buffer holds %zu "
                     "(use dg_nms_batched_f32 with a caller workspace)",
                     (long long)batch, (long long)nmax, need, (size_t)DG_NMS_NOALLOC_SCRATCH_BYTES);
        return DG_ENOMEM;
    }
    void *ws = nullptr;
    int rc = nms_scratch_ptr(&ws);
    if (rc) return rc;
    return dg_nms_batched_f32(boxes, scores, counts, batch, nmax, threshold, quirk_drop_last, keep, keep_count, ws,
                              (size_t)DG_NMS_NOALLOC_SCRATCH_BYTES, s);
}

// ---------------------------------------------------------------- per-class NMS (BASELINE.json configs[1])
// NO COUNTERPART IN THE REFERENCE: its nms (yolo/yolo_v1/yolo.py:188-203) is class-agnostic and evaluate() splits the
// survivors by class afterwards (:282-298).  Per-class mode = the reference's nms applied to each class's boxes of an
// image separately (boxes of different classes never suppress each other), quirk and tie rule included.  It is built
// from the batched NMS above without touching its kernels: every (image, class) pair becomes one ragged segment
//   bin   : stable counting sort of an image's boxes by class into segment (image * C + class), remembering each
//           box's index in the image (stable => ties keep "lower index first" inside a class);
//   nms   : dg_nms_batched_f32 over batch * C segments with their counts (same IoU bits: no coordinate offsets);
//   unbin : per image the classes' keep lists, mapped back to image indices, concatenated in class order.
#define PC_MAX_CLASSES 256
struct PcWs {
    float *boxes, *scores;
    int32_t *orig, *counts, *kcnt;
    int64_t *keep;
    void *nms_ws;
    size_t nms_bytes, total;
};
static PcWs pc_layout(int64_t batch, int64_t nmax, int64_t C, void *base)
{
    PcWs w;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p + off; off += (bytes + 255) & ~(size_t)255; return r; };
    const size_t seg = (size_t)batch * C;
    w.boxes = (float *)take(seg * nmax * 16);
    w.scores = (float *)take(seg * nmax * 4);
    w.orig = (int32_t *)take(seg * nmax * 4);
    w.counts = (int32_t *)take(seg * 4);
    w.kcnt = (int32_t *)take(seg * 4);
    w.keep = (int64_t *)take(seg * nmax * 8);
    w.nms_bytes = dg_nms_workspace_bytes((int64_t)seg, nmax);
    w.nms_ws = take(w.nms_bytes);
    w.total = off;
    return w;
}

// one workgroup per image; chunks of 1024 boxes in index order; rank inside the class = boxes of that class before
// it (earlier chunks: s_base, earlier waves of this chunk: s_wave, earlier lanes: ballot)
__global__ __launch_bounds__(1024) void k_nms_class_bin(const float4 *__restrict__ boxes, const float *__restrict__ scores,
                                                        const int64_t *__restrict__ labels, const int32_t *__restrict__ counts,
                                                        int64_t nmax, int C, float4 *__restrict__ sboxes, float *__restrict__ sscores,
                                                        int32_t *__restrict__ sorig, int32_t *__restrict__ scounts,
                                                        int32_t *__restrict__ status)
{
    __shared__ int s_base[PC_MAX_CLASSES];
    __shared__ int s_wave[16][PC_MAX_CLASSES];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = nms_count(counts, img, nmax);
    boxes += (int64_t)img * nmax;
    scores += (int64_t)img * nmax;
    labels += (int64_t)img * nmax;
    for (int c = tid; c < C; c += 1024) s_base[c] = 0;
    bool bad = false;
    for (int c0 = 0; c0 < n; c0 += 1024) {
        for (int t = tid; t < 16 * PC_MAX_CLASSES; t += 1024) (&s_wave[0][0])[t] = 0;
        __syncthreads();
        const int i = c0 + tid;
        int cls = -1;
        if (i < n) {
            const int64_t l = labels[i];
            if (l >= 0 && l < C) cls = (int)l;
            else bad = true;
        }
        // lanes of one class: peel the classes present in the wave one by one (wave-uniform loop)
        int rank = 0;
        u64 todo = __ballot(cls >= 0);
        while (todo) {
            const int c = __builtin_amdgcn_readlane(cls, __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo)));
            const u64 m = __ballot(cls == c);
            if (cls == c) {
                rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
                if (rank == 0) s_wave[wave][c] = __builtin_popcountll(m);
            }
            todo &= ~m;
        }
        __syncthreads();
        if (cls >= 0) {
            int pos = s_base[cls] + rank;
            for (int w = 0; w < wave; ++w) pos += s_wave[w][cls];
            const int64_t o = ((int64_t)img * C + cls) * nmax + pos;
            sboxes[o] = boxes[i];
            sscores[o] = scores[i];
            sorig[o] = i;
        }
        __syncthreads();
        for (int c = tid; c < C; c += 1024) {
            int s = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) s += s_wave[w][c];
            s_base[c] += s;
        }
        __syncthreads();
    }
    __syncthreads();
    for (int c = tid; c < C; c += 1024) scounts[(int64_t)img * C + c] = s_base[c];
    if (bad) atomicOr(status, 1);
}

__global__ __launch_bounds__(256) void k_nms_class_unbin(const int64_t *__restrict__ skeep, const int32_t *__restrict__ skcnt,
                                                         const int32_t *__restrict__ sorig, int64_t nmax, int C,
                                                         int64_t *__restrict__ keep, int32_t *__restrict__ keep_count,
                                                         int32_t *__restrict__ class_count)
{
    __shared__ int s_off[PC_MAX_CLASSES + 1];
    const int img = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int o = 0;
        for (int c = 0; c < C; ++c) { s_off[c] = o; o += skcnt[(int64_t)img * C + c]; }
        s_off[C] = o;
        keep_count[img] = o;
    }
    __syncthreads();
    for (int c = 0; c < C; ++c) {
        const int64_t seg = (int64_t)img * C + c;
        const int k = s_off[c + 1] - s_off[c];
        if (class_count && tid == 0) class_count[seg] = k;
        for (int r = tid; r < k; r += 256) keep[(int64_t)img * nmax + s_off[c] + r] = (int64_t)sorig[seg * nmax + skeep[seg * nmax + r]];
    }
}

extern "C" size_t dg_nms_per_class_workspace_bytes(int64_t batch, int64_t nmax, int64_t num_classes)
{
    if (batch <= 0 || nmax <= 0 || num_classes <= 0 || num_classes > PC_MAX_CLASSES) return 0;
    return pc_layout(batch, nmax, num_classes, nullptr).total;
}

extern "C" int dg_nms_batched_per_class_f32(const float *boxes, const float *scores, const int64_t *labels, const int32_t *counts,
                                            int64_t batch, int64_t nmax, int64_t num_classes, float threshold, int quirk_drop_last,
                                            int64_t *keep, int32_t *keep_count, int32_t *class_count, int32_t *status, void *ws,
                                            size_t ws_bytes, dg_stream_t s)
{
    DG_REQUIRE(batch >= 0 && nmax >= 0, "nms_per_class: negative size");
    DG_REQUIRE(num_classes >= 1 && num_classes <= PC_MAX_CLASSES, "nms_per_class: num_classes must be in [1,%d]", PC_MAX_CLASSES);
    DG_REQUIRE(batch * num_classes <= 65535, "nms_per_class: batch * num_classes > 65535");
    DG_REQUIRE(keep_count && status, "nms_per_class: null keep_count / status");
    hipStream_t st = (hipStream_t)s;
    if (batch == 0) return DG_OK;
    if (nmax == 0) {
        if (hipMemsetAsync(keep_count, 0, (size_t)batch * 4, st) != hipSuccess) return DG_EHIP;
        if (class_count && hipMemsetAsync(class_count, 0, (size_t)batch * num_classes * 4, st) != hipSuccess) return DG_EHIP;
        return DG_OK;
    }
    DG_REQUIRE(boxes && scores && labels && keep, "nms_per_class: null pointer");
    DG_REQUIRE((((uintptr_t)boxes) & 15) == 0, "nms_per_class: boxes must be 16-byte aligned");
    DG_REQUIRE(ws && (((uintptr_t)ws) & 255) == 0, "nms_per_class: workspace null or not 256-byte aligned");
    PcWs w = pc_layout(batch, nmax, num_classes, ws);
    if (ws_bytes < w.total) { dg_set_error("nms_per_class: workspace %zu < %zu bytes", ws_bytes, w.total); return DG_ENOMEM; }
    hipLaunchKernelGGL(k_nms_class_bin, dim3((unsigned)batch), dim3(1024), 0, st, (const float4 *)boxes, scores, labels, counts, nmax,
                       (int)num_classes, (float4 *)w.boxes, w.scores, w.orig, w.counts, status);
    DG_CHECK_LAUNCH("k_nms_class_bin");
    int rc = dg_nms_batched_f32(w.boxes, w.scores, w.counts, batch * num_classes, nmax, threshold, quirk_drop_last, w.keep, w.kcnt,
                                w.nms_ws, w.nms_bytes, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_nms_class_unbin, dim3((unsigned)batch), dim3(256), 0, st, w.keep, w.kcnt, w.orig, nmax, (int)num_classes, keep,
                       keep_count, class_count);
    DG_CHECK_LAUNCH("k_nms_class_unbin");
    return DG_OK;
}

// ---------------------------------------------------------------- N2: rcnn/fast_rcnn.py:590-610
// The reference's function called `nms` there suppresses nothing: it orders the rects by np.argsort(scores)[::-1]
// and returns element 0, i.e. the rect of the maximal score on the ORIGINAL float64 scores.  One workgroup: the index
// of the maximum, a NaN ranking above every number (argsort puts NaNs last) and, among equal maxima, the highest index
// (what reversing an ascending order yields when equal keys stay in index order).
__global__ __launch_bounds__(1024) void k_top_score_f64(const double *__restrict__ scores, int64_t n, int64_t *__restrict__ out)
{
    __shared__ double sv[1024];
    __shared__ long long si[1024];
    const int tid = threadIdx.x;
    double bv = 0.0;
    long long bi = -1;
    auto better = [](double v, long long i, double bv, long long bi) {        // (v,i) outranks (bv,bi)
        if (bi < 0) return true;
        const bool vn = v != v, bn = bv != bv;
        if (vn != bn) return vn;
        if (!vn && v != bv) return v > bv;
        return i > bi;
    };
    for (int64_t i = tid; i < n; i += 1024) {
        const double v = scores[i];
        if (better(v, i, bv, bi)) { bv = v; bi = i; }
    }
    sv[tid] = bv; si[tid] = bi;
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if (tid < d && si[tid + d] >= 0 && better(sv[tid + d], si[tid + d], sv[tid], si[tid])) { sv[tid] = sv[tid + d]; si[tid] = si[tid + d]; }
        __syncthreads();
    }
    if (tid == 0) *out = si[0];
}

extern "C" int dg_top_score_f64(const double *scores, int64_t n, int64_t *out_index, dg_stream_t s)
{
    DG_REQUIRE(n >= 1 && scores && out_index, "top_score: need n >= 1 and non-null pointers");
    hipLaunchKernelGGL(k_top_score_f64, dim3(1), dim3(1024), 0, (hipStream_t)s, scores, n, out_index);
    DG_CHECK_LAUNCH("k_top_score_f64");
    return DG_OK;
}
