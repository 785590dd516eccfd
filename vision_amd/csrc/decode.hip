// decode.hip — YOLOv1 decode (batched), YOLOv2 cell encode/decode, R-CNN box regression (gfx950).
#include "common.hpp"

// ---------------------------------------------------------------- D1: yolo/yolo_v1/yolo.py:115-186
// One wave per image: the (7,7,10+C) prediction is staged in LDS with coalesced loads, lane = cell.
// Box corners are computed in float64 and rounded once, as the reference's fp32-tensor + float64-grid
// arithmetic does (:129-131,:137,:144-145,:180).  Ordered compaction (slot-1 hits in row-major cell
// order, then slot-2 hits, :164-178) uses wave ballots + popcounts.
#define Y1_S 7
#define Y1_CELLS 49
#define Y1_MAXCH 96

__device__ __forceinline__ void y1_box(const float *o, int cx, int cy, double *b)
{
    double ltx = (double)(cx * 1) / 7.0, lty = (double)(cy * 1) / 7.0;
    double x = (double)((o[0] * 1.0f) / 7.0f) + ltx;
    double y = (double)((o[1] * 1.0f) / 7.0f) + lty;
    double hw = (double)(o[2] / 2.0f), hh = (double)(o[3] / 2.0f);
    b[0] = x - hw; b[1] = y - hh; b[2] = x + hw; b[3] = y + hh;
}

// Candidates of one image as the NMS of the fused kernel wants them (LDS): list order = the reference's (:164-178).
#define Y1_MAXCAND 128            // 2 * Y1_CELLS = 98, padded to two waves
struct Y1Cand {
    float4 box[Y1_MAXCAND];
    float score[Y1_MAXCAND];
    int label[Y1_MAXCAND];
    int count;
};

// The decode of one image, cells = lanes of ONE wave (`lane` 0..63), from the prediction staged in LDS (`so`).  Writes the
// image's rows of the global outputs and, when `cand` is given, the same candidates into LDS.
__device__ __forceinline__ void y1_cells(const float *so, int lane, int img, int C, float ct, float pt,
                                         float *__restrict__ out_box, int64_t *__restrict__ out_label,
                                         float *__restrict__ out_conf, float *__restrict__ out_score,
                                         int32_t *__restrict__ out_count, Y1Cand *cand)
{
    const int N = 10 + C;
    const bool active = lane < Y1_CELLS;
    const int cell = active ? lane : 0;
    const int cy = cell / Y1_S, cx = cell % Y1_S;
    const float *o = so + cell * N;
    float cs = o[10];
    int cl = 0;
    for (int c = 1; c < C; ++c) {          // torch.max over classes: first maximal index (:124)
        float v = o[10 + c];
        if (v > cs) { cs = v; cl = c; }
    }
    const float conf1 = o[4], conf2 = o[9];
    const bool p1 = active && (conf1 * cs >= pt) && (cs > ct);   // :161
    const bool p2 = active && (conf2 * cs >= pt) && (cs > ct);   // :162
    const u64 b1 = __ballot(p1), b2 = __ballot(p2);
    const u64 lt = (1ull << lane) - 1ull;
    const int c1 = __builtin_popcountll(b1);
    const int64_t ob = (int64_t)img * 2 * Y1_CELLS;
    if (p1 || p2) {
        double bx[4];
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {
            if (slot == 0 ? p1 : p2) {
                int pos = slot == 0 ? __builtin_popcountll(b1 & lt) : c1 + __builtin_popcountll(b2 & lt);
                y1_box(o + 5 * slot, cx, cy, bx);
                float4 f = make_float4((float)bx[0], (float)bx[1], (float)bx[2], (float)bx[3]);
                reinterpret_cast<float4 *>(out_box)[ob + pos] = f;
                out_label[ob + pos] = cl;
                out_conf[ob + pos] = slot == 0 ? conf1 : conf2;
                out_score[ob + pos] = cs;
                if (cand) { cand->box[pos] = f; cand->score[pos] = cs; cand->label[pos] = cl; }
            }
        }
    }
    if (lane == 0) {
        out_count[img] = c1 + __builtin_popcountll(b2);
        if (cand) cand->count = c1 + __builtin_popcountll(b2);
    }
}

// The ground-truth cell of one image (:121,:133-134,:140-152) by ONE wave, from the target staged in LDS (`st`).
__device__ __forceinline__ void y1_target(const float *st, int lane, int img, int C, double *__restrict__ tgt_box,
                                          int64_t *__restrict__ tgt_label)
{
    const int N = 10 + C;
    const bool active = lane < Y1_CELLS;
    const int cell = active ? lane : 0;
    const int cy = cell / Y1_S, cx = cell % Y1_S;
    const float *o = st + cell * N;
    float sum = 0.f;
    for (int c = 0; c < N; ++c) sum += o[c];
    const u64 objm = __ballot(active && sum != 0.f);
    if (objm == 0) {
        if (lane == 0) {
            tgt_label[img] = 0;
            for (int k = 0; k < 4; ++k) tgt_box[4 * img + k] = __longlong_as_double(0x7ff8000000000000LL);
        }
    } else if (lane == __builtin_ctzll(objm)) {
        int best = 0;
        for (int c = 1; c < C; ++c) if (o[10 + c] >= o[10 + best]) best = c;
        tgt_label[img] = best + 1;
        double bx[4];
        y1_box(o, cx, cy, bx);
        for (int k = 0; k < 4; ++k) tgt_box[4 * img + k] = bx[k];
    }
}

__global__ __launch_bounds__(64) void k_yolo1_decode(const float *__restrict__ output, const float *__restrict__ target,
                                                     int C, float ct, float pt,
                                                     float *__restrict__ out_box, int64_t *__restrict__ out_label,
                                                     float *__restrict__ out_conf, float *__restrict__ out_score,
                                                     int32_t *__restrict__ out_count,
                                                     double *__restrict__ tgt_box, int64_t *__restrict__ tgt_label)
{
    __shared__ float so[Y1_CELLS * Y1_MAXCH];
    const int img = blockIdx.x, lane = threadIdx.x;
    const int total = Y1_CELLS * (10 + C);
    const float *g = output + (int64_t)img * total;
    for (int t = lane; t < total; t += 64) so[t] = g[t];
    __syncthreads();
    y1_cells(so, lane, img, C, ct, pt, out_box, out_label, out_conf, out_score, out_count, nullptr);
    if (target) {
        __syncthreads();
        const float *tg = target + (int64_t)img * total;
        for (int t = lane; t < total; t += 64) so[t] = tg[t];
        __syncthreads();
        y1_target(so, lane, img, C, tgt_box, tgt_label);
    }
}

// ---------------------------------------------------------------- decode -> nms of evaluate(), yolo/yolo_v1/yolo.py:281-282, one launch
// One 512-thread workgroup per image; the image's <= 98 candidates never leave LDS between the decode and the NMS
// (config 2 as four launches -- decode, sort, mask, reduce -- is launch-latency-bound: 50 us per batch of 64).
//   decode: wave 0 = the 49 cells of the prediction, wave 1 = the ground-truth cell of the target, at the same time;
//   sort:   thread t ranks candidate t by counting smaller keys (key = class | descending score | list index; class = 0
//           unless per_class, so the order is k_nms_sort_small's);
//   mask:   the M(M-1)/2 pairs j > i dealt evenly to all threads, the EXACT test of k_nms_mask (nms_suppressed, no
//           division); per_class: only boxes of one class suppress each other; bits set by LDS atomics;
//   reduce: wave 0, rows in registers (lane l: rows l and l+64), the serial chain reads them with v_readlane; a class run
//           is a segment: the `numel()>1` quirk (the candidate left alone is dropped, :196) is applied per segment;
//   keep:   kept positions -> list indices by popcount prefix; per_class: class order, then score (dg_nms_batched_per_class_f32).
__device__ __forceinline__ void bits128_range(int a, int b, u64 &lo, u64 &hi)      // bits [a,b) of a 128-bit word
{
    auto below = [](int n) -> u64 { return n <= 0 ? 0ull : (n >= 64 ? ~0ull : ((1ull << n) - 1ull)); };
    lo = below(b) & ~below(a);
    hi = below(b - 64) & ~below(a - 64);
}

__device__ __forceinline__ u64 y1_readlane64(u64 v, int l)      // the builtin returns a signed int: widen through u32
{
    const u32 lo = __builtin_amdgcn_readlane((u32)v, l), hi = __builtin_amdgcn_readlane((u32)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}

#define Y1_FUSED_T 512
template <bool TIE_EVEN>
__global__ __launch_bounds__(Y1_FUSED_T) void k_yolo1_decode_nms(const float *__restrict__ output, const float *__restrict__ target,
                                                                 int C, float ct, float pt, double mid, int per_class, int quirk,
                                                                 float *__restrict__ out_box, int64_t *__restrict__ out_label,
                                                                 float *__restrict__ out_conf, float *__restrict__ out_score,
                                                                 int32_t *__restrict__ out_count, int64_t *__restrict__ keep,
                                                                 int32_t *__restrict__ keep_count, int32_t *__restrict__ class_count,
                                                                 double *__restrict__ tgt_box, int64_t *__restrict__ tgt_label)
{
    __shared__ float so[Y1_CELLS * Y1_MAXCH], st[Y1_CELLS * Y1_MAXCH];
    __shared__ Y1Cand cand;
    __shared__ u64 ckey[Y1_MAXCAND];
    __shared__ float4 sbox[Y1_MAXCAND];
    __shared__ float sarea[Y1_MAXCAND];
    __shared__ int sorder[Y1_MAXCAND], slabel[Y1_MAXCAND];
    __shared__ u64 rowlo[Y1_MAXCAND], rowhi[Y1_MAXCAND];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    {   // prediction and target staged together; wave 0 decodes the cells while wave 1 finds the ground-truth cell
        const int total = Y1_CELLS * (10 + C);
        const float *g = output + (int64_t)img * total;
        for (int t = tid; t < total; t += Y1_FUSED_T) so[t] = g[t];
        if (target) {
            const float *tg = target + (int64_t)img * total;
            for (int t = tid; t < total; t += Y1_FUSED_T) st[t] = tg[t];
        }
        __syncthreads();
        if (tid < 64) y1_cells(so, lane, img, C, ct, pt, out_box, out_label, out_conf, out_score, out_count, &cand);
        else if (tid < 128 && target) y1_target(st, lane, img, C, tgt_box, tgt_label);
    }
    if (class_count && tid < C) class_count[(int64_t)img * C + tid] = 0;
    __syncthreads();
    const int M = cand.count, Mr = (M + 7) & ~7;                     // M <= 98, Mr <= 104: inside the padded arrays

    // ---- sort by rank counting (keys past M are ~0: never below a live key)
    u64 mykey = ~0ull;
    if (tid < M) mykey = ((u64)(per_class ? cand.label[tid] : 0) << 40) | ((u64)dg_score_desc(cand.score[tid]) << 8) | (u64)tid;
    if (tid < Y1_MAXCAND) { ckey[tid] = mykey; rowlo[tid] = 0ull; rowhi[tid] = 0ull; }
    __syncthreads();
    if (tid < M) {
        int rank = 0;
        for (int j = 0; j < Mr; j += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += ckey[j + u] < mykey;
        }
        const float4 b = cand.box[tid];
        sbox[rank] = b;
        sarea[rank] = dg_area(b.x, b.y, b.z, b.w);
        sorder[rank] = tid;
        slabel[rank] = cand.label[tid];
    }
    __syncthreads();

    // ---- suppression rows (bit j of row i, j > i).  A lone wave issues an instruction every ~5 cycles, so the M(M-1)/2
    // pairs are dealt evenly to all 512 threads: group g = rows g and M-1-g (M-1 pairs between them), 8 threads per group;
    // suppressed pairs (few) set their bit with an LDS atomic.
    {
        const int g = tid >> 3, q = tid & 7;
        const int A = g, B = M - 1 - g;
        if (A <= B) {
            const int nA = M - 1 - A, L = nA + (B != A ? g : 0);
            const float4 boxA = sbox[A], boxB = sbox[B];
            const float areaA = sarea[A], areaB = sarea[B];
            const int labA = slabel[A], labB = slabel[B];
            for (int e = q; e < L; e += 8) {
                const bool inA = e < nA;
                const int i = inA ? A : B, j = inA ? A + 1 + e : B + 1 + (e - nA);
                const float4 p = inA ? boxA : boxB, c = sbox[j];
                const float ix = fmaxf(fminf(p.z, c.z) - fmaxf(p.x, c.x), 0.0f);
                const float iy = fmaxf(fminf(p.w, c.w) - fmaxf(p.y, c.y), 0.0f);
                const float inter = ix * iy;
                const float den = (((inA ? areaA : areaB) + sarea[j]) - inter) + 1e-7f;
                bool sup = nms_suppressed<TIE_EVEN>(inter, den, mid);
                if (per_class) sup = sup && (slabel[j] == (inA ? labA : labB));
                if (sup) atomicOr((unsigned long long *)(j < 64 ? &rowlo[i] : &rowhi[i]), 1ull << (j & 63));
            }
        }
    }
    __syncthreads();
    if (tid >= 64) return;

    // ---- serial chain (wave 0): rows in registers (lane l: rows l and l+64), read with v_readlane; the chain hops from
    // one kept box to the next alive one (ctz), first through positions 0..63, then 64..127, inside one segment at a time
    const u64 r0lo = rowlo[tid], r0hi = rowhi[tid], r1hi = rowhi[tid + 64];      // rows >= 64 have no bits below 64
    // per_class: position i starts a class run when its label differs from position i-1's
    u64 startlo = per_class ? __ballot(tid > 0 && tid < M && slabel[tid] != slabel[tid - 1]) : 0ull;
    u64 starthi = per_class ? __ballot(tid + 64 < M && slabel[tid + 64] != slabel[tid + 63]) : 0ull;
    u64 remlo = 0, remhi = 0, keptlo = 0, kepthi = 0;
    int s0 = 0;
    while (s0 < M) {
        // segment [s0, e0): up to the next class start (class-agnostic: everything)
        int e0 = M;
        if (startlo) e0 = __builtin_ctzll(startlo);
        else if (starthi) e0 = 64 + __builtin_ctzll(starthi);
        if (e0 < 64) startlo &= startlo - 1; else if (e0 < M) starthi &= starthi - 1;
        u64 mlo, mhi;
        bits128_range(s0, e0, mlo, mhi);
        u64 blo = 0, bhi = 0;
        int last = -1, total = 0;
        u64 alive = ~remlo & mlo;
        while (alive) {
            const int r = __builtin_ctzll(alive);
            keptlo |= 1ull << r;
            blo = remlo; bhi = remhi; last = r; total += 1;
            remlo |= y1_readlane64(r0lo, r);
            remhi |= y1_readlane64(r0hi, r);
            alive = ~remlo & mlo & ~((2ull << r) - 1ull);
        }
        alive = ~remhi & mhi;
        while (alive) {
            const int r = __builtin_ctzll(alive);
            kepthi |= 1ull << r;
            blo = remlo; bhi = remhi; last = 64 + r; total += 1;
            remhi |= y1_readlane64(r1hi, r);
            alive = ~remhi & mhi & ~((2ull << r) - 1ull);
        }
        if (quirk && last >= 0) {
            // yolo.py:196 `while ids_sorted.numel()>1`: the segment's last kept box is dropped iff nothing ranked below it
            // (in its segment) was still alive when it became the head, i.e. before its own row was applied
            u64 alo, ahi;
            bits128_range(last + 1, e0, alo, ahi);
            if (((~blo & alo) | (~bhi & ahi)) == 0ull) {
                if (last < 64) keptlo &= ~(1ull << last);
                else kepthi &= ~(1ull << (last - 64));
                total -= 1;
            }
        }
        if (class_count && per_class && tid == 0 && total > 0) class_count[(int64_t)img * C + slabel[s0]] = total;
        s0 = e0;
    }

    // ---- keep list: sorted position -> index in the image's candidate list
    int64_t *kp = keep + (int64_t)img * (2 * Y1_CELLS);
    const u64 below = (1ull << tid) - 1ull;
    const int nlo = __builtin_popcountll(keptlo);
    if ((keptlo >> tid) & 1ull) kp[__builtin_popcountll(keptlo & below)] = (int64_t)sorder[tid];
    if ((kepthi >> tid) & 1ull) kp[nlo + __builtin_popcountll(kepthi & below)] = (int64_t)sorder[tid + 64];
    if (tid == 0) keep_count[img] = nlo + __builtin_popcountll(kepthi);
}

extern "C" int dg_yolo1_decode_f32(const float *output, const float *target, int64_t batch, int64_t num_classes,
                                   float conf_threshold, float prob_threshold,
                                   float *out_box, int64_t *out_label, float *out_conf, float *out_score, int32_t *out_count,
                                   double *tgt_box, int64_t *tgt_label, dg_stream_t s)
{
    DG_REQUIRE(batch >= 0, "yolo1_decode: batch < 0");
    if (batch == 0) return DG_OK;
    DG_REQUIRE(num_classes >= 1 && 10 + num_classes <= Y1_MAXCH, "yolo1_decode: num_classes must be in [1,%d]", Y1_MAXCH - 10);
    DG_REQUIRE(output && out_box && out_label && out_conf && out_score && out_count, "yolo1_decode: null pointer");
    DG_REQUIRE(!target || (tgt_box && tgt_label), "yolo1_decode: target given without tgt outputs");
    DG_REQUIRE((((uintptr_t)out_box) & 15) == 0, "yolo1_decode: out_box must be 16-byte aligned");
    hipLaunchKernelGGL(k_yolo1_decode, dim3((unsigned)batch), dim3(64), 0, (hipStream_t)s, output, target, (int)num_classes,
                       conf_threshold, prob_threshold, out_box, out_label, out_conf, out_score, out_count, tgt_box, tgt_label);
    DG_CHECK_LAUNCH("k_yolo1_decode");
    return DG_OK;
}

extern "C" int dg_yolo1_decode_nms_f32(const float *output, const float *target, int64_t batch, int64_t num_classes,
                                       float conf_threshold, float prob_threshold, float nms_threshold, int per_class,
                                       int quirk_drop_last, float *out_box, int64_t *out_label, float *out_conf, float *out_score,
                                       int32_t *out_count, int64_t *keep, int32_t *keep_count, int32_t *class_count,
                                       double *tgt_box, int64_t *tgt_label, dg_stream_t s)
{
    DG_REQUIRE(batch >= 0, "yolo1_decode_nms: batch < 0");
    if (batch == 0) return DG_OK;
    DG_REQUIRE(num_classes >= 1 && 10 + num_classes <= Y1_MAXCH, "yolo1_decode_nms: num_classes must be in [1,%d]", Y1_MAXCH - 10);
    DG_REQUIRE(output && out_box && out_label && out_conf && out_score && out_count && keep && keep_count,
               "yolo1_decode_nms: null pointer");
    DG_REQUIRE(!target || (tgt_box && tgt_label), "yolo1_decode_nms: target given without tgt outputs");
    DG_REQUIRE((((uintptr_t)out_box) & 15) == 0, "yolo1_decode_nms: out_box must be 16-byte aligned");
    double mid;
    const bool tie_even = dg_nms_threshold_mid(nms_threshold, &mid);
    if (tie_even)
        hipLaunchKernelGGL(k_yolo1_decode_nms<true>, dim3((unsigned)batch), dim3(Y1_FUSED_T), 0, (hipStream_t)s, output, target,
                           (int)num_classes, conf_threshold, prob_threshold, mid, per_class != 0, quirk_drop_last != 0, out_box,
                           out_label, out_conf, out_score, out_count, keep, keep_count, class_count, tgt_box, tgt_label);
    else
        hipLaunchKernelGGL(k_yolo1_decode_nms<false>, dim3((unsigned)batch), dim3(Y1_FUSED_T), 0, (hipStream_t)s, output, target,
                           (int)num_classes, conf_threshold, prob_threshold, mid, per_class != 0, quirk_drop_last != 0, out_box,
                           out_label, out_conf, out_score, out_count, keep, keep_count, class_count, tgt_box, tgt_label);
    DG_CHECK_LAUNCH("k_yolo1_decode_nms");
    return DG_OK;
}

// ---------------------------------------------------------------- YOLOv2: yolo/yolo_v2/utils.py:228-234
// torch's float floor_divide / remainder (fmod based), restated for fp32.
__device__ __forceinline__ float t_remainder(float a, float b)
{
    float m = fmodf(a, b);
    if (m != 0.f && ((b < 0.f) != (m < 0.f))) m += b;
    return m;
}
__device__ __forceinline__ float t_floordiv(float a, float b)
{
    float m = fmodf(a, b);
    float d = (a - m) / b;
    if (m != 0.f && ((b < 0.f) != (m < 0.f))) d -= 1.f;
    float f;
    if (d != 0.f) { f = floorf(d); if (d - f > 0.5f) f += 1.f; }
    else f = copysignf(0.f, a / b);
    return f;
}

__global__ __launch_bounds__(256) void k_yolo2_get_point(const float4 *__restrict__ bbox, int64_t n,
                                                         int64_t *__restrict__ points, float4 *__restrict__ enc)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float cell = (float)(1.0 / 13.0);
    float4 b = bbox[i];
    float cx = b.x + b.z / 2.0f, cy = b.y + b.w / 2.0f;
    points[2 * i] = (int64_t)t_floordiv(cx, cell);
    points[2 * i + 1] = (int64_t)t_floordiv(cy, cell);
    enc[i] = make_float4(t_remainder(cx, cell) / cell, t_remainder(cy, cell) / cell, b.z, b.w);
}

__global__ __launch_bounds__(256) void k_yolo2_decode_cells(const int64_t *__restrict__ points, const float4 *__restrict__ enc,
                                                            int64_t n, float4 *__restrict__ xyxy)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 e = enc[i];
    float cx = ((float)points[2 * i] + e.x) / 13.0f, cy = ((float)points[2 * i + 1] + e.y) / 13.0f;
    float hw = e.z / 2.0f, hh = e.w / 2.0f;
    xyxy[i] = make_float4(cx - hw, cy - hh, cx + hw, cy + hh);
}

__global__ __launch_bounds__(256) void k_yolo2_decode_anchor(const float *__restrict__ pred, int64_t G, int64_t A, int64_t C,
                                                             const float *__restrict__ anchors, float4 *__restrict__ xyxy,
                                                             float *__restrict__ conf)
{
    int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= G * G * A) return;
    const int64_t a = k % A, gx = (k / A) % G, gy = k / (A * G);
    const float *p = pred + k * (5 + C);
    double sx = 1.0 / (1.0 + exp(-(double)p[0])), sy = 1.0 / (1.0 + exp(-(double)p[1]));
    double bx = (sx + (double)gx) / (double)G, by = (sy + (double)gy) / (double)G;
    double bw = (double)anchors[2 * a] * exp((double)p[2]), bh = (double)anchors[2 * a + 1] * exp((double)p[3]);
    xyxy[k] = make_float4((float)(bx - bw / 2), (float)(by - bh / 2), (float)(bx + bw / 2), (float)(by + bh / 2));
    conf[k] = (float)(1.0 / (1.0 + exp(-(double)p[4])));
}

extern "C" int dg_yolo2_get_point_f32(const float *bbox, int64_t n, int64_t *points, float *enc, dg_stream_t s)
{
    DG_REQUIRE(n >= 0, "yolo2_get_point: n < 0");
    if (n == 0) return DG_OK;
    DG_REQUIRE(bbox && points && enc, "yolo2_get_point: null pointer");
    DG_REQUIRE((((uintptr_t)bbox | (uintptr_t)enc) & 15) == 0, "yolo2_get_point: 16-byte alignment required");
    hipLaunchKernelGGL(k_yolo2_get_point, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, (const float4 *)bbox, n,
                       points, (float4 *)enc);
    DG_CHECK_LAUNCH("k_yolo2_get_point");
    return DG_OK;
}

extern "C" int dg_yolo2_decode_cells_f32(const int64_t *points, const float *enc, int64_t n, float *xyxy, dg_stream_t s)
{
    DG_REQUIRE(n >= 0, "yolo2_decode_cells: n < 0");
    if (n == 0) return DG_OK;
    DG_REQUIRE(points && enc && xyxy, "yolo2_decode_cells: null pointer");
    DG_REQUIRE((((uintptr_t)xyxy | (uintptr_t)enc) & 15) == 0, "yolo2_decode_cells: 16-byte alignment required");
    hipLaunchKernelGGL(k_yolo2_decode_cells, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, points,
                       (const float4 *)enc, n, (float4 *)xyxy);
    DG_CHECK_LAUNCH("k_yolo2_decode_cells");
    return DG_OK;
}

extern "C" int dg_yolo2_decode_anchor_f32(const float *pred, int64_t G, int64_t A, int64_t C, const float *anchors,
                                          float *xyxy, float *conf, dg_stream_t s)
{
    DG_REQUIRE(G > 0 && A > 0 && C >= 0, "yolo2_decode_anchor: bad shape");
    DG_REQUIRE(pred && anchors && xyxy && conf, "yolo2_decode_anchor: null pointer");
    DG_REQUIRE((((uintptr_t)xyxy) & 15) == 0, "yolo2_decode_anchor: 16-byte alignment required");
    hipLaunchKernelGGL(k_yolo2_decode_anchor, dim3((unsigned)dg_cdiv(G * G * A, 256)), dim3(256), 0, (hipStream_t)s, pred, G, A, C,
                       anchors, (float4 *)xyxy, conf);
    DG_CHECK_LAUNCH("k_yolo2_decode_anchor");
    return DG_OK;
}

// ---------------------------------------------------------------- R-CNN box regression
// calcul: rcnn/utils.py:153-171; inverse: rcnn/rcnn.py:287-310 (variant 0), rcnn/fast_rcnn.py:543-560 (variant 1)
__global__ __launch_bounds__(256) void k_rcnn_calcul(const float4 *__restrict__ pos, int64_t n, const float *__restrict__ grt,
                                                     float4 *__restrict__ t)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float g_w = grt[2] - grt[0], g_h = grt[3] - grt[1];
    float g_x = grt[0] + g_w / 2, g_y = grt[1] + g_h / 2;
    float4 p = pos[i];
    float p_w = p.z - p.x, p_h = p.w - p.y;
    float p_x = p.x + p_w / 2, p_y = p.y + p_h / 2;
    t[i] = make_float4((g_x - p_x) / p_w, (g_y - p_y) / p_h, logf(g_w / p_w), logf(g_h / p_h));
}

__global__ __launch_bounds__(256) void k_rcnn_decode(const int4 *__restrict__ p_rect, const float4 *__restrict__ tset, int64_t n,
                                                     int variant, float4 *__restrict__ g)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int4 p = p_rect[i];
    float4 t = tset[i];
    float pw = (float)(p.z - p.x), ph = (float)(p.w - p.y);
    float px, py;
    if (variant == 0) { px = (float)p.x + pw / 2; py = (float)p.y + ph / 2; }
    else { px = (float)(p.z + p.x) / 2; py = (float)(p.w + p.y) / 2; }
    float gx = pw * t.x + px, gy = ph * t.y + py;
    float gw = pw * expf(t.z), gh = ph * expf(t.w);
    g[i] = make_float4(gx - gw / 2, gy - gh / 2, gx + gw / 2, gy + gh / 2);
}

extern "C" int dg_rcnn_calcul_f32(const float *positive, int64_t n, const float *grt, float *t, dg_stream_t s)
{
    DG_REQUIRE(n >= 0, "rcnn_calcul: n < 0");
    if (n == 0) return DG_OK;
    DG_REQUIRE(positive && grt && t, "rcnn_calcul: null pointer");
    DG_REQUIRE((((uintptr_t)positive | (uintptr_t)t) & 15) == 0, "rcnn_calcul: 16-byte alignment required");
    hipLaunchKernelGGL(k_rcnn_calcul, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, (const float4 *)positive, n, grt,
                       (float4 *)t);
    DG_CHECK_LAUNCH("k_rcnn_calcul");
    return DG_OK;
}

extern "C" int dg_rcnn_decode_f32(const int32_t *p_rect, const float *tset, int64_t n, int variant, float *g, dg_stream_t s)
{
    DG_REQUIRE(n >= 0, "rcnn_decode: n < 0");
    if (n == 0) return DG_OK;
    DG_REQUIRE(p_rect && tset && g, "rcnn_decode: null pointer");
    DG_REQUIRE((((uintptr_t)p_rect | (uintptr_t)tset | (uintptr_t)g) & 15) == 0, "rcnn_decode: 16-byte alignment required");
    hipLaunchKernelGGL(k_rcnn_decode, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, (const int4 *)p_rect,
                       (const float4 *)tset, n, variant, (float4 *)g);
    DG_CHECK_LAUNCH("k_rcnn_decode");
    return DG_OK;
}

// ---------------------------------------------------------------- proposal labelling: the quota walk of do_ss ("next" row 4)
// rcnn/rcnn.py:35-65 (rcnn/fast_rcnn.py:123-155): proposals are visited in order; one with iou >= pos_thr is taken as a
// positive while fewer than min_num positives have been taken, one with iou < neg_thr as a negative while fewer than
// quota_neg have; the first visit that finds a quota already full raises that quota's flag, and the walk stops after the
// visit at which both flags are up.  As a scan: with cp[e] / cn[e] the numbers of candidates before e, e is taken as a
// positive iff cp[e] < min_num, the positive flag goes up at the first e with cp[e] >= min_num (likewise cn / quota_neg),
// the last visit is e* = the later of the two (n-1 if a flag never goes up), and the output slot of a selection is the
// number of selections before it.  One 1024-thread workgroup, two passes over the ious (ballot + wave prefix).
// out[0] = selections, out[1] = status: bit 0 a visited proposal carries an assert flag (rcnn/utils.py:51-54,65-66),
// bit 1 / bit 2 the positive / negative quota flag went up.
#define LQ_T 1024
__global__ __launch_bounds__(LQ_T) void k_rcnn_label_quota(const double *__restrict__ iou, const int32_t *__restrict__ flags, int64_t n,
                                                           long long min_num, long long quota_neg, double pos_thr, double neg_thr,
                                                           int32_t label, int32_t *__restrict__ sel, int32_t *__restrict__ lab,
                                                           int32_t *__restrict__ out)
{
    __shared__ int wcnt[2][LQ_T / 64];
    __shared__ long long s_ef, s_eb;
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_ef = n; s_eb = n; s_bad = 0; }           // n = "never"
    __syncthreads();
    const u64 below = (1ull << lane) - 1ull;
    // one chunk of the scan: candidates before element e inside the chunk (exclusive), chunk totals in tp / tn
    auto chunk_scan = [&](int64_t e, bool &p, bool &q, long long &cpe, long long &cne, int &tp, int &tn) {
        const double v = (e < n) ? iou[e] : 0.0;
        p = e < n && v >= pos_thr;
        q = e < n && v < neg_thr;
        const u64 bp = __ballot(p), bq = __ballot(q);
        if (lane == 0) { wcnt[0][wave] = __builtin_popcountll(bp); wcnt[1][wave] = __builtin_popcountll(bq); }
        __syncthreads();
        int wp = 0, wq = 0;
        tp = 0; tn = 0;
#pragma unroll
        for (int w = 0; w < LQ_T / 64; ++w) {
            const int a = wcnt[0][w], b = wcnt[1][w];
            if (w < wave) { wp += a; wq += b; }
            tp += a; tn += b;
        }
        __syncthreads();
        cpe = wp + __builtin_popcountll(bp & below);
        cne = wq + __builtin_popcountll(bq & below);
    };
    // pass 1: where the two flags go up
    long long base_p = 0, base_n = 0;
    for (int64_t c0 = 0; c0 < n; c0 += LQ_T) {
        bool p, q; long long cpe, cne; int tp, tn;
        const int64_t e = c0 + tid;
        chunk_scan(e, p, q, cpe, cne, tp, tn);
        if (e < n) {
            const u64 f = __ballot(base_p + cpe >= min_num), b = __ballot(base_n + cne >= quota_neg);
            if (f && lane == __builtin_ctzll(f)) atomicMin((unsigned long long *)&s_ef, (unsigned long long)e);
            if (b && lane == __builtin_ctzll(b)) atomicMin((unsigned long long *)&s_eb, (unsigned long long)e);
        }
        base_p += tp; base_n += tn;
        __syncthreads();
        if (s_ef < n && s_eb < n) break;                      // uniform: both known, later chunks are never visited
    }
    __syncthreads();
    const long long ef = s_ef, eb = s_eb;
    const int64_t elast = (ef < n && eb < n) ? (ef > eb ? ef : eb) : n - 1;
    // pass 2: the selections of the visited proposals
    base_p = 0; base_n = 0;
    long long total = 0;
    for (int64_t c0 = 0; c0 <= elast; c0 += LQ_T) {
        bool p, q; long long cpe, cne; int tp, tn;
        const int64_t e = c0 + tid;
        chunk_scan(e, p, q, cpe, cne, tp, tn);
        if (e <= elast) {
            const long long cp = base_p + cpe, cn = base_n + cne;
            if (flags[e] != 0) s_bad = 1;
            const bool tp_ = p && cp < min_num, tn_ = q && cn < quota_neg;
            const long long slot = (cp < min_num ? cp : min_num) + (cn < quota_neg ? cn : quota_neg);
            if (tp_) { sel[slot] = (int32_t)e; lab[slot] = label; }
            if (tn_) { sel[slot + (tp_ ? 1 : 0)] = (int32_t)e; lab[slot + (tp_ ? 1 : 0)] = 0; }
            if (e == elast) total = slot + (tp_ ? 1 : 0) + (tn_ ? 1 : 0);
        }
        base_p += tp; base_n += tn;
    }
    __syncthreads();
    if (n > 0 && (int64_t)tid == (elast % LQ_T)) {
        out[0] = (int32_t)total;
        out[1] = (s_bad ? 1 : 0) | (ef < n ? 2 : 0) | (eb < n ? 4 : 0);
    }
    if (n == 0 && tid == 0) { out[0] = 0; out[1] = 0; }
}

extern "C" int dg_rcnn_label_quota_f64(const double *iou, const int32_t *flags, int64_t n, int64_t min_num, int64_t quota_neg,
                                       double pos_thr, double neg_thr, int32_t label, int32_t *sel, int32_t *lab, int32_t *out,
                                       dg_stream_t s)
{
    DG_REQUIRE(n >= 0 && n <= 0x7fffffff && min_num >= 0 && quota_neg >= 0, "rcnn_label_quota: bad size");
    DG_REQUIRE(out && (n == 0 || (iou && flags && sel && lab)), "rcnn_label_quota: null pointer");
    hipLaunchKernelGGL(k_rcnn_label_quota, dim3(1), dim3(LQ_T), 0, (hipStream_t)s, iou, flags, n, (long long)min_num, (long long)quota_neg,
                       pos_thr, neg_thr, label, sel, lab, out);
    DG_CHECK_LAUNCH("k_rcnn_label_quota");
    return DG_OK;
}

// ---------------------------------------------------------------- YOLOv1 target encoder ("next" row 4)
// encode(bb,label,num_classes): yolo/yolo_v1/modules.py:151-174, float64 like NumPy.  bb = {left, top, w, h} relative
// to the image; the object's cell gets [x_cell, y_cell, w, h, 1] in both box slots and a one-hot class.
// One thread per box writes its cell (the target was zeroed by the caller's memset).  status: bit 0 = a cell index
// outside [-7,7) (NumPy raises IndexError), bit 1 = label outside 1..C.
__global__ __launch_bounds__(256) void k_yolo1_encode(const double *__restrict__ bb, const int64_t *__restrict__ label, int64_t n,
                                                      int C, double *__restrict__ target, int32_t *__restrict__ status)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    const int D = 10 + C;
    const double cell = 1.0 / 7.0;
    const double w = bb[4 * b + 2], h = bb[4 * b + 3];
    const double cx = bb[4 * b] + w / 2, cy = bb[4 * b + 1] + h / 2;
    int i = (int)(cx * 7), j = (int)(cy * 7);              // astype(int): truncation
    const double xc = (cx - (double)i * cell) / cell, yc = (cy - (double)j * cell) / cell;
    const int64_t lab = label[b];
    int bad = 0;
    if (i < -7 || i >= 7 || j < -7 || j >= 7) bad |= 1;
    if (lab < 1 || lab > C) bad |= 2;
    if (bad) { atomicOr(status, bad); return; }
    if (i < 0) i += 7;                                       // NumPy negative index
    if (j < 0) j += 7;
    double *t = target + ((b * 7 + j) * 7 + i) * D;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) { t[5 * s2] = xc; t[5 * s2 + 1] = yc; t[5 * s2 + 2] = w; t[5 * s2 + 3] = h; t[5 * s2 + 4] = 1.0; }
    t[10 + lab - 1] = 1.0;
}

extern "C" int dg_yolo1_encode_f64(const double *bb, const int64_t *label, int64_t n, int64_t num_classes, double *target,
                                   int32_t *status, dg_stream_t s)
{
    DG_REQUIRE(n >= 0 && num_classes >= 1 && num_classes <= 4096, "yolo1_encode: bad shape");
    if (n == 0) return DG_OK;
    DG_REQUIRE(bb && label && target && status, "yolo1_encode: null pointer");
    hipStream_t st = (hipStream_t)s;
    if (hipMemsetAsync(target, 0, (size_t)n * 49 * (10 + num_classes) * sizeof(double), st) != hipSuccess) {
        dg_set_error("yolo1_encode: memset failed");
        return DG_EHIP;
    }
    hipLaunchKernelGGL(k_yolo1_encode, dim3((unsigned)dg_cdiv(n, 256)), dim3(256), 0, st, bb, label, n, (int)num_classes, target, status);
    DG_CHECK_LAUNCH("k_yolo1_encode");
    return DG_OK;
}
