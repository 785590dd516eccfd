/* asan_selftest.c -- memory-safety run of the CPU oracle (TEST INFRASTRUCTURE, like oracle.c itself).
 * Built by `make -C oracle asan-run` with -fsanitize=address,undefined: every buffer handed to an oracle function is a
 * malloc of EXACTLY the size the function's contract names, so any read or write outside it traps; shapes include the
 * edge cases the parity tests use (n = 0, n = 1, ragged counts, crops of one cell, odd channel counts).  Values are not
 * checked here -- tests/test_oracle_golden.py pins them against the reference's own outputs -- only that the restatement
 * stays inside its buffers and free of undefined behaviour. */
#include "oracle.c"
#include <stdio.h>

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static double urand(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) / 9007199254740992.0; }
static float *fbuf(size_t n) { float *p = malloc(n * sizeof(float) + (n == 0)); for (size_t i = 0; i < n; ++i) p[i] = (float)urand(); return p; }
static double *dbuf(size_t n) { double *p = malloc(n * sizeof(double) + (n == 0)); for (size_t i = 0; i < n; ++i) p[i] = urand(); return p; }
static void boxes_f(float *b, size_t n) { for (size_t i = 0; i < n; ++i) { b[4*i] = (float)(urand()*0.7); b[4*i+1] = (float)(urand()*0.7); b[4*i+2] = b[4*i] + (float)(0.01 + urand()*0.3); b[4*i+3] = b[4*i+1] + (float)(0.01 + urand()*0.3); } }

int main(void)
{
    static const int64_t sizes[] = {0, 1, 2, 3, 63, 64, 65, 500, 2000};
    for (size_t si = 0; si < sizeof(sizes) / sizeof(sizes[0]); ++si) {
        const int64_t n = sizes[si];
        float *a = fbuf(4 * n), *b = fbuf(4 * n), *o = fbuf(n), *sc = fbuf(n);
        boxes_f(a, n); boxes_f(b, n);
        orc_iou_elementwise_f32(a, b, n, 1e-7f, o);
        double *ad = dbuf(4 * n), *bd = dbuf(4 * n), *od = dbuf(n);
        orc_iou_elementwise_f64(ad, bd, n, 1e-7, od);
        const int64_t m = n > 300 ? 300 : n;
        float *pw = fbuf((size_t)m * n);
        orc_iou_pairwise_f32(a, m, b, n, 1e-7f, pw);
        for (int q = 0; q < 2; ++q)
            for (int t = 0; t < 3; ++t) {
                int64_t *keep = malloc(sizeof(int64_t) * (size_t)n + (n == 0)), nk = -1;
                if (orc_nms_f32(a, sc, n, t == 0 ? 0.1 : (t == 1 ? 0.5 : 0.9), q, keep, &nk) != ORC_OK || nk < 0 || nk > n) return 2;
                free(keep);
            }
        int64_t *pts = malloc(sizeof(int64_t) * 2 * (size_t)n + (n == 0));
        float *enc = fbuf(4 * n), *xy = fbuf(4 * n);
        orc_yolo2_get_point(a, n, pts, enc);
        orc_yolo2_decode_cells(pts, enc, n, xy);
        float grt[4] = {0.2f, 0.2f, 0.7f, 0.8f}, *tt = fbuf(4 * n), *gg = fbuf(4 * n);
        orc_rcnn_calcul_f32(a, n, grt, tt);
        int32_t *pr = malloc(sizeof(int32_t) * 4 * (size_t)n + (n == 0));
        for (int64_t i = 0; i < 4 * n; ++i) pr[i] = (int32_t)(urand() * 200) + 1;
        orc_rcnn_decode_f32(pr, tt, n, 0, gg); orc_rcnn_decode_f32(pr, tt, n, 1, gg);
        /* k-means: wh (n,2), k = 5 and 16 */
        for (int k = 5; k <= 16; k += 11) {
            float *wh = fbuf(2 * n), *cent = fbuf(2 * (size_t)k);
            uint8_t *lab = malloc((size_t)n + (n == 0)); memset(lab, 255, (size_t)n);
            int64_t *acc = calloc((size_t)(3 * k + 2), sizeof(int64_t));
            orc_kmeans_step(wh, n, cent, k, lab, acc); orc_kmeans_update(acc, k, cent);
            free(wh); free(cent); free(lab); free(acc);
        }
        free(a); free(b); free(o); free(sc); free(ad); free(bd); free(od); free(pw); free(pts); free(enc); free(xy); free(tt); free(gg); free(pr);
    }
    {   /* scalar IoU incl. the assert paths */
        double p[4] = {0, 0, 1, 1}, q[4] = {0.5, 0.5, 2, 2}, bad[4] = {1, 1, 0, 0}, out;
        if (orc_iou_scalar_f64(p, q, &out) != ORC_OK || orc_iou_scalar_f64(bad, q, &out) != ORC_EASSERT) return 3;
    }
    for (int C = 1; C <= 20; C += 19) {   /* YOLOv1 decode and loss, C = 1 and 20 */
        const int D = 10 + C;
        float *out = fbuf(49 * (size_t)D), *tgt = fbuf(49 * (size_t)D);
        float *ob = fbuf(98 * 4), *oc = fbuf(98), *os = fbuf(98);
        int64_t *ol = malloc(sizeof(int64_t) * 98), lt;
        double bt[4];
        if (orc_yolo1_decode(out, tgt, C, 0.2, 0.01, ob, ol, oc, os, bt, &lt) < 0) return 4;
        if (orc_yolo1_decode(out, NULL, C, 0.99, 0.99, ob, ol, oc, os, bt, &lt) < 0) return 4;
        const int64_t nimg = 3;
        float *y = fbuf((size_t)nimg * 49 * D), *t = fbuf((size_t)nimg * 49 * D), loss, *grad = fbuf((size_t)nimg * 49 * D);
        for (int64_t i = 0; i < nimg * 49; ++i) t[i * D + 4] = (i % 11 == 0) ? 1.f : 0.f;
        double comp[5];
        orc_yolo1_loss(y, t, nimg, C, 1, comp, &loss, grad); orc_yolo1_loss(y, t, nimg, C, 0, comp, &loss, NULL);
        free(out); free(tgt); free(ob); free(oc); free(os); free(ol); free(y); free(t); free(grad);
    }
    {   /* anchor decode */
        const int64_t G = 13, A = 5, C = 3;
        float *pred = fbuf((size_t)(G * G * A * (5 + C))), *anc = fbuf(2 * A), *xy = fbuf((size_t)(G * G * A * 4)), *cf = fbuf((size_t)(G * G * A));
        orc_yolo2_decode_anchor(pred, G, A, C, anc, xy, cf);
        free(pred); free(anc); free(xy); free(cf);
    }
    {   /* RoI pooling: odd channel count, crops from one cell to the whole map, both modes, several output sizes */
        const int64_t B = 2, C = 5; const int H = 9, W = 13; const int64_t n = 40;
        float *img = fbuf((size_t)(B * C * H * W));
        double *rois = dbuf(4 * n); int64_t *idx = malloc(sizeof(int64_t) * n);
        for (int64_t i = 0; i < n; ++i) {
            double x = urand() * 0.9, y = urand() * 0.9, w = (i % 5 == 0) ? 1e-3 : urand() * (1 - x), h = (i % 7 == 0) ? 1e-3 : urand() * (1 - y);
            rois[4*i] = x; rois[4*i+1] = y; rois[4*i+2] = x + w; rois[4*i+3] = y + h; idx[i] = i % B;
        }
        rois[0] = 0; rois[1] = 0; rois[2] = 1; rois[3] = 1;
        static const int ps[][2] = {{7, 7}, {1, 1}, {3, 5}, {14, 14}, {2, 9}};
        for (size_t k = 0; k < sizeof(ps) / sizeof(ps[0]); ++k)
            for (int mode = 0; mode < 2; ++mode) {
                const size_t e = (size_t)n * C * ps[k][0] * ps[k][1];
                float *out = fbuf(e), *go = fbuf(e), *gi = fbuf((size_t)(B * C * H * W));
                int32_t *am = malloc(sizeof(int32_t) * e);
                if (orc_roipool_fwd2_f32(img, B, C, H, W, rois, idx, n, ps[k][0], ps[k][1], mode, out, mode ? am : NULL) != ORC_OK) return 5;
                if (orc_roipool_bwd2_f32(go, mode ? am : NULL, B, C, H, W, rois, idx, n, ps[k][0], ps[k][1], mode, gi) != ORC_OK) return 5;
                free(out); free(go); free(gi); free(am);
            }
        double empty[4] = {0.0, 0.0, 0.0, 0.0}; int64_t i0 = 0; float o1[5 * 49]; int32_t a1[5 * 49];
        if (orc_roipool_fwd_f32(img, B, C, H, W, empty, &i0, 1, 7, 1, o1, a1) != ORC_EEMPTYROI) return 6;
        free(img); free(rois); free(idx);
    }
    puts("oracle asan/ubsan self-test: clean");
    return 0;
}
