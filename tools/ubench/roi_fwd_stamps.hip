// Diagnostic build of the RoI forward kernel with in-kernel cycle stamps (never shipped).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -DDG_STAMPS -I. tools/ubench/roi_fwd_stamps.hip -o tools/ubench/roi_fwd_stamps
#include "../../vision_amd/csrc/roipool.hip"
#include "../../vision_amd/csrc/abi.hip"
#include <vector>
#include <random>
int main(int argc, char **argv)
{
    const int images = argc > 1 ? atoi(argv[1]) : 8, C = 512, H = 38, W = 50, n = images * 2000;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> u(0, 1);
    std::vector<double> rois(4 * (size_t)n);
    std::vector<int64_t> idx(n);
    for (int i = 0; i < n; ++i) {
        const double lo = argc > 3 ? atof(argv[2]) : 0.02, hi = argc > 3 ? atof(argv[3]) : 0.30;      // RoI side as a fraction of the map
        double w = lo + (hi - lo) * u(rng), h = lo + (hi - lo) * u(rng), x = u(rng) * (argc > 3 ? 1.0 - w : 0.7), y = u(rng) * (argc > 3 ? 1.0 - h : 0.7);
        rois[4 * i] = x; rois[4 * i + 1] = y; rois[4 * i + 2] = x + w; rois[4 * i + 3] = y + h;
        idx[i] = i / 2000;
    }
    std::vector<float> img((size_t)images * C * H * W);
    for (auto &v : img) v = (float)u(rng);
    double *d_rois; int64_t *d_idx; float *d_img, *d_nhwc, *d_out; int32_t *d_win, *d_st, *d_arg, *d_nan;
    hipMalloc(&d_rois, rois.size() * 8); hipMalloc(&d_idx, n * 8); hipMalloc(&d_img, img.size() * 4); hipMalloc(&d_nhwc, img.size() * 4);
    hipMalloc(&d_out, (size_t)n * C * 49 * 4); hipMalloc(&d_arg, (size_t)n * C * 49 * 4); hipMalloc(&d_win, n * 20); hipMalloc(&d_st, 4); hipMalloc(&d_nan, 4);
    hipMemcpy(d_rois, rois.data(), rois.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_idx, idx.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice); hipMemset(d_st, 0, 4); hipMemset(d_nan, 0, 4);
    dg_roi_windows(d_rois, d_idx, n, images, H, W, d_win, d_st, 0);
    dg_nchw_to_nhwc_f32(d_img, images, C, H, W, d_nhwc, d_nan, 0);
    for (int it = 0; it < 3; ++it) {
        unsigned long long z[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(dg_stamp_acc), z, sizeof(z));
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        int rc = dg_roipool_fwd_f32(d_nhwc, images, C, H, W, d_win, n, 7, 1, d_nan, d_out, d_arg, 0);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpyFromSymbol(z, HIP_SYMBOL(dg_stamp_acc), sizeof(z));
        double tot = (double)z[6];
        printf("rc=%d %.3f ms | issue %.1f%% dma-wait %.1f%% rows %.1f%% park %.1f%% | staged-total %.1f%% (stage-write %.1f%%) out-phase %.1f%% | cycles/wave %.0f\n", rc, ms,
               100 * z[0] / tot, 100 * z[1] / tot, 100 * z[2] / tot, 100 * z[3] / tot, 100 * (z[4] - z[5]) / tot, 100 * z[7] / tot, 100 * z[5] / tot, tot / 2048);
    }
    return 0;
}
