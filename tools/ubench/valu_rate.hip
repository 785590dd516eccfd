// micro-benchmark: VALU issue rate of the compare/select patterns used by the pooling and NMS kernels
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
template <int PAT>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters)
{
    float b[8], v[8];
    int c[8];
    for (int i = 0; i < 8; ++i) { b[i] = in[threadIdx.x + i]; v[i] = in[threadIdx.x + 8 + i]; c[i] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (PAT == 0) { b[i] = __builtin_fmaf(b[i], v[i], v[i]); }                         // 1 VALU
                if (PAT == 1) { bool g = v[i] > b[i]; b[i] = g ? v[i] : b[i]; c[i] = g ? r : c[i]; v[i] += 1.0f; }   // cmp + 2 cndmask + add
                if (PAT == 2) { c[i] = (v[i] > b[i]) ? r : c[i]; b[i] = fmaxf(b[i], v[i]); v[i] += 1.0f; }           // cmp + cndmask + max + add
                if (PAT == 3) { b[i] = fmaxf(b[i], v[i]); v[i] += 1.0f; }                           // max + add
                if (PAT == 4) { bool g = v[i] > b[i]; b[i] = g ? v[i] : b[i]; v[i] += 1.0f; }       // cmp + cndmask + add
            }
        }
    }
    float s = 0; int t = 0;
    for (int i = 0; i < 8; ++i) { s += b[i] + v[i]; t += c[i]; }
    out[blockIdx.x * 256 + threadIdx.x] = s + t;
}
template <int PAT> void run(const char *name, int nvalu, int wavesPerSimd, float *out, float *in)
{
    int iters = 200;
    int blocks = 256 * wavesPerSimd;    // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<PAT><<<blocks, 256>>>(out, in, 2);
    (void)hipEventRecord(a);
    k<PAT><<<blocks, 256>>>(out, in, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double instr_per_wave = (double)iters * REP * 8 * nvalu;
    double simd_cycles = ms * 1e-3 * 2.4e9;
    printf("%-34s waves/SIMD=%d  %.3f ms  %.2f cycles per VALU instr per SIMD (at 2.4 GHz)\n", name, wavesPerSimd, ms,
           simd_cycles / (instr_per_wave * wavesPerSimd));
}
int main()
{
    float *out, *in; (void)hipMalloc(&out, 256 * 2048 * 4); (void)hipMalloc(&in, 4096 * 4); (void)hipMemset(in, 0, 4096 * 4);
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("fma", 1, w, out, in);
        run<1>("cmp + 2 cndmask + add", 4, w, out, in);
        run<2>("cmp + cndmask + max + add", 4, w, out, in);
        run<3>("max + add", 2, w, out, in);
        run<4>("cmp + cndmask + add", 3, w, out, in);
    }
    return 0;
}
