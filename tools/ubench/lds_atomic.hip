// micro-benchmark: LDS atomic add throughput (float vs int, conflict-free vs random vs same address)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <typename T, int PATTERN>
__global__ __launch_bounds__(256) void k(T *out, int iters, const int *rnd)
{
    __shared__ T buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = 0;
    __syncthreads();
    int lane = threadIdx.x;
    int a = PATTERN == 0 ? lane : (PATTERN == 1 ? rnd[blockIdx.x * 256 + lane] & 8191 : (PATTERN == 2 ? 0 : (lane * 32) & 8191));
    for (int i = 0; i < iters; ++i) {
        atomicAdd(&buf[a], (T)1);
        a = (a + (PATTERN == 1 ? 977 : 256)) & 8191;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = buf[5];
}
template <typename T, int P> void run(const char *name, const int *rnd)
{
    T *out; hipMalloc(&out, 4096 * sizeof(T));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int iters = 2000, blocks = 1024;
    k<T, P><<<blocks, 256>>>(out, 10, rnd);
    hipEventRecord(a);
    k<T, P><<<blocks, 256>>>(out, iters, rnd);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double waveinstr = (double)blocks * 4 * iters;
    // 256 CUs; cycles per wave-instr per CU at 2.4 GHz
    printf("%-28s %.3f ms  -> %.1f cycles per wave-instr per CU\n", name, ms, ms * 1e-3 * 2.4e9 / (waveinstr / 256));
    hipFree(out);
}
int main()
{
    int *h = (int *)malloc(1024 * 256 * 4), *d;
    for (int i = 0; i < 1024 * 256; ++i) h[i] = rand();
    hipMalloc(&d, 1024 * 256 * 4); hipMemcpy(d, h, 1024 * 256 * 4, hipMemcpyHostToDevice);
    run<float, 0>("f32 conflict-free", d); run<float, 1>("f32 random", d); run<float, 2>("f32 same address", d); run<float, 3>("f32 same bank (32-way)", d);
    run<int, 0>("i32 conflict-free", d); run<int, 1>("i32 random", d); run<int, 2>("i32 same address", d);
    return 0;
}
