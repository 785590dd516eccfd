// Store-bandwidth ceiling for the RoI forward's output pattern (never shipped).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/store_bw.hip -o tools/ubench/store_bw
// Patterns: 0 = plain linear fill (grid-stride, 16 B per lane); 1 = the forward kernel's: persistent single-wave
// workgroups, task = 12.25 KB chunk of `out` + the same chunk of `arg`, chunk index = task (block-strided);
// 2 = as 1 but a task writes one 24.5 KB chunk (out and arg interleaved per task: what a fused output would look like).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void st(float *p, v4f v)
{
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(p));
    else *reinterpret_cast<v4f *>(p) = v;
}
template <bool NT> __global__ void k_linear(float *a, size_t n4)
{
    const v4f v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) st<NT>(a + 4 * i, v);
}
// torch-like shape: no grid stride, block b owns a contiguous 16 KB tile (4 x 16 B per thread, 4 KB apart)
template <bool NT> __global__ void k_tiled(float *a, size_t n4)
{
    const v4f v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const size_t i = base + k * 256; if (i < n4) st<NT>(a + 4 * i, v); }
}
__global__ void k_two(float *a, float *b, size_t n4)
{
    const v4f v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { st<false>(a + 4 * i, v); st<false>(b + 4 * i, v); }
}
template <bool NT> __global__ void k_linear_read(const float *a, size_t n4, float *sink)
{
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const v4f v = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4f *>(a + 4 * i)) : *reinterpret_cast<const v4f *>(a + 4 * i);
        acc += v;
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) sink[0] = acc.x;
}
template <bool NT, int PAT> __global__ __launch_bounds__(64) void k_tasks(float *out, float *arg, long ntasks)
{
    const int lane = threadIdx.x;
    const v4f v = {1.f, 2.f, 3.f, (float)lane};
    for (long t = blockIdx.x; t < ntasks; t += gridDim.x) {
        float *o = (PAT == 1 ? out + t * 3136 : out + t * 6272) + lane * 4;
        float *a = (PAT == 1 ? arg + t * 3136 : out + t * 6272 + 3136) + lane * 4;
#pragma unroll
        for (int k = 0; k < 12; ++k) st<NT>(o + k * 256, v);
        if (lane < 16) st<NT>(o + 12 * 256, v);
#pragma unroll
        for (int k = 0; k < 12; ++k) st<NT>(a + k * 256, v);
        if (lane < 16) st<NT>(a + 12 * 256, v);
    }
}
// the uint16-argmax layout: 12.25 KB of values + 6.125 KB of indices per task
template <bool NT> __global__ __launch_bounds__(64) void k_tasks16(float *out, float *arg16, long ntasks)
{
    const int lane = threadIdx.x;
    const v4f v = {1.f, 2.f, 3.f, (float)lane};
    for (long t = blockIdx.x; t < ntasks; t += gridDim.x) {
        float *o = out + t * 3136 + lane * 4;
        float *a = arg16 + t * 1568 + lane * 4;
#pragma unroll
        for (int k = 0; k < 12; ++k) st<NT>(o + k * 256, v);
        if (lane < 16) st<NT>(o + 12 * 256, v);
#pragma unroll
        for (int k = 0; k < 6; ++k) st<NT>(a + k * 256, v);
        if (lane < 8) st<NT>(a + 6 * 256, v);
    }
}
int main(int argc, char **argv)
{
    const long ntasks = 512000L * 8;                 // config 3: 256 images x 2000 RoIs x 8 channel blocks
    const size_t half = (size_t)ntasks * 3136 * 4;   // 51.4 GB
    float *buf;
    if (hipMalloc(&buf, 2 * half) != hipSuccess) { printf("alloc failed\n"); return 1; }
    float *out = buf, *arg = buf + half / 4;
    double launch_bytes = 2.0 * half;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("%-52s %8.3f ms  %6.2f TB/s\n", name, best, launch_bytes / best / 1e9);
    };
    for (int blocks : {2048, 4096, 8192, 65536}) {
        char nm[96];
        snprintf(nm, sizeof nm, "linear nt, %d x 256 threads", blocks);
        timeit(nm, [&] { k_linear<true><<<blocks, 256>>>(buf, 2 * half / 16); });
        snprintf(nm, sizeof nm, "linear plain, %d x 256 threads", blocks);
        timeit(nm, [&] { k_linear<false><<<blocks, 256>>>(buf, 2 * half / 16); });
    }
    if (argc > 1 && argv[1][0] == 'a') {
        launch_bytes = 1.5 * half;
        for (int waves : {2048, 4096}) {
            char nm[96];
            snprintf(nm, sizeof nm, "uint16-argmax pattern nt, %d waves", waves);
            timeit(nm, [&] { k_tasks16<true><<<waves, 64>>>(out, arg, ntasks); });
            snprintf(nm, sizeof nm, "uint16-argmax pattern plain, %d waves", waves);
            timeit(nm, [&] { k_tasks16<false><<<waves, 64>>>(out, arg, ntasks); });
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'g') {
        // the forward's store pattern against the number of single-wave workgroups (tasks are block-strided)
        for (int waves : {2048, 8192, 16384, 65536, 262144, 524288, 1048576, 4096000}) {
            char nm[96];
            snprintf(nm, sizeof nm, "forward pattern nt, %d workgroups", waves);
            timeit(nm, [&] { k_tasks<true, 1><<<waves, 64>>>(out, arg, ntasks); });
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 't') {
        for (double gb : {4.0, 102.76, 4.0, 25.7, 102.76}) {
            size_t n4 = (size_t)(gb * 1e9 / 16);
            if (n4 > 2 * half / 16) n4 = 2 * half / 16;
            launch_bytes = n4 * 16.0;
            char nm[96];
            snprintf(nm, sizeof nm, "tiled plain fill of %.1f GB", gb);
            timeit(nm, [&] { k_tiled<false><<<(unsigned)((n4 + 1023) / 1024), 256>>>(buf, n4); });
            snprintf(nm, sizeof nm, "tiled nt fill of %.1f GB", gb);
            timeit(nm, [&] { k_tiled<true><<<(unsigned)((n4 + 1023) / 1024), 256>>>(buf, n4); });
            snprintf(nm, sizeof nm, "grid-stride plain fill of %.1f GB", gb);
            timeit(nm, [&] { k_linear<false><<<2048, 256>>>(buf, n4); });
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 's') {
        // fill rate against the size of the region written (a fresh sub-range of the same allocation each time)
        for (double gb : {1.0, 4.0, 12.85, 25.7, 51.4, 77.1, 102.8, 77.1, 51.4, 25.7, 12.85, 4.0, 1.0, 4.0, 102.8}) {
            size_t n4 = (size_t)(gb * 1e9 / 16);
            if (n4 > 2 * half / 16) n4 = 2 * half / 16;          // never past the allocation
            char nm[96];
            snprintf(nm, sizeof nm, "linear plain fill of %.1f GB", gb);
            launch_bytes = n4 * 16.0;
            timeit(nm, [&] { k_linear<false><<<2048, 256>>>(buf, n4); });
            snprintf(nm, sizeof nm, "two streams (x and x + 51.4 GB), %.1f GB total", gb);
            timeit(nm, [&] { k_two<<<2048, 256>>>(buf, buf + half / 4, n4 / 2); });
        }
        return 0;
    }
    if (argc > 1) {
        for (int blocks : {2048, 4096, 8192, 65536}) {
            char nm[96];
            snprintf(nm, sizeof nm, "linear read plain, %d x 256 threads", blocks);
            timeit(nm, [&] { k_linear_read<false><<<blocks, 256>>>(buf, 2 * half / 16, buf); });
            snprintf(nm, sizeof nm, "linear read nt, %d x 256 threads", blocks);
            timeit(nm, [&] { k_linear_read<true><<<blocks, 256>>>(buf, 2 * half / 16, buf); });
        }
        return 0;
    }
    for (int waves : {1024, 2048, 3072, 4096, 8192}) {
        char nm[96];
        snprintf(nm, sizeof nm, "forward pattern nt, %d waves", waves);
        timeit(nm, [&] { k_tasks<true, 1><<<waves, 64>>>(out, arg, ntasks); });
        snprintf(nm, sizeof nm, "forward pattern plain, %d waves", waves);
        timeit(nm, [&] { k_tasks<false, 1><<<waves, 64>>>(out, arg, ntasks); });
        snprintf(nm, sizeof nm, "fused 24.5 KB chunks nt, %d waves", waves);
        timeit(nm, [&] { k_tasks<true, 2><<<waves, 64>>>(out, arg, ntasks); });
    }
    return 0;
}
