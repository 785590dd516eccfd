// micro-benchmark: round trip of a flag between two workgroups, on the same XCD and on different ones, for the cache-scope
// bits of the load that polls (the store that publishes is `sc0 sc1`, system scope write-through, or `sc1`).  What it is for:
// the NMS fixed point's workgroups exchange 16 KB of state per pass through agent-scope loads (a trip to the memory side,
// about 2.4 us); if a poll with a narrower scope sees another CU's store through the XCD's own L2 when both run on one XCD,
// 64 workgroups placed on one XCD could exchange in a fraction of that.
// build: hipcc --offload-arch=gfx950 -O2 -o pingpong pingpong.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int LD> __device__ __forceinline__ unsigned ld(const unsigned *p)
{
    unsigned v;
    if (LD == 0) asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == 1) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == 2) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == 4) asm volatile("global_load_dword %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == 5) asm volatile("buffer_inv sc0\n global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == 6) asm volatile("buffer_inv sc1\n global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int ST> __device__ __forceinline__ void st(unsigned *p, unsigned v)
{
    if (ST == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : : "v"(p), "v"(v) : "memory");
    if (ST == 1) asm volatile("global_store_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : : "v"(p), "v"(v) : "memory");
    if (ST == 2) asm volatile("global_store_dword %0, %1, off\n s_waitcnt vmcnt(0)" : : "v"(p), "v"(v) : "memory");
}

// workgroups a and b play; the others leave at once.  flags: [0] = ping, [32] = pong (different cache lines); out: xcc ids,
// rounds completed, clocks
template <int LD, int ST>
__global__ void k(unsigned *flags, int a, int b, int rounds, unsigned limit, unsigned long long *out)
{
    const int g = blockIdx.x;
    if (threadIdx.x != 0 || (g != a && g != b)) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const bool first = g == a;
    out[first ? 0 : 1] = xcc & 0xf;
    unsigned *mine = flags + (first ? 0 : 32), *theirs = flags + (first ? 32 : 0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    int done = 0;
    for (int i = 1; i <= rounds; ++i) {
        if (first) st<ST>(mine, (unsigned)i);
        unsigned spins = 0;
        while (ld<LD>(theirs) != (unsigned)i) { if (++spins > limit) { out[first ? 2 : 3] = done; out[first ? 4 : 5] = __builtin_readcyclecounter() - t0; return; } }
        if (!first) st<ST>(mine, (unsigned)i);
        done = i;
    }
    out[first ? 2 : 3] = done;
    out[first ? 4 : 5] = __builtin_readcyclecounter() - t0;
}

template <int LD, int ST> void run(const char *name, int a, int b)
{
    unsigned *flags; unsigned long long *out, h[6];
    (void)hipMalloc(&flags, 256 * 4); (void)hipMalloc(&out, 6 * 8);
    (void)hipMemset(flags, 0, 256 * 4); (void)hipMemset(out, 0, 6 * 8);
    const int rounds = 200;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<LD, ST><<<16, 64>>>(flags, a, b, rounds, 20000u, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-34s wg %2d (xcc %llu) <-> wg %2d (xcc %llu): %3llu / %d rounds, %.2f us per round trip (kernel %.3f ms)\n", name, a, h[0], b, h[1],
           h[2], rounds, h[2] ? ms * 1e3 / (double)h[2] : 0.0, ms);
    (void)hipFree(flags); (void)hipFree(out);
}

int main()
{
    for (int pair = 0; pair < 2; ++pair) {
        const int a = 0, b = pair == 0 ? 8 : 1;
        run<0, 0>("load plain, store sc0 sc1", a, b);
        run<1, 0>("load sc0, store sc0 sc1", a, b);
        run<2, 0>("load sc1, store sc0 sc1", a, b);
        run<3, 0>("load sc0 sc1, store sc0 sc1", a, b);
        run<4, 0>("load nt, store sc0 sc1", a, b);
        run<5, 0>("buffer_inv sc0 + load, store sc0 sc1", a, b);
        run<6, 0>("buffer_inv sc1 + load, store sc0 sc1", a, b);
        run<1, 1>("load sc0, store sc1", a, b);
        run<2, 1>("load sc1, store sc1", a, b);
        run<1, 2>("load sc0, store plain", a, b);
        run<2, 2>("load sc1, store plain", a, b);
        run<5, 2>("buffer_inv sc0 + load, store plain", a, b);
    }
    return 0;
}
