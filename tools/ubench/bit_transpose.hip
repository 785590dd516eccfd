// Checks dg_bit_transpose64 (tools/ubench/bit_transpose.hpp) against a host transpose on random 64 x 64 bit matrices.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/bit_transpose.hip -o /tmp/bit_transpose && /tmp/bit_transpose
#include "bit_transpose.hpp"
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
__global__ void k(const uint64_t *in, uint64_t *out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t v = in[blockIdx.x * 64 + lane];
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    dg_bit_transpose64(lo, hi, lane);
    out[blockIdx.x * 64 + lane] = ((uint64_t)hi << 32) | lo;
}
int main()
{
    const int nb = 256;
    std::vector<uint64_t> h(nb * 64), g(nb * 64);
    srand(7);
    for (auto &x : h) { x = 0; for (int i = 0; i < 4; ++i) x = (x << 16) ^ (uint64_t)(rand() & 0xffff); if (rand() % 3 == 0) x &= (uint64_t)rand() * rand(); }
    for (int c = 0; c < 64; ++c) h[c] = 1ull << ((c * 7) % 64);          // a permutation matrix in block 0
    uint64_t *din, *dout;
    hipMalloc(&din, h.size() * 8); hipMalloc(&dout, h.size() * 8);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, din, dout);
    if (hipMemcpy(g.data(), dout, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 2; }
    int bad = 0;
    for (int b = 0; b < nb; ++b)
        for (int r = 0; r < 64; ++r) {
            uint64_t want = 0;
            for (int c = 0; c < 64; ++c) want |= ((h[b * 64 + c] >> r) & 1ull) << c;
            if (want != g[b * 64 + r]) { if (bad < 5) printf("block %d row %d: got %016llx want %016llx\n", b, r, (unsigned long long)g[b * 64 + r], (unsigned long long)want); ++bad; }
        }
    printf("bit_transpose: %d mismatching words of %d\n", bad, nb * 64);
    return bad != 0;
}
