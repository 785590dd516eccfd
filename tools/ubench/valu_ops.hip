// micro-benchmark: issue cost of single VALU instructions on gfx950 (inline asm, 8 independent chains), for the
// argmax scan of the wide-crop RoI forward: which of compare / select / exec-masked move / bit ops are full rate.
// build: hipcc --offload-arch=gfx950 -O2 -o valu_ops valu_ops.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 32
#define NCH 8
template <int PAT>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters)
{
    float b[NCH], v[NCH];
    int c[NCH], r[NCH];
    double p[NCH], q[NCH];
    for (int i = 0; i < NCH; ++i) {
        b[i] = in[threadIdx.x + i]; v[i] = in[threadIdx.x + 8 + i]; c[i] = i; r[i] = threadIdx.x + i;
        p[i] = in[threadIdx.x + 16 + i]; q[i] = in[threadIdx.x + 24 + i];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rr = 0; rr < REP; ++rr) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                if (PAT == 0) asm volatile("v_max_f32 %0, %0, %1" : "+v"(b[i]) : "v"(v[i]));
                if (PAT == 1) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(v[i]), "v"(b[i]) : "vcc");
                if (PAT == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c[i]) : "v"(r[i]) : "vcc");
                if (PAT == 3) asm volatile("v_cmpx_gt_f32 vcc, %0, %1\n s_mov_b64 exec, -1" : : "v"(v[i]), "v"(b[i]) : "vcc");
                if (PAT == 4) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(c[i]) : "v"(r[i]), "v"(v[i]));
                if (PAT == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(v[i]), "v"(v[(i + 1) % NCH]));
                if (PAT == 6) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "+v"(p[i]) : "v"(q[i]), "v"(q[(i + 1) % NCH]));
                if (PAT == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(c[i]) : "v"(r[i]));
                if (PAT == 8)
                    asm volatile("v_cmp_gt_f32 vcc, %2, %0\n v_cndmask_b32 %1, %1, %3, vcc\n v_max_f32 %0, %0, %2"
                                 : "+v"(b[i]), "+v"(c[i]) : "v"(v[i]), "v"(r[i]) : "vcc");
                if (PAT == 9)
                    asm volatile("v_cmpx_gt_f32 vcc, %3, %4\n v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]\n s_mov_b64 exec, -1"
                                 : "+v"(p[i]) : "v"(q[i]), "v"(q[(i + 1) % NCH]), "v"(v[i]), "v"(b[i]) : "vcc");
                if (PAT == 10)
                    asm volatile("v_cmpx_gt_f32 vcc, %2, %0\n v_mov_b32 %1, %3\n v_mov_b32 %0, %2\n s_mov_b64 exec, -1"
                                 : "+v"(b[i]), "+v"(c[i]) : "v"(v[i]), "v"(r[i]) : "vcc");
                if (PAT == 11) {
                    float nb; int d;
                    asm volatile("v_max_f32 %2, %0, %4\n v_sub_f32 %3, %0, %2\n v_ashrrev_i32 %3, 31, %3\n v_bfi_b32 %1, %3, %5, %1\n v_mov_b32 %0, %2"
                                 : "+v"(b[i]), "+v"(c[i]), "=&v"(nb), "=&v"(d) : "v"(v[i]), "v"(r[i]));
                }
                if (PAT == 12) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1" : : "v"(v[i]), "v"(b[i]) : "s20", "s21");
                if (PAT == 13) asm volatile("v_maximum3_f32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(v[i]), "v"(v[(i + 1) % NCH]));
                if (PAT == 14) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(b[i]) : "v"(v[i]));
                if (PAT == 15) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(c[i]));
                if (PAT == 16) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(c[i]) : "v"(r[i]) : "s20", "s21");
                if (PAT == 17) asm volatile("v_max_u32 %0, %0, %1" : "+v"(c[i]) : "v"(r[i]));
                if (PAT == 18) asm volatile("v_and_b32 %0, %0, %1" : "+v"(c[i]) : "v"(r[i]));
                if (PAT == 19)   // max on the value; code by exec-masked move
                    asm volatile("v_cmpx_gt_f32 vcc, %2, %0\n v_mov_b32 %1, %3\n s_mov_b64 exec, -1\n v_max_f32 %0, %0, %2"
                                 : "+v"(b[i]), "+v"(c[i]) : "v"(v[i]), "v"(r[i]) : "vcc");
                if (PAT == 20) asm volatile("v_cmp_gt_f32 vcc, %1, %0\n v_max_f32 %0, %0, %1" : "+v"(b[i]) : "v"(v[i]) : "vcc");
                if (PAT == 21) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
                if (PAT == 22) asm volatile("v_max_i32 %0, %0, %1" : "+v"(c[i]) : "v"(r[i]));
                if (PAT == 23) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(v[i]), "v"(v[(i + 1) % NCH]));
                if (PAT == 24) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(c[i]) : "v"(r[i]));
                if (PAT == 25) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(c[i]), "v"(r[i]) : "vcc");
                if (PAT == 26) asm volatile("v_min_f64 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
                if (PAT == 27) asm volatile("v_max_f64 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
                if (PAT == 28) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(p[i]), "v"(q[i]) : "vcc");
                if (PAT == 29) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(p[i]) : "v"(q[i]));
                if (PAT == 30) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c[i]) : "v"(r[i]));
                if (PAT == 31) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(c[i]) : "v"(r[i]));
                if (PAT == 32) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(c[i]));
                if (PAT == 33) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(b[i]) : "v"(v[i]));
                if (PAT == 34) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(b[i]) : "v"(v[i]));
                if (PAT == 35) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(c[i]) : "v"(r[i]));
                if (PAT == 36) asm volatile("v_min_f32 %0, %0, %1" : "+v"(b[i]) : "v"(v[i]));
                if (PAT == 37) asm volatile("v_rcp_f32 %0, %0" : "+v"(b[i]));
                if (PAT == 38) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(p[i]) : "v"(b[i]));
                if (PAT == 39) asm volatile("v_or_b32 %0, %0, %1" : "+v"(c[i]) : "v"(r[i]));
            }
        }
    }
    float s = 0; int t = 0;
    for (int i = 0; i < NCH; ++i) { s += b[i] + v[i] + (float)p[i]; t += c[i]; }
    out[blockIdx.x * 256 + threadIdx.x] = s + t;
}
template <int PAT> void run(const char *name, int nvalu, int wavesPerSimd, float *out, float *in)
{
    int iters = 400;
    int blocks = 256 * wavesPerSimd;    // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<PAT><<<blocks, 256>>>(out, in, 2);
    (void)hipEventRecord(a);
    k<PAT><<<blocks, 256>>>(out, in, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double groups_per_wave = (double)iters * REP * NCH;
    double simd_cycles = ms * 1e-3 * 2.4e9;
    printf("%-44s waves/SIMD=%d  %.3f ms  %.2f cycles per group per SIMD (at 2.4 GHz), %d VALU in group\n", name, wavesPerSimd, ms,
           simd_cycles / (groups_per_wave * wavesPerSimd), nvalu);
}
int main()
{
    float *out, *in; (void)hipMalloc(&out, 256 * 2048 * 4); (void)hipMalloc(&in, 4096 * 4); (void)hipMemset(in, 0, 4096 * 4);
    for (int w = 2; w <= 4; w *= 2) {
        run<0>("v_max_f32", 1, w, out, in);
        run<1>("v_cmp_gt_f32 vcc", 1, w, out, in);
        run<12>("v_cmp_gt_f32_e64 sgpr", 1, w, out, in);
        run<25>("v_cmp_eq_u32 vcc", 1, w, out, in);
        run<2>("v_cndmask_b32 vcc", 1, w, out, in);
        run<16>("v_cndmask_b32_e64 sgpr", 1, w, out, in);
        run<3>("v_cmpx_gt_f32 + s_mov exec", 1, w, out, in);
        run<4>("v_bfi_b32", 1, w, out, in);
        run<5>("v_max3_f32", 1, w, out, in);
        run<13>("v_maximum3_f32", 1, w, out, in);
        run<23>("v_med3_f32", 1, w, out, in);
        run<6>("v_pk_mov_b32", 1, w, out, in);
        run<21>("v_pk_add_f32", 1, w, out, in);
        run<7>("v_mov_b32", 1, w, out, in);
        run<14>("v_sub_f32", 1, w, out, in);
        run<15>("v_ashrrev_i32", 1, w, out, in);
        run<17>("v_max_u32", 1, w, out, in);
        run<22>("v_max_i32", 1, w, out, in);
        run<18>("v_and_b32", 1, w, out, in);
        run<24>("v_lshl_or_b32", 1, w, out, in);
        run<36>("v_min_f32", 1, w, out, in);
        run<33>("v_mul_f32", 1, w, out, in);
        run<34>("v_fma_f32", 1, w, out, in);
        run<37>("v_rcp_f32", 1, w, out, in);
        run<30>("v_add_u32", 1, w, out, in);
        run<31>("v_xor_b32", 1, w, out, in);
        run<39>("v_or_b32", 1, w, out, in);
        run<32>("v_lshlrev_b32", 1, w, out, in);
        run<35>("v_mad_u32_u24", 1, w, out, in);
        run<26>("v_min_f64", 1, w, out, in);
        run<27>("v_max_f64", 1, w, out, in);
        run<29>("v_fma_f64", 1, w, out, in);
        run<38>("v_cvt_f64_f32", 1, w, out, in);
        run<28>("v_cmp_lt_u64 vcc", 1, w, out, in);
        run<8>("cmp + cndmask + max (current scan)", 3, w, out, in);
        run<20>("cmp + max", 2, w, out, in);
        run<9>("cmpx + pk_mov + s_mov exec", 2, w, out, in);
        run<10>("cmpx + mov + mov + s_mov exec", 3, w, out, in);
        run<19>("cmpx + mov + s_mov exec + max", 3, w, out, in);
        run<11>("max + sub + ashr + bfi + mov", 5, w, out, in);
    }
    return 0;
}
