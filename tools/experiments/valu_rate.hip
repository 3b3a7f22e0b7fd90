// Issue rate of a few VALU instructions on gfx950 (GPU box): 8 waves per SIMD, 8 independent chains per wave, so that neither
// latency nor occupancy limits the rate.  Prints cycles per wave64 instruction per SIMD at the clock s_memtime implies.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/experiments/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096, CH = 8;
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned *out, unsigned seed) {
    unsigned a[CH], b = seed + threadIdx.x, c = seed * 3u + 1u;
    double d[CH];
    for (int i = 0; i < CH; ++i) { a[i] = b + i; d[i] = (double)(b + i); }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_alignbit_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_bfi_b32 %0, %2, %1, %0" : "+v"(a[i]) : "v"(b), "s"(c));
            if (OP == 3) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 4) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 5) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 6) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(d[(i + 1) % CH]) : "vcc");
            if (OP == 7) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) % CH]));
            if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
            if (OP == 9) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 10) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 11) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            if (OP == 12) asm volatile("v_sub_co_u32 %1, vcc, %2, %0\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a[i]), "=&v"(c) : "v"(b) : "vcc");   // 2 instructions
            if (OP == 13) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
            if (OP == 14) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 15) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) % CH]));
            if (OP == 16) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
            if (OP == 17) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 18) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 19) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 20) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 21) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 22) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");   // 2 instructions
        }
    }
    unsigned r = 0;
    for (int i = 0; i < CH; ++i) r ^= a[i] ^ (unsigned)d[i];
    if (r == 0x12345u) out[0] = r;
}
template <int OP> int run(const char *name, unsigned *out) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = 256 * 2 * 8;                                     // 2 workgroups of 1024 per CU resident: 8 waves per SIMD; 8 rounds
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, out, 7u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, out, 7u);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double winst = (double)blocks * 16 * ITER * CH;               // wave-instructions
    const double per_simd = winst / (256.0 * 4.0);
    printf("%-16s %8.3f ms  %.3e wave-instr/s  %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, ms,
           winst / (ms * 1e-3), ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    return 0;
}
int main() {
    unsigned *out;
    CHK(hipMalloc(&out, 64));
    run<0>("v_add_u32", out); run<3>("v_sub_u32", out); run<1>("v_alignbit_b32", out); run<2>("v_bfi_b32 (sgpr)", out);
    run<4>("v_and_or_b32", out); run<5>("v_lshl_or_b32", out); run<8>("v_cndmask_b32", out); run<9>("v_mad_u32_u24", out);
    run<10>("v_bcnt_u32_b32", out); run<11>("v_cmp_lt_u32", out); run<6>("v_cmp_lt_f64", out); run<7>("v_add_f64", out);
    run<13>("v_lshrrev_b32", out); run<14>("v_and_b32", out); run<19>("v_min_u32", out); run<18>("v_add3_u32", out); run<17>("v_mul_lo_u32", out);
    run<21>("v_add_f32", out); run<20>("v_fma_f32", out); run<15>("v_mul_f64", out); run<16>("v_cvt_u32_f64", out);
    printf("pairs (two instructions per step; the figures are per PAIR):\n");
    run<12>("sub_co + addc_co", out); run<22>("cmp + cndmask", out);
    return 0;
}
