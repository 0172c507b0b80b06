// Micro-benchmark (dev tool): issue cost, in shader cycles per wave64 instruction and SIMD, of the VALU forms of the attention softmax
// (exp2, fp16 split, packed fp32, max3) with 4 waves per SIMD -- what bounds cross_attn_f16x3_kernel once the matrix pipe is not.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define OP8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + .1f, a2 = a0 + .2f, a3 = a0 + .3f, a4 = a0 + .4f, a5 = a0 + .5f, a6 = a0 + .6f, a7 = a0 + .7f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    unsigned h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, h6 = 0, h7 = 0;
    float b = 1.0001f, c = 0.5f;
    f2 pb = {b, b};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (MODE == 1) { REP8(asm volatile("v_cvt_pk_f16_f32 %0, %8, %9\n v_cvt_pk_f16_f32 %1, %9, %10\n v_cvt_pk_f16_f32 %2, %10, %11\n v_cvt_pk_f16_f32 %3, %11, %12\n v_cvt_pk_f16_f32 %4, %12, %13\n v_cvt_pk_f16_f32 %5, %13, %14\n v_cvt_pk_f16_f32 %6, %14, %15\n v_cvt_pk_f16_f32 %7, %15, %8" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));) }
        if (MODE == 2) { REP8(asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %8, %9, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %8, %9, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %8, %9, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel_hi:[1,0,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h0), "v"(b));) }
        if (MODE == 3) { REP8(asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (MODE == 4) { REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));) }
        if (MODE == 5) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (MODE == 6) { REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (MODE == 7) { REP8(asm volatile("v_fma_mixlo_f16 %0, %8, %9, %10 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %1, %8, %9, %11 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %2, %8, %9, %12 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %3, %8, %9, %13 op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 %0, %8, %9, %14 op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 %1, %8, %9, %15 op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 %2, %8, %9, %16 op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 %3, %8, %9, %17 op_sel_hi:[1,0,0]" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(h4), "v"(b), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));) }
        if (MODE == 8) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));) }
        if (MODE == 9) { REP8(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7));) }
        if (MODE == 10) { REP8(asm volatile("v_pk_mul_f16 %0, %0, %8\n v_pk_mul_f16 %1, %1, %8\n v_pk_mul_f16 %2, %2, %8\n v_pk_mul_f16 %3, %3, %8\n v_pk_mul_f16 %4, %4, %8\n v_pk_mul_f16 %5, %5, %8\n v_pk_mul_f16 %6, %6, %8\n v_pk_mul_f16 %7, %7, %8" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(h0));) }
        if (MODE == 11) { REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(h1));) }
        if (MODE == 12) { REP8(asm volatile("v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %9\n v_cvt_f32_f16 %2, %10\n v_cvt_f32_f16 %3, %11\n v_cvt_f32_f16 %4, %12\n v_cvt_f32_f16 %5, %13\n v_cvt_f32_f16 %6, %14\n v_cvt_f32_f16 %7, %15" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(h0), "v"(h1), "v"(h2), "v"(h3), "v"(h4), "v"(h5), "v"(h6), "v"(h7));) }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1] + (float)(h0 ^ h1 ^ h2 ^ h3 ^ h4 ^ h5 ^ h6 ^ h7);
}
template <int MODE> void run(const char* name, int waves_per_simd) {
    float* out; long long* cyc; hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    const int iters = 2000, blocks = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * waves_per_simd), 0, 0, out, cyc, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * waves_per_simd), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double instr_per_wave = (double)iters * 64;
    // wave-instructions per second and SIMD -> ns per instruction per SIMD (x clock = cycles)
    const double ns = ms * 1e6 / (instr_per_wave * waves_per_simd);
    printf("%-24s waves/SIMD=%d  ticks/instr/wave %.2f   %.3f ms   %.2f ns per wave-instruction and SIMD (= %.2f cycles at 2.4 GHz)\n", name, waves_per_simd, (double)c / instr_per_wave, ms, ns, ns * 2.4);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 4}) {
        run<0>("v_fma_f32", w); run<6>("v_add_f32", w); run<11>("v_and_b32", w); run<3>("v_max3_f32", w); run<1>("v_cvt_pk_f16_f32", w); run<12>("v_cvt_f32_f16", w);
        run<2>("v_fma_mix_f32", w); run<7>("v_fma_mixlo/hi_f16", w); run<4>("v_pk_add_f32", w); run<8>("v_pk_fma_f32", w); run<10>("v_pk_mul_f16", w);
        run<5>("v_exp_f32", w); run<9>("v_exp_f16", w);
    }
    return 0;
}
