// Micro-benchmark (dev tool): issue rate of the VALU forms the conv kernels can use, per SIMD, in shader cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters, float s0) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    float b = 1.0001f, c = 0.5f;
    f2 pb = {b, b}, pc = {c, c};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (MODE == 1) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));) }
        if (MODE == 2) { REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s0), "v"(c));) }
        if (MODE == 3) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (MODE == 4) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %2, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %4, %4, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %6, %6, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel_hi:[1,0,1]" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));) }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
}
template <int MODE> void run(const char* name, int waves_per_simd, int flop_per_lane_instr) {
    float* out; long long* cyc; hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    const int iters = 2000, blocks = 256 * 4;      // 4 blocks/CU of (64*waves_per_simd) threads -> waves_per_simd waves on each SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * waves_per_simd), 0, 0, out, cyc, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * waves_per_simd), 0, 0, out, cyc, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double instr_per_wave = (double)iters * 64;
    double tflops = (double)blocks * 64 * waves_per_simd * instr_per_wave * flop_per_lane_instr / (ms * 1e-3) / 1e12;
    printf("%-28s waves/SIMD=%d  counter ticks per instr per wave = %.2f  (%.3f ms, %.1f TFLOP/s chip)\n", name, waves_per_simd, (double)c / instr_per_wave, ms, tflops);
}
int main() {
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("v_fma_f32", 1, 2); run<1>("v_pk_fma_f32", 1, 4); run<4>("v_pk_fma_f32 op_sel bcast", 1, 4); run<2>("v_fmac_f32 sgpr", 1, 2); run<3>("v_exp_f32", 1, 1); }
        if (w == 2) { run<0>("v_fma_f32", 2, 2); run<1>("v_pk_fma_f32", 2, 4); run<2>("v_fmac_f32 sgpr", 2, 2); run<3>("v_exp_f32", 2, 1); }
        if (w == 4) { run<0>("v_fma_f32", 4, 2); run<1>("v_pk_fma_f32", 4, 4); }
    }
    return 0;
}
