// Micro-benchmark (dev tool): does VALU work issue in the shadow of an MFMA?  Each wave runs REP x [1 MFMA + N v_fma_f32].
// If the matrix pipe is separate, time stays flat in N up to ~7; if the instruction runs on the VALU lanes, it grows from N = 1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int N>
__global__ void k(float* out, int iters) {
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 1e-6f;
    f16x4 ha = {(_Float16)a, (_Float16)a, (_Float16)a, (_Float16)a}, hb = {(_Float16)b, (_Float16)b, (_Float16)b, (_Float16)b};
    f16x8 ha8 = {(_Float16)a, (_Float16)a, (_Float16)a, (_Float16)a, (_Float16)a, (_Float16)a, (_Float16)a, (_Float16)a}, hb8 = ha8;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4& acc = u == 0 ? acc0 : (u == 1 ? acc1 : (u == 2 ? acc2 : acc3));
            if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            if (KIND == 1) acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, hb, acc, 0, 0, 0);
            if (KIND == 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha8, hb8, acc, 0, 0, 0);
#pragma unroll
            for (int n = 0; n < N; ++n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[n & 7]) : "v"(b));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + s;
}
template <int KIND, int N> void run(const char* name) {
    float* out; hipMalloc(&out, 1 << 22);
    const int iters = 4000, blocks = 1024;          // 1 wave per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k<KIND, N>), dim3(blocks), dim3(64), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k<KIND, N>), dim3(blocks), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s + %2d v_fma per MFMA: %.3f ms  -> %.1f ns per [MFMA + N fma]\n", name, N, ms, ms * 1e6 / (iters * 4.0));
    hipFree(out);
}
#define SWEEP(K, NAME) run<K, 0>(NAME); run<K, 2>(NAME); run<K, 4>(NAME); run<K, 6>(NAME); run<K, 8>(NAME); run<K, 12>(NAME);
int main() {
    SWEEP(0, "mfma_f32_16x16x4_f32")
    SWEEP(1, "mfma_f32_16x16x16_f16")
    SWEEP(2, "mfma_f32_16x16x32_f16")
    return 0;
}
