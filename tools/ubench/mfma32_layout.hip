// Layout probe for v_mfma_f32_16x16x32_f16 on gfx950: checks D = A.B against the assumed operand layout
//   A: lane (i = l & 15, g = l >> 4) holds A[i][k = 8g + e], e = 0..7;  B: lane (j = l & 15, g) holds B[k = 8g + e][j];
//   D: lane holds D[4g + r][j = l & 15], r = 0..3.
// hipcc --offload-arch=gfx950 -O2 -o mfma32_layout mfma32_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const _Float16* A, const _Float16* B, float* D) {   // A [16][32], B [32][16], D [16][16]
    const int l = threadIdx.x, q = l & 15, g = l >> 4;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = A[q * 32 + 8 * g + e]; b[e] = B[(8 * g + e) * 16 + q]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + q] = c[r];
}
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// hazard probe: a K = 16 MFMA feeding a dependent K = 32 one on the same accumulator (and the reverse order)
template <int ORDER>
__global__ void mixed(const _Float16* A, const _Float16* B, const _Float16* A4, const _Float16* B4, float* D) {
    const int l = threadIdx.x, q = l & 15, g = l >> 4;
    f16x8 a, b; f16x4 a4, b4;
    for (int e = 0; e < 8; ++e) { a[e] = A[q * 32 + 8 * g + e]; b[e] = B[(8 * g + e) * 16 + q]; }
    for (int e = 0; e < 4; ++e) { a4[e] = A4[q * 16 + 4 * g + e]; b4[e] = B4[(4 * g + e) * 16 + q]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    if (ORDER == 0) { c = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    else { c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c, 0, 0, 0); }
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + q] = c[r];
}
template <int K32>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
    f16x8 a8, b8; f16x4 a4, b4;
    for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(threadIdx.x * 0.001f + e); b8[e] = (_Float16)(0.5f + e); }
    for (int e = 0; e < 4; ++e) { a4[e] = a8[e]; b4[e] = b8[e]; }
    f32x4 c[8];
    for (int t = 0; t < 8; ++t) c[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (K32) c[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[t], 0, 0, 0);
            else c[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[t], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) s += c[t][0] + c[t][1] + c[t][2] + c[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int K32>
static void time_rate(const char* name) {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 2048;       // 2048 blocks x 4 waves = 8 waves per SIMD over 1024 SIMDs
    hipLaunchKernelGGL(rate<K32>, dim3(blocks), dim3(256), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<K32>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 * iters * 8;              // MFMA instructions
    const double flop = n * 2.0 * 16 * 16 * (K32 ? 32 : 16);
    printf("%s: %.3f ms, %.2f ns per MFMA per SIMD, %.0f TFLOP/s\n", name, ms, ms * 1e6 / (n / 1024.0), flop / (ms * 1e-3) / 1e12);
}
// dependent chains (one accumulator) with NV independent VALU FMAs between consecutive MFMAs: what a softmax-then-PV loop looks like
template <int K32, int NV>
__global__ __launch_bounds__(256) void chain(float* out, int iters) {
    f16x8 a8, b8; f16x4 a4, b4;
    for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(threadIdx.x * 0.001f + e); b8[e] = (_Float16)(0.5f + e); }
    for (int e = 0; e < 4; ++e) { a4[e] = a8[e]; b4[e] = b8[e]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 0.01f + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (K32) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c, 0, 0, 0);
            else c = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
        }
    }
    float s = c[0] + c[1] + c[2] + c[3];
    for (int k = 0; k < 8; ++k) s += v[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// same, with NV v_exp_f32 (transcendental unit) between consecutive MFMAs
template <int K32, int NV>
__global__ __launch_bounds__(256) void chain_exp(float* out, int iters) {
    f16x8 a8, b8; f16x4 a4, b4;
    for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(threadIdx.x * 0.001f + e); b8[e] = (_Float16)(0.5f + e); }
    for (int e = 0; e < 4; ++e) { a4[e] = a8[e]; b4[e] = b8[e]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = -(threadIdx.x * 0.01f + k);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (K32) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c, 0, 0, 0);
            else c = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k & 7] = __builtin_amdgcn_exp2f(v[k & 7]);
        }
    }
    float s = c[0] + c[1] + c[2] + c[3];
    for (int k = 0; k < 8; ++k) s += v[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int K32, int NV>
static void time_chain_exp(const char* name, int wps) {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * wps;
    hipLaunchKernelGGL((chain_exp<K32, NV>), dim3(blocks), dim3(256), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain_exp<K32, NV>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s, %d v_exp_f32 between, %d waves/SIMD: %.1f ns per group per wave, %.1f ns per SIMD slot\n", name, NV, wps, ms * 1e6 / (iters * 8.0), ms * 1e6 / (iters * 8.0 * wps));
}
template <int K32, int NV>
static void time_chain(const char* name, int wps) {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * wps;                   // wps waves per SIMD (256 CUs x 4 SIMDs, 4 waves per block)
    hipLaunchKernelGGL((chain<K32, NV>), dim3(blocks), dim3(256), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain<K32, NV>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s, %d VALU between, %d waves/SIMD: %.1f ns per (MFMA + VALU group) per wave, %.1f ns per SIMD slot\n", name, NV, wps, ms * 1e6 / (iters * 8.0), ms * 1e6 / (iters * 8.0 * wps));
}
int main() {
    for (int wps = 1; wps <= 3; wps += 2) {
        time_chain_exp<0, 0>("dependent 16x16x16", wps); time_chain_exp<0, 1>("dependent 16x16x16", wps);
        time_chain_exp<0, 2>("dependent 16x16x16", wps); time_chain_exp<0, 4>("dependent 16x16x16", wps);
        time_chain<0, 0>("dependent 16x16x16", wps); time_chain<1, 0>("dependent 16x16x32", wps);
        time_chain<0, 4>("dependent 16x16x16", wps); time_chain<1, 4>("dependent 16x16x32", wps);
        time_chain<0, 8>("dependent 16x16x16", wps); time_chain<1, 8>("dependent 16x16x32", wps);
        time_chain<0, 16>("dependent 16x16x16", wps); time_chain<1, 16>("dependent 16x16x32", wps);
    }
    time_rate<0>("v_mfma_f32_16x16x16_f16");
    time_rate<1>("v_mfma_f32_16x16x32_f16");
    _Float16 hA[16 * 32], hB[32 * 16];
    float ref[256] = {0}, hD[256];
    for (int i = 0; i < 512; ++i) { hA[i] = (_Float16)((float)((i * 37) % 23 - 11) * 0.125f); hB[i] = (_Float16)((float)((i * 53) % 19 - 9) * 0.25f); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) ref[i * 16 + j] += (float)hA[i * 32 + k] * (float)hB[k * 16 + j];
    _Float16 *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    float err = 0.f;
    for (int i = 0; i < 256; ++i) err = fmaxf(err, fabsf(hD[i] - ref[i]));
    printf("v_mfma_f32_16x16x32_f16 assumed layout: max |err| = %g (%s)\n", err, err < 1e-3f ? "layout confirmed" : "LAYOUT MISMATCH");
    {   // K = 16 then dependent K = 32 (and the reverse) on one accumulator
        _Float16 hA4[16 * 16], hB4[16 * 16];
        for (int i = 0; i < 256; ++i) { hA4[i] = (_Float16)((float)((i * 29) % 17 - 8) * 0.25f); hB4[i] = (_Float16)((float)((i * 31) % 13 - 6) * 0.5f); }
        float ref2[256];
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = ref[i * 16 + j]; for (int k = 0; k < 16; ++k) s += (float)hA4[i * 16 + k] * (float)hB4[k * 16 + j]; ref2[i * 16 + j] = s; }
        _Float16 *dA4, *dB4; hipMalloc(&dA4, sizeof(hA4)); hipMalloc(&dB4, sizeof(hB4));
        hipMemcpy(dA4, hA4, sizeof(hA4), hipMemcpyHostToDevice); hipMemcpy(dB4, hB4, sizeof(hB4), hipMemcpyHostToDevice);
        for (int order = 0; order < 2; ++order) {
            if (order == 0) hipLaunchKernelGGL(mixed<0>, dim3(1), dim3(64), 0, 0, dA, dB, dA4, dB4, dD);
            else hipLaunchKernelGGL(mixed<1>, dim3(1), dim3(64), 0, 0, dA, dB, dA4, dB4, dD);
            hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
            float e2 = 0.f;
            for (int i = 0; i < 256; ++i) e2 = fmaxf(e2, fabsf(hD[i] - ref2[i]));
            printf("%s on one accumulator: max |err| = %g\n", order == 0 ? "K=16 then dependent K=32" : "K=32 then dependent K=16", e2);
        }
    }
    return 0;
}
