// Stand-alone reproducer (dev tool) for the packed-fp32 hazard of profiles/r03_pk_f32_hazard.txt: does a packed fp32 VALU instruction return a
// wrong half when its wave shares a SIMD with matrix-core waves of ANOTHER kernel (another stream)?
//   victim    : every lane computes d = (c0 * x0, c1 * x1) with ONE packed instruction, ITER times, and compares each result with the two
//               scalar products.  Forms: 0 = v_pk_mul_f32 v, s[c0:c1], v   (SGPR pair operand -- what the SLP vectorizer generated)
//                                        1 = v_pk_mul_f32 v, v, v           (VGPR operands)
//                                        2 = v_pk_fma_f32 v, v, v, v        (VGPR operands -- what attention.hip's softmax relies on)
//                                        3 = compiler-generated packed multiply by a uniform pair (whatever hipcc selects)
//   aggressor : waves that issue v_mfma_f32_16x16x32_f16 back to back, on a second stream, on every CU.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o tools/ubench/pk_f32_hazard tools/ubench/pk_f32_hazard.hip (the flag keeps the
// REFERENCE products scalar)   Run: ./pk_f32_hazard [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int FORM>
__global__ __launch_bounds__(1024) void victim(f32x2 c, const float* __restrict__ in, unsigned* bad, unsigned* bad_lanes, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float x0 = in[t & 4095], x1 = in[(t + 17) & 4095];
    f32x2 cv = c;                       // (forms 1, 2: the pair in VGPRs)
    asm volatile("" : "+v"(cv));
    float c0 = c[0], c1 = c[1];         // the reference multiplies SCALARS (the barriers keep the compiler from re-pairing them)
    asm volatile("" : "+s"(c0));
    asm volatile("" : "+s"(c1));
    unsigned wrong = 0;
    for (int i = 0; i < iters; ++i) {
        const f32x2 x = {x0, x1};
        f32x2 d;
        if (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "s"(c), "v"(x));
        if (FORM == 1) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(cv), "v"(x));
        if (FORM == 2) { const f32x2 z = {0.f, 0.f}; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(cv), "v"(x), "v"(z)); }
        if (FORM == 3) d = c * x;
        float e0 = c0 * x0;
        asm volatile("" : "+v"(e0));
        float e1 = c1 * x1;
        asm volatile("" : "+v"(e1));
        wrong += (d[0] != e0 || d[1] != e1) ? 1u : 0u;
        x0 += 0.25f;
        asm volatile("" : "+v"(x0));
        x1 -= 0.125f;
    }
    if (wrong) { atomicAdd(bad, wrong); atomicAdd(bad_lanes, 1u); }
}

__global__ __launch_bounds__(256) void aggressor(float* out, int iters) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const _Float16 h = (_Float16)(threadIdx.x * 0.001f);
    const f16x8 u = {h, h, h, h, h, h, h, h};
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, u, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, u, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, u, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, u, a3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

template <int FORM>
void run(const char* name, int launches, bool with_aggressor, const float* in, float* sink) {
    unsigned *bad, *lanes, h[2] = {0, 0};
    hipMalloc(&bad, 4); hipMalloc(&lanes, 4);
    hipMemset(bad, 0, 4); hipMemset(lanes, 0, 4);
    hipStream_t sv, sa;
    hipStreamCreate(&sv); hipStreamCreate(&sa);
    const f32x2 c = {1.2345678f, -0.87654321f};
    unsigned bad_launches = 0, prev = 0;
    for (int l = 0; l < launches; ++l) {
        if (with_aggressor) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), 0, sa, sink, 3000);            // ~4 waves per SIMD of MFMA
        hipLaunchKernelGGL(HIP_KERNEL_NAME(victim<FORM>), dim3(32), dim3(1024), 0, sv, c, in, bad, lanes, 64);   // the shape of sampler_small_kernel
        if ((l & 63) == 63 || l + 1 == launches) {
            hipDeviceSynchronize();
            hipMemcpy(h, bad, 4, hipMemcpyDeviceToHost);
            if (h[0] != prev) { ++bad_launches; prev = h[0]; }
        }
    }
    hipMemcpy(h, bad, 4, hipMemcpyDeviceToHost);
    hipMemcpy(h + 1, lanes, 4, hipMemcpyDeviceToHost);
    printf("%-58s %s: %u wrong results in %d launches (%u lane-launches affected, >= %u check windows of 64 launches)\n", name,
           with_aggressor ? "next to MFMA waves of another stream" : "alone on the GPU                    ", h[0], launches, h[1], bad_launches);
    hipStreamDestroy(sv); hipStreamDestroy(sa);
    hipFree(bad); hipFree(lanes);
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 4000;
    float *in, *sink, hin[4096];
    for (int i = 0; i < 4096; ++i) hin[i] = 0.5f + 0.001f * i;
    hipMalloc(&in, sizeof(hin)); hipMalloc(&sink, 1024 * 256 * 4);
    hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
    for (int agg = 0; agg < 2; ++agg) {
        run<0>("v_pk_mul_f32 v, s[a:b], v   (SGPR pair operand, asm)", launches, agg, in, sink);
        run<1>("v_pk_mul_f32 v, v, v        (VGPR operands, asm)", launches, agg, in, sink);
        run<2>("v_pk_fma_f32 v, v, v, v     (VGPR operands, asm)", launches, agg, in, sink);
        run<3>("c * x on float2, uniform c  (compiler's choice)", launches, agg, in, sink);
    }
    return 0;
}
