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
__global__ __launch_bounds__(1024) void victim(f32x2 c, const float* __restrict__ in, unsigned* bad, unsigned* bad_lanes, int iters, const float* pairs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float x0 = in[t & 4095], x1 = in[(t + 17) & 4095];
    f32x2 cv = c;                       // (forms 1, 2: the pair in VGPRs)
    asm volatile("" : "+v"(cv));
    float c0 = c[0], c1 = c[1];         // the reference multiplies SCALARS (the barriers keep the compiler from re-pairing them)
    asm volatile("" : "+s"(c0));
    asm volatile("" : "+s"(c1));
    unsigned wrong = 0;
    float one = 1.0f;                   // v_div_fixup_f32(x, 1, x) = x / 1 = x: the low half keeps its value, but is WRITTEN right before the packed read
    asm volatile("" : "+v"(one));
    for (int i = 0; i < iters; ++i) {
        const f32x2 x = {x0, x1};
        f32x2 d;
        if (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "s"(c), "v"(x));
        if (FORM == 1) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(cv), "v"(x));
        if (FORM == 2) { const f32x2 z = {0.f, 0.f}; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(cv), "v"(x), "v"(z)); }
        if (FORM == 3) d = c * x;
        // 4 / 5: the pair is LOADED into fixed SGPRs, used by the packed instruction, and the same SGPRs are overwritten right behind it --
        // by a scalar load of a zero pair (4) or by s_mov (5): a write-after-read on the SGPR operand of a multi-pass VALU instruction
        if (FORM == 4) asm volatile("s_load_dwordx2 s[20:21], %1, 0x0\n s_waitcnt lgkmcnt(0)\n v_pk_mul_f32 %0, s[20:21], %2\n s_load_dwordx2 s[20:21], %1, 0x8\n s_waitcnt lgkmcnt(0)"
                                    : "=&v"(d) : "s"(pairs), "v"(x) : "s20", "s21", "memory");
        if (FORM == 5) asm volatile("s_load_dwordx2 s[20:21], %1, 0x0\n s_waitcnt lgkmcnt(0)\n v_pk_mul_f32 %0, s[20:21], %2\n s_mov_b32 s20, 0\n s_mov_b32 s21, 0"
                                    : "=&v"(d) : "s"(pairs), "v"(x) : "s20", "s21", "memory");
        // 6: as 4 with two plain v_mul_f32 (single-register SGPR operands)
        if (FORM == 6) asm volatile("s_load_dwordx2 s[20:21], %1, 0x0\n s_waitcnt lgkmcnt(0)\n v_mul_f32 %0, s20, %2\n s_load_dwordx2 s[20:21], %1, 0x8\n s_waitcnt lgkmcnt(0)"
                                    : "=&v"(d[0]) : "s"(pairs), "v"(x0) : "s20", "s21", "memory");
        if (FORM == 6) d[1] = c1 * x1;
        // 7 .. 9 -- the pattern the bisection of the real kernel ends at (profiles/r04_pk_f32_hazard_bisect.txt): ONE half of the packed
        // instruction's 64-bit vector operand is written by the IMMEDIATELY preceding VALU instruction, the instruction works in place.
        //   7: v_div_fixup_f32 writes the low half (what hipcc emitted: x0 / s, then {c1, c2} * {x0, xt})   8: v_mul_f32 writes the low half
        //   9: as 7 with two independent instructions in between (the form that never failed in the kernel)
        if (FORM == 7) asm volatile("v_mov_b32 v101, %3\n s_nop 4\n v_div_fixup_f32 v100, %2, %4, %2\n v_pk_mul_f32 v[100:101], %1, v[100:101]\n s_nop 4\n v_mov_b32 %0, v100\n v_mov_b32 %5, v101"
                                    : "=&v"(d[0]), "+s"(c) , "+v"(x0), "+v"(x1), "+v"(one), "=&v"(d[1]) : : "v100", "v101");
        if (FORM == 8) asm volatile("v_mov_b32 v101, %3\n s_nop 4\n v_mul_f32 v100, %4, %2\n v_pk_mul_f32 v[100:101], %1, v[100:101]\n s_nop 4\n v_mov_b32 %0, v100\n v_mov_b32 %5, v101"
                                    : "=&v"(d[0]), "+s"(c) , "+v"(x0), "+v"(x1), "+v"(one), "=&v"(d[1]) : : "v100", "v101");
        if (FORM == 9) asm volatile("v_mov_b32 v101, %3\n s_nop 4\n v_div_fixup_f32 v100, %2, %4, %2\n v_mov_b32 v102, %2\n v_mov_b32 v103, %3\n v_pk_mul_f32 v[100:101], %1, v[100:101]\n s_nop 4\n v_mov_b32 %0, v100\n v_mov_b32 %5, v101"
                                    : "=&v"(d[0]), "+s"(c) , "+v"(x0), "+v"(x1), "+v"(one), "=&v"(d[1]) : : "v100", "v101", "v102", "v103");
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

// aggressor kinds: 0 matrix-core instructions only; 1 plain VALU with SGPR operands (v_fma_f32 v, s, v, v: the shape of the direct-conv /
// CrossEmbed inner loops: weights as wave-uniform scalars); 2 both, interleaved; 3 LDS traffic + VALU with SGPR operands
template <int KIND>
__global__ __launch_bounds__(256) void aggressor(float* out, int iters, const float* __restrict__ w) {
    __shared__ float lds[256 * 8];
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const _Float16 h = (_Float16)(threadIdx.x * 0.001f);
    const f16x8 u = {h, h, h, h, h, h, h, h};
    float v0 = threadIdx.x * 0.5f, v1 = 1.0f, v2 = 2.0f, v3 = 3.0f;
    for (int k = 0; k < 8; ++k) lds[threadIdx.x * 8 + k] = v0 + k;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0 || KIND == 2) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, u, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, u, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, u, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, u, a3, 0, 0, 0);
        }
        if (KIND >= 1) {
            const float s0 = w[(i * 4) & 255], s1 = w[(i * 4 + 1) & 255], s2 = w[(i * 4 + 2) & 255], s3 = w[(i * 4 + 3) & 255];    // uniform: scalar loads
            asm volatile("v_fma_f32 %0, %4, %0, %1\n v_fma_f32 %1, %5, %1, %2\n v_fma_f32 %2, %6, %2, %3\n v_fma_f32 %3, %7, %3, %0"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"(s0), "s"(s1), "s"(s2), "s"(s3));
        }
        if (KIND == 3) { v0 += lds[((threadIdx.x + i) & 255) * 8 + (i & 7)]; lds[threadIdx.x * 8 + (i & 7)] = v1; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3;
}

template <int FORM>
void run(const char* name, int launches, int with_aggressor, const float* in, float* sink, const float* pairs) {
    unsigned *bad, *lanes, h[2] = {0, 0};
    hipMalloc(&bad, 4); hipMalloc(&lanes, 4);
    hipMemset(bad, 0, 4); hipMemset(lanes, 0, 4);
    hipStream_t sv, sa;
    hipStreamCreate(&sv); hipStreamCreate(&sa);
    const f32x2 c = {1.2345678f, -0.87654321f};
    unsigned bad_launches = 0, prev = 0;
    for (int l = 0; l < launches; ++l) {
        if (with_aggressor == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(aggressor<0>), dim3(1024), dim3(256), 0, sa, sink, 3000, in);
        if (with_aggressor == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(aggressor<1>), dim3(1024), dim3(256), 0, sa, sink, 3000, in);
        if (with_aggressor == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(aggressor<2>), dim3(1024), dim3(256), 0, sa, sink, 3000, in);
        if (with_aggressor == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(aggressor<3>), dim3(1024), dim3(256), 0, sa, sink, 3000, in);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(victim<FORM>), dim3(32), dim3(1024), 0, sv, c, in, bad, lanes, 64, pairs);   // the shape of sampler_small_kernel
        if ((l & 63) == 63 || l + 1 == launches) {
            hipDeviceSynchronize();
            hipMemcpy(h, bad, 4, hipMemcpyDeviceToHost);
            if (h[0] != prev) { ++bad_launches; prev = h[0]; }
        }
    }
    hipMemcpy(h, bad, 4, hipMemcpyDeviceToHost);
    hipMemcpy(h + 1, lanes, 4, hipMemcpyDeviceToHost);
    printf("%-58s %s: %u wrong results in %d launches (%u lane-launches affected, >= %u check windows of 64 launches)\n", name,
           with_aggressor == 0 ? "alone on the GPU             " : with_aggressor == 1 ? "next to MFMA waves           " : with_aggressor == 2 ? "next to SGPR-operand VALU    "
           : with_aggressor == 3 ? "next to MFMA + SGPR-op. VALU " : "next to LDS + SGPR-op. VALU  ", h[0], launches, h[1], bad_launches);
    hipStreamDestroy(sv); hipStreamDestroy(sa);
    hipFree(bad); hipFree(lanes);
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 4000;
    float *in, *sink, hin[4096];
    for (int i = 0; i < 4096; ++i) hin[i] = 0.5f + 0.001f * i;
    hipMalloc(&in, sizeof(hin)); hipMalloc(&sink, 1024 * 256 * 4);
    hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
    float* pairs; const float hp[4] = {1.2345678f, -0.87654321f, 0.0f, 0.0f};         // the pair of run(), then a zero pair
    hipMalloc(&pairs, sizeof(hp));
    hipMemcpy(pairs, hp, sizeof(hp), hipMemcpyHostToDevice);
    for (int agg = 0; agg < 5; ++agg) {
        run<0>("v_pk_mul_f32 v, s[a:b], v   (SGPR pair operand, asm)", launches, agg, in, sink, pairs);
        run<1>("v_pk_mul_f32 v, v, v        (VGPR operands, asm)", launches, agg, in, sink, pairs);
        run<2>("v_pk_fma_f32 v, v, v, v     (VGPR operands, asm)", launches, agg, in, sink, pairs);
        run<3>("c * x on float2, uniform c  (compiler's choice)", launches, agg, in, sink, pairs);
        run<4>("s_load pair; v_pk_mul_f32 v, s, v; s_load SAME pair regs", launches, agg, in, sink, pairs);
        run<5>("s_load pair; v_pk_mul_f32 v, s, v; s_mov SAME pair regs", launches, agg, in, sink, pairs);
        run<6>("s_load pair; v_mul_f32 v, s, v;    s_load SAME pair regs", launches, agg, in, sink, pairs);
        run<7>("v_div_fixup_f32 v100 ; v_pk_mul_f32 v[100:101], s, v[100:101]", launches, agg, in, sink, pairs);
        run<8>("v_mul_f32 v100       ; v_pk_mul_f32 v[100:101], s, v[100:101]", launches, agg, in, sink, pairs);
        run<9>("v_div_fixup v100; 2 x v_mov; v_pk_mul_f32 v[100:101] ...    ", launches, agg, in, sink, pairs);
    }
    return 0;
}
