// Shared device helpers for the MinImagen gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "minimagen_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MI_MAX_CIN 256      // direct-conv family: input channels (after concat) per launch
#define MI_MAX_GROUPS 32

// SiLU as layers.py:128/144 computes it (x * sigmoid(x)).  MI_SILU_EXP2=1: sigmoid through the hardware exp2
// (v_exp_f32, ~1 ulp on the exponential, argument scaled by -log2(e)); 0: libm expf.  1-ulp reciprocal either way.
#ifndef MI_SILU_EXP2
#define MI_SILU_EXP2 1
#endif
__device__ __forceinline__ float mi_silu(float v) {
    if (MI_SILU_EXP2) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
    return v * __builtin_amdgcn_rcpf(1.0f + expf(-v));
}

// fp16 hi/lo split of four fp32 values for the 3-term MFMA products: hi = fp16(x), lo = fp16(x - hi) (22 mantissa bits kept).
// (A mask + v_cvt_pkrtz formulation was measured 16 % slower on MI355X than these plain conversions.)
typedef _Float16 mi_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mi_split_f16(const float (&x)[4], mi_f16x4& hi, mi_f16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 h = (_Float16)x[e];
        hi[e] = h;
        lo[e] = (_Float16)(x[e] - (float)h);
    }
}

__device__ __forceinline__ float mi_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float mi_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

int mi_conv_mfma_launch(const mi_conv_params& p, hipStream_t st);   // conv_mfma.hip

// host-side error plumbing (capi.hip)
void mi_set_error(const char* fmt, ...);
int mi_check_launch(const char* what);
