// Shared device helpers for the MinImagen gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "minimagen_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MI_MAX_CIN 256      // direct-conv family: input channels (after concat) per launch
#define MI_MAX_GROUPS 32

// SiLU as layers.py:128/144 computes it (x * sigmoid(x)); accurate exp, 1-ulp reciprocal.
__device__ __forceinline__ float mi_silu(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + expf(-v));
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

// host-side error plumbing (capi.hip)
void mi_set_error(const char* fmt, ...);
int mi_check_launch(const char* what);
