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
// (A mask + v_cvt_pkrtz formulation was measured 16 % slower on MI355X than plain conversions; the v_fma_mix_f32 form below 4 % faster
//  on the attention kernel.  -DMI_SPLIT_PLAIN builds the plain form on the device too.)
typedef _Float16 mi_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 mi_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mi_split_f16(const float (&x)[4], mi_f16x4& hi, mi_f16x4& lo) {
#if !defined(MI_SPLIT_PLAIN) && !defined(HIPEMU)
    // x - float(hi) straight from the packed halves with v_fma_mix_f32 (fma(f16 -> f32, -1, x); exact, Sterbenz): no separate
    // v_cvt_f32_f16 per element.  The emulator build takes the plain form below, which computes the same bits.
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        const mi_f16x2 h2 = {(_Float16)x[e], (_Float16)x[e + 1]};
        const unsigned hb = __builtin_bit_cast(unsigned, h2);
        float l0, l1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hb), "v"(x[e]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hb), "v"(x[e + 1]));
        hi[e] = h2[0]; hi[e + 1] = h2[1];
        lo[e] = (_Float16)l0; lo[e + 1] = (_Float16)l1;
    }
#else
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 h = (_Float16)x[e];
        hi[e] = h;
        lo[e] = (_Float16)(x[e] - (float)h);
    }
#endif
}

// mi_act.bmod: row b of a tensor shared between the guidance halves lives at b % bmod.  The engine only ever shares between two
// halves (B2 = 2 bmod), so the (wave-uniform but ~40-instruction) integer division is kept off the common path.
__device__ __forceinline__ int mi_row_of(int b, int bmod) {
    if (bmod <= 0 || b < bmod) return b;
    return b < 2 * bmod ? b - bmod : b % bmod;
}

// GroupNorm statistics of a conv input (in0 ++ in1): the producers left per-tile partial (sum, sum of squares) per channel;
// the first `nlanes` work-items (a multiple of 64; TPC <= 64 lanes per channel, never straddling a wave) add them up in a fixed
// order in fp64 -> chS / chQ.  Deterministic, and the result does not depend on how the batch is sharded.  Called first thing in
// the kernel so that these short latency-bound loads are in flight before the bulk input loads.
__device__ __forceinline__ void mi_gn_channel_totals(const mi_act& in0, const mi_act& in1, int C0, int Cin, int b, int lane, int nlanes,
                                                     double* chS, double* chQ) {
    int TPC = 1;
    while (TPC < 64 && TPC * 2 * Cin <= nlanes) TPC *= 2;
    const int CPP = nlanes / TPC;                   // channels per pass
    for (int base = 0; base < Cin; base += CPP) {
        const int c = base + lane / TPC, sub = lane % TPC;
        double s = 0.0, q = 0.0;
        if (c < Cin) {
            const bool second = c >= C0;
            const mi_act& a = second ? in1 : in0;
            const int cc = second ? c - C0 : c;
            const int ba = mi_row_of(b, a.bmod);
            const float* st = a.stats + ((size_t)(ba * a.C + cc) * a.nt) * 2;
            int t = sub;
            for (; t + 3 * TPC < a.nt; t += 4 * TPC) {          // four independent loads in flight, added in tile order
                const float2 v0 = *reinterpret_cast<const float2*>(st + 2 * t);
                const float2 v1 = *reinterpret_cast<const float2*>(st + 2 * (t + TPC));
                const float2 v2 = *reinterpret_cast<const float2*>(st + 2 * (t + 2 * TPC));
                const float2 v3 = *reinterpret_cast<const float2*>(st + 2 * (t + 3 * TPC));
                s += (double)v0.x; q += (double)v0.y;
                s += (double)v1.x; q += (double)v1.y;
                s += (double)v2.x; q += (double)v2.y;
                s += (double)v3.x; q += (double)v3.y;
            }
            for (; t < a.nt; t += TPC) {
                const float2 v = *reinterpret_cast<const float2*>(st + 2 * t);
                s += (double)v.x;
                q += (double)v.y;
            }
            s *= (double)a.scale;
            q *= (double)a.scale * (double)a.scale;
        }
        for (int o = TPC >> 1; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
        if (c < Cin && sub == 0) { chS[c] = s; chQ[c] = q; }
    }
}

// mean and 1/std of one group from the channel totals: moments in fp64, the final rsqrt in fp32 like the reference
__device__ __forceinline__ void mi_gn_group_moments(const double* chS, const double* chQ, int c_lo, int c_hi, double count, float eps,
                                                    float& mean_out, float& rstd_out) {
    double s = 0.0, q = 0.0;
    for (int c = c_lo; c < c_hi; ++c) { s += chS[c]; q += chQ[c]; }
    const double inv_n = 1.0 / count;
    const double mean = s * inv_n;
    double var = q * inv_n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    mean_out = (float)mean;
    rstd_out = 1.0f / sqrtf((float)(var + (double)eps));
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
