// Helpers shared by the row-paired matrix-core conv kernels (conv_rp.hip: tiles; conv_stripe.hip: full-width stripes): operand types, the
// fp16 hi / lo split of an 8-channel pixel chunk, the LDS placement of the B fragments, power-of-two exponent helpers.
#pragma once
#include "common.hip.h"
#include <type_traits>
#include <utility>

typedef _Float16 rp_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 rp_f16x2 __attribute__((ext_vector_type(2)));
typedef float rp_f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int RP_MAXC = 64;          // input channels (after concat) per launch on this path

// 8 fp32 values -> 8 fp16 hi + 8 fp16 lo (lo = fp16(x - hi), both round-to-nearest: v_cvt_pk_f16_f32 on gfx950)
__device__ __forceinline__ void rp_split8(const float (&y)[8], uint4& hi, uint4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const rp_f32x2 v = {y[2 * i], y[2 * i + 1]};
        const rp_f16x2 h2 = __builtin_convertvector(v, rp_f16x2);
        h[i] = __builtin_bit_cast(unsigned, h2);
        l[i] = mi_split_lo2(h[i], y[2 * i], y[2 * i + 1]);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ uint4 rp_hi8(const float (&y)[8]) {
    unsigned h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const rp_f32x2 v = {y[2 * i], y[2 * i + 1]};
        h[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, rp_f16x2));
    }
    return make_uint4(h[0], h[1], h[2], h[3]);
}

template <class F, int... I>
__device__ __forceinline__ void rp_for_rounds(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

// B fragments lie in global memory as [block of 64 lanes][lane][hi | lo] (packing.pack_conv_weight_rp); read like that from LDS, a wave's 64 hi
// (or lo) chunks are 32 bytes apart: lanes i and i + 8 of every 16-lane pass hit the same banks (2-way conflict on every B read: the 14-23 % of
// LDS cycles the PMC passes showed as bank conflicts in every instantiation).  In LDS the two halves of a block are therefore kept as planes:
// chunk k = 128 blk + 2 lane + t  ->  128 blk + 64 t + lane.
__device__ __forceinline__ int rp_wl_index(int k) { return (k & ~127) | ((k & 1) << 6) | ((k & 127) >> 1); }

// exponent e with |m| in [2^(e-1), 2^e); 0 for zero / non-finite input (-> no scaling)
__device__ __forceinline__ int rp_exponent(float m) {
    const unsigned u = __float_as_uint(m) & 0x7fffffffu;
    const int be = (int)(u >> 23);
    if (be == 0 || be == 255) return 0;
    return be - 126;
}
__device__ __forceinline__ int rp_clamp_exp(int k) { return k < -60 ? -60 : (k > 60 ? 60 : k); }

}  // namespace
