// Shared device helpers for the MinImagen gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include "minimagen_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Pointers that reach a kernel inside a by-value parameter struct and then go through a select (in0 / in1, res0 / res1) lose their
// address space: the compiler emits FLAT loads, which count on vmcnt AND lgkmcnt -- every LDS wait then also waits for the global
// loads in flight and software prefetch is dead.  mi_global() pins the global address space (identity on the emulator).
#if defined(HIPEMU)
template <class T> using mi_gptr = T*;
template <class T> static inline T* mi_global(T* p) { return p; }
#else
template <class T> using mi_gptr = T __attribute__((address_space(1)))*;
template <class T> __device__ __forceinline__ mi_gptr<T> mi_global(T* p) { return (mi_gptr<T>)p; }
#endif

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

// ---- training path helpers (train_bwd.hip, conv_wgrad.hip)
// mean / rstd of the group of channel c of image b from the per-channel statistics [B][C][nt][2]; executed by one wave, result in all lanes
__device__ __forceinline__ void mi_group_moments(const double* stats, int nt, int C, int groups, int HW, float eps, int b, int c, float& mu, float& r) {
    const int lane = threadIdx.x & 63;
    const int cpg = C / groups, g = c / cpg;
    const double* base = stats + ((size_t)b * C + (size_t)g * cpg) * nt * 2;
    double s = 0.0, q = 0.0;
    for (int i = lane; i < cpg * nt; i += 64) { s += base[2 * i]; q += base[2 * i + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    const double n = (double)cpg * (double)HW;
    const double m = s / n;
    double var = q / n - m * m;
    if (var < 0.0) var = 0.0;
    mu = (float)m;
    r = (float)(1.0 / sqrt(var + (double)eps));
}

__device__ __forceinline__ float mi_silu_grad(float v) {          // d/dv (v * sigmoid(v))
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
    return sg * (1.0f + v * (1.0f - sg));
}

// fp16 hi/lo split of four fp32 values for the 3-term MFMA products: hi = fp16(x), lo = fp16(x - hi) (22 mantissa bits kept).
// (A mask + v_cvt_pkrtz formulation was measured 16 % slower on MI355X than plain conversions.  -DMI_SPLIT_PLAIN builds the plain form
//  on the device too.)
typedef _Float16 mi_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 mi_f16x2 __attribute__((ext_vector_type(2)));
// lo halves of two values whose fp16 hi halves are packed in hb: x - float(hi) straight from the packed halves with v_fma_mix_f32
// (fma(f16 -> f32, -1, x); exact, Sterbenz), then one packed conversion.  The emulator build takes the plain form, which computes the
// same bits.  NOTE on inline asm here: the compiler's hazard recogniser does not look inside inline asm (MFMA / transcendental results
// read too early, asm results read too early by an MFMA).  This form is framed by compiler-generated instructions on both sides (the
// packed conversions); v_fma_mixlo_f16 / v_fma_mixhi_f16 (which would save the second conversion) and v_pk_* written as asm returned
// wrong results on the MI355X inside the attention kernel for exactly that reason.
__device__ __forceinline__ unsigned mi_split_lo2(unsigned hb, float x0, float x1) {
#if !defined(MI_SPLIT_PLAIN) && !defined(HIPEMU)
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hb), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hb), "v"(x1));
    const mi_f16x2 l2 = {(_Float16)l0, (_Float16)l1};
    return __builtin_bit_cast(unsigned, l2);
#else
    const mi_f16x2 h2 = __builtin_bit_cast(mi_f16x2, hb);
    const mi_f16x2 l2 = {(_Float16)(x0 - (float)h2[0]), (_Float16)(x1 - (float)h2[1])};
    return __builtin_bit_cast(unsigned, l2);
#endif
}
__device__ __forceinline__ void mi_split_f16(const float (&x)[4], mi_f16x4& hi, mi_f16x4& lo) {
    unsigned hb[2], lb[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const mi_f16x2 h2 = {(_Float16)x[2 * e], (_Float16)x[2 * e + 1]};
        hb[e] = __builtin_bit_cast(unsigned, h2);
        lb[e] = mi_split_lo2(hb[e], x[2 * e], x[2 * e + 1]);
    }
    hi = __builtin_bit_cast(mi_f16x4, make_uint2(hb[0], hb[1]));
    lo = __builtin_bit_cast(mi_f16x4, make_uint2(lb[0], lb[1]));
}

// bf16 activation storage of the reduced-precision configuration (mi_act.st == 1): expansion is a shift, rounding is to nearest even
// (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ float mi_bf16_to_f32(unsigned u16) { return __uint_as_float(u16 << 16); }
__device__ __forceinline__ unsigned mi_f32_to_bf16x2(float a, float b) {        // a in the low half
#if !defined(HIPEMU)
    typedef __bf16 mi_bf16x2 __attribute__((ext_vector_type(2)));
    typedef float mi_cvt_f32x2 __attribute__((ext_vector_type(2)));
    const mi_cvt_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mi_bf16x2));
#else
    auto one = [](float x) -> unsigned {
        const unsigned u = __float_as_uint(x);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;          // NaN stays NaN
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    };
    return one(a) | (one(b) << 16);
#endif
}
__device__ __forceinline__ float4 mi_bf16x4_to_f32(uint2 v) {
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ uint2 mi_f32x4_to_bf16(float4 v) { return make_uint2(mi_f32_to_bf16x2(v.x, v.y), mi_f32_to_bf16x2(v.z, v.w)); }

__device__ __forceinline__ uint2 mi_ldg2u(const void* p) {          // 8 bytes (four bf16) through a global-address-space pointer
    typedef unsigned mi_u32x2 __attribute__((ext_vector_type(2)));
    const mi_u32x2 v = *reinterpret_cast<mi_gptr<const mi_u32x2>>(mi_global(reinterpret_cast<const float*>(p)));
    return make_uint2(v[0], v[1]);
}
__device__ __forceinline__ void mi_stg2u(void* p, uint2 v) {
    typedef unsigned mi_u32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<mi_gptr<mi_u32x2>>(mi_global(reinterpret_cast<float*>(p))) = (mi_u32x2){v.x, v.y};
}

// packed fp32 pairs: the compiler selects v_pk_fma_f32 / v_pk_add_f32 for two-element vectors (two lanes' worth of work per issue
// slot) and keeps track of the hazards around them, which inline asm would not
typedef float mi_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mi_f32x2 mi_pk_fma(mi_f32x2 a, mi_f32x2 b, mi_f32x2 c) {
#if !defined(HIPEMU)
    return __builtin_elementwise_fma(a, b, c);
#else
    return (mi_f32x2){fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])};
#endif
}
__device__ __forceinline__ mi_f32x2 mi_pk_add(mi_f32x2 a, mi_f32x2 b) { return a + b; }

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
            const double* st = a.stats + ((size_t)(ba * a.C + cc) * a.nt) * 2;
            int t = sub;
            for (; t + 3 * TPC < a.nt; t += 4 * TPC) {          // four independent loads in flight, added in tile order
                const double2 v0 = *reinterpret_cast<const double2*>(st + 2 * t);
                const double2 v1 = *reinterpret_cast<const double2*>(st + 2 * (t + TPC));
                const double2 v2 = *reinterpret_cast<const double2*>(st + 2 * (t + 2 * TPC));
                const double2 v3 = *reinterpret_cast<const double2*>(st + 2 * (t + 3 * TPC));
                s += v0.x; q += v0.y;
                s += v1.x; q += v1.y;
                s += v2.x; q += v2.y;
                s += v3.x; q += v3.y;
            }
            for (; t < a.nt; t += TPC) {
                const double2 v = *reinterpret_cast<const double2*>(st + 2 * t);
                s += v.x;
                q += v.y;
            }
            s *= (double)a.scale;
            q *= (double)a.scale * (double)a.scale;
        }
        for (int o = TPC >> 1; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
        if (c < Cin && sub == 0) { chS[c] = s; chQ[c] = q; }
    }
}

// Producer side of the statistics: a work-item accumulates ITS elements of one channel about a local shift c (its first value), so the
// squares it adds are deviations, not magnitudes -- sum (x - c)^2 keeps ~24 bits of the VARIANCE where a plain fp32 sum x^2 keeps 24 bits
// of the second moment (useless once mean^2 >> variance: a large DC offset, a constant image: tests/test_unet.py hostile weights).  The
// conversion to the plain (sum, sum of squares) the consumers add up happens once per work-item in fp64, and so do all further reductions.
struct mi_stat_acc { float c, s, q; int n; };
__device__ __forceinline__ void mi_stat_begin(mi_stat_acc& a, float first) { a.c = first; a.s = 0.0f; a.q = 0.0f; a.n = 0; }
__device__ __forceinline__ void mi_stat_add(mi_stat_acc& a, float y, bool ok) {
    const float d = ok ? y - a.c : 0.0f;
    a.s += d;
    a.q = fmaf(d, d, a.q);
    a.n += ok ? 1 : 0;
}
__device__ __forceinline__ void mi_stat_finish(const mi_stat_acc& a, double& S, double& Q) {
    const double c = (double)a.c, n = (double)a.n, s = (double)a.s;
    S = fma(n, c, s);
    Q = fma(c, fma(n, c, 2.0 * s), (double)a.q);             // sum (d + c)^2 = sum d^2 + 2 c sum d + n c^2
}
// one value per lane: (sum, sum of squares) over the 16 lanes lq = 0 .. 15 of this lane's group (valid at lq == 0) / over the whole wave
// (valid in every lane), shifted by the group's / wave's first lane's value
__device__ __forceinline__ void mi_stat_reduce16(float y, bool ok, int lane, double& S, double& Q) {
    mi_stat_acc a;
    a.c = __shfl(y, lane & 48);
    a.s = ok ? y - a.c : 0.0f;
    a.q = a.s * a.s;
    a.n = ok ? 1 : 0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { a.s += __shfl_xor(a.s, o); a.q += __shfl_xor(a.q, o); a.n += __shfl_xor(a.n, o); }
    mi_stat_finish(a, S, Q);
}
__device__ __forceinline__ void mi_stat_reduce64(float y, bool ok, double& S, double& Q) {
    mi_stat_acc a;
    a.c = __shfl(y, 0);
    a.s = ok ? y - a.c : 0.0f;
    a.q = a.s * a.s;
    a.n = ok ? 1 : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { a.s += __shfl_xor(a.s, o); a.q += __shfl_xor(a.q, o); a.n += __shfl_xor(a.n, o); }
    mi_stat_finish(a, S, Q);
}

// global-memory accessors with the address space pinned (see mi_global)
__device__ __forceinline__ float mi_ldg(const float* p) { return *mi_global(p); }
__device__ __forceinline__ float4 mi_ldg4(const float* p) {
    const f32x4 v = *reinterpret_cast<mi_gptr<const f32x4>>(mi_global(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint4 mi_ldg4u(const void* p) {
    const f32x4 v = *reinterpret_cast<mi_gptr<const f32x4>>(mi_global(reinterpret_cast<const float*>(p)));
    return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
__device__ __forceinline__ void mi_stg(float* p, float v) { *mi_global(p) = v; }
__device__ __forceinline__ void mi_stg4(float* p, float4 v) { *reinterpret_cast<mi_gptr<f32x4>>(mi_global(p)) = (f32x4){v.x, v.y, v.z, v.w}; }

// Buffer-addressed global memory: a wave-uniform base (resource descriptor in SGPRs) + a 32-bit per-lane byte offset + a scalar byte
// offset -- no 64-bit address arithmetic on the VALU per access, and the accesses stay ordinary counted vector-memory operations.
#if defined(HIPEMU)
struct mi_buf { char* base; };
static inline mi_buf mi_make_buf(const void* base) { return {(char*)base}; }
static inline float mi_buf_load_f32(const mi_buf& r, unsigned voff, unsigned soff) { float v; memcpy(&v, r.base + voff + soff, 4); return v; }
static inline f32x4 mi_buf_load_f32x4(const mi_buf& r, unsigned voff, unsigned soff) { f32x4 v; memcpy(&v, r.base + voff + soff, 16); return v; }
static inline void mi_buf_store_f32x4(const mi_buf& r, unsigned voff, unsigned soff, f32x4 v) { memcpy(r.base + voff + soff, &v, 16); }
#else
typedef __amdgpu_buffer_rsrc_t mi_buf;
typedef unsigned mi_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mi_buf mi_make_buf(const void* base) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ float mi_buf_load_f32(const mi_buf& r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ f32x4 mi_buf_load_f32x4(const mi_buf& r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void mi_buf_store_f32x4(const mi_buf& r, unsigned voff, unsigned soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mi_u32x4, v), r, (int)voff, (int)soff, 0);
}
#endif

// Split form of mi_gn_channel_totals for kernels that software-pipeline their loads: mi_gn_totals_issue() puts up to MI_STATS_K
// partial (sum, sumsq) pairs per lane into registers with unconditional loads (clamped addresses), so the kernel can issue its bulk
// loads right behind them and consume the statistics while those are still in flight (vmcnt is in order); mi_gn_totals_finish() adds
// them up in fp64 in a fixed order.  Returns false (nothing issued) when the problem does not fit the fast path -- the caller then
// uses mi_gn_channel_totals.
#define MI_STATS_K 8
struct mi_stats_regs { double2 v[MI_STATS_K]; int c, tpc; float scale; };
__device__ __forceinline__ bool mi_gn_totals_issue(const mi_act& in0, const mi_act& in1, int C0, int Cin, int b, int lane, int nlanes, mi_stats_regs& r) {
    int TPC = 1;
    while (TPC < 64 && TPC * 2 * Cin <= nlanes) TPC *= 2;
    const int nt_max = (Cin > C0 && in1.nt > in0.nt) ? in1.nt : in0.nt;
    if (TPC * Cin > nlanes || nt_max > MI_STATS_K * TPC) return false;
    const int c = lane / TPC, sub = lane % TPC;
    const bool live = c < Cin;
    const bool second = live && c >= C0;
    const mi_act& a = second ? in1 : in0;
    const int cc = live ? (second ? c - C0 : c) : 0;
    const int ba = mi_row_of(b, a.bmod);
    const mi_gptr<const double> st = mi_global(a.stats) + ((size_t)(ba * a.C + cc) * a.nt) * 2;
#pragma unroll
    for (int k = 0; k < MI_STATS_K; ++k) {
        const int t = sub + k * TPC;
        const bool ok = live && t < a.nt;
        const double x = st[2 * (ok ? t : 0)], y = st[2 * (ok ? t : 0) + 1];
        r.v[k] = make_double2(ok ? x : 0.0, ok ? y : 0.0);
    }
    r.c = live ? c : -1;
    r.tpc = TPC;
    r.scale = a.scale;
    return true;
}
__device__ __forceinline__ void mi_gn_totals_finish(const mi_stats_regs& r, int lane, double* chS, double* chQ) {
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int k = 0; k < MI_STATS_K; ++k) { s += r.v[k].x; q += r.v[k].y; }
    s *= (double)r.scale;
    q *= (double)r.scale * (double)r.scale;
    for (int o = r.tpc >> 1; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if (r.c >= 0 && (lane % r.tpc) == 0) { chS[r.c] = s; chQ[r.c] = q; }
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
// a * b rounded on its own: never contracted with a following add into one fma (hipcc's default -ffp-contract=fast-honor-pragmas would), for
// values whose bits have to agree between two kernels that place the multiply differently
__device__ __forceinline__ float mi_mul_rounded(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}

__device__ __forceinline__ float mi_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- inter-workgroup hand-off inside one launch (sampler.hip: the grouped sampler tail).  Per-XCD L2s are not coherent with each other and a CU's
// vector L1 is never refreshed by another CU's stores, so everything one workgroup hands to another goes through WRITE-THROUGH (sc1)
// stores and L1-bypassing (sc1) loads -- placement independent -- and one agent-scope flag per producer: payload stores -> every storing
// wave drains (s_waitcnt vmcnt(0)) -> __syncthreads() -> ONE lane stores the flag; the consumer polls the flag relaxed, then reads the
// payload with sc1 loads (valid because the producer stored sc1).
typedef unsigned long long mi_u64;
// an optimisation barrier on one integer: the compiler must treat the value as unknown from here on (no code is emitted)
#if defined(HIPEMU)
#define MI_OPAQUE(x) asm volatile("" : "+r"(x))
#else
#define MI_OPAQUE(x) asm volatile("" : "+v"(x))
#endif
#if defined(HIPEMU)
#include <sched.h>
static inline mi_u64 mi_agent_load_u64(const mi_u64* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
static inline void mi_agent_store_u64(mi_u64* p, mi_u64 v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
static inline mi_u64 mi_agent_add_u64(mi_u64* p, mi_u64 v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
static inline void mi_agent_store_u32(unsigned* p, unsigned v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
static inline unsigned mi_agent_load_u32(const unsigned* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
// 16 bytes = two 8-byte {tag, value} granules: each half is one atomic access (what the hardware was observed to do for sc1 dwordx4)
static inline f32x4 mi_buf_load_f32x4_sc1(const mi_buf& r, unsigned voff) {
    mi_u64 h[2] = {__atomic_load_n(reinterpret_cast<const mi_u64*>(r.base + voff), __ATOMIC_ACQUIRE), __atomic_load_n(reinterpret_cast<const mi_u64*>(r.base + voff + 8), __ATOMIC_ACQUIRE)};
    f32x4 v; memcpy(&v, h, 16); return v;
}
static inline void mi_buf_store_f32x4_sc1(const mi_buf& r, unsigned voff, f32x4 v) {
    mi_u64 h[2]; memcpy(h, &v, 16);
    __atomic_store_n(reinterpret_cast<mi_u64*>(r.base + voff), h[0], __ATOMIC_RELEASE);
    __atomic_store_n(reinterpret_cast<mi_u64*>(r.base + voff + 8), h[1], __ATOMIC_RELEASE);
}
static inline void mi_drain_vmem() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void mi_sleep() { sched_yield(); }
#else
__device__ __forceinline__ mi_u64 mi_agent_load_u64(const mi_u64* p) { return __hip_atomic_load(mi_global(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mi_agent_store_u64(mi_u64* p, mi_u64 v) { __hip_atomic_store(mi_global(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ mi_u64 mi_agent_add_u64(mi_u64* p, mi_u64 v) { return __hip_atomic_fetch_add(mi_global(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mi_agent_store_u32(unsigned* p, unsigned v) { __hip_atomic_store(mi_global(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned mi_agent_load_u32(const unsigned* p) { return __hip_atomic_load(mi_global(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// aux / cache-policy bit 4 = sc1 on gfx940+ (bit 0 = sc0, bit 1 = nt)
__device__ __forceinline__ f32x4 mi_buf_load_f32x4_sc1(const mi_buf& r, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 16));
}
__device__ __forceinline__ void mi_buf_store_f32x4_sc1(const mi_buf& r, unsigned voff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mi_u32x4, v), r, (int)voff, 0, 16);
}
// inline asm: the compiler cannot drop or move it (it does drop builtin waits it believes redundant)
__device__ __forceinline__ void mi_drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void mi_sleep() { __builtin_amdgcn_s_sleep(2); }
#endif

int mi_conv_rp_launch(const mi_conv_params& p, hipStream_t st);     // conv_rp.hip
int mi_conv_wide_launch(const mi_conv_params& p, hipStream_t st);   // conv_wide.hip
int mi_conv_stripe_launch(const mi_conv_params& p, hipStream_t st); // conv_stripe.hip

// host-side error plumbing (capi.hip)
void mi_set_error(const char* fmt, ...);
int mi_check_launch(const char* what);
