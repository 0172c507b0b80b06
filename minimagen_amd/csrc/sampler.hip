// Sampler epilogue kernels: K11 (CFG combine + x0), K12 (bit-exact dynamic-threshold quantile),
// K13/K15 (posterior step, final clamp), K14 (cubic resize + low-res augmentation) and the
// counter-based normal generator.  All elementwise math keeps the reference's operation order
// with un-fused multiplies/adds (__fmul_rn/__fadd_rn) so the only differences to the reference's
// fp32 results come from the U-Net's summation order.
#include "common.hip.h"

// hipcc contracts a*b+c into FMA by default and its __fmul_rn/__fadd_rn are plain operators: switch contraction
// off for this file so the elementwise sampler math rounds exactly like the reference's separate torch ops.
#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------ K11 epilogue
template <bool HIST>
__global__ __launch_bounds__(256) void cfg_x0_kernel(const mi_cfg_x0_params p) {
    __shared__ unsigned lh[HIST ? MI_Q_BINS : 1];      // pass 0 of the radix select (bits 30..20 of |x0|), fused into the producer of x0
    const int b = blockIdx.y;
    const int t = p.t_state ? *p.t_state - p.t_off : 0;
    const float ca = p.coef ? p.coef[t * 8 + 0] : 0.0f, cb = p.coef ? p.coef[t * 8 + 1] : 0.0f;
    if constexpr (HIST) {
        for (int i = threadIdx.x; i < MI_Q_BINS; i += 256) lh[i] = 0u;
        __syncthreads();
    }
    auto one = [&](float c, float nl, float xt, float& pred, float& x0) {
        pred = c;
        if (p.two) pred = __fadd_rn(nl, __fmul_rn(__fsub_rn(c, nl), p.cond_scale));     // Unet.py:506
        x0 = __fsub_rn(__fmul_rn(ca, xt), __fmul_rn(cb, pred));                          // diffusion_model.py:159-162
        if constexpr (HIST) { if (p.x0) atomicAdd(&lh[__float_as_uint(fabsf(x0)) >> 20], 1u); }
    };
    if ((p.n & 3) == 0) {                           // 16-byte accesses (every image size the cascade uses)
        const size_t ob = (size_t)b * p.n, on = (size_t)(b + p.B) * p.n;
        for (int q = blockIdx.x * 256 + threadIdx.x; q < (p.n >> 2); q += gridDim.x * 256) {
            const float4 c = mi_ldg4(p.pred2 + ob + 4 * q);
            const float4 nl = p.two ? mi_ldg4(p.pred2 + on + 4 * q) : c;
            const float4 xt = p.x0 ? mi_ldg4(p.x_t + ob + 4 * q) : c;
            float4 pr, x0;
            one(c.x, nl.x, xt.x, pr.x, x0.x); one(c.y, nl.y, xt.y, pr.y, x0.y); one(c.z, nl.z, xt.z, pr.z, x0.z); one(c.w, nl.w, xt.w, pr.w, x0.w);
            if (p.pred_out) mi_stg4(p.pred_out + ob + 4 * q, pr);
            if (p.x0) mi_stg4(p.x0 + ob + 4 * q, x0);
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < p.n; i += gridDim.x * 256) {
            const float c = p.pred2[(size_t)b * p.n + i];
            const float nl = p.two ? p.pred2[(size_t)(b + p.B) * p.n + i] : c;
            const float xt = p.x0 ? p.x_t[(size_t)b * p.n + i] : c;
            float pred, x0;
            one(c, nl, xt, pred, x0);
            if (p.pred_out) p.pred_out[(size_t)b * p.n + i] = pred;
            if (p.x0) p.x0[(size_t)b * p.n + i] = x0;
        }
    }
    if constexpr (HIST) {
        __syncthreads();
        // both order statistics share the pass-0 histogram (no prefix yet): selector slot 0 only, read for both
        unsigned* gh = p.hist0 + ((size_t)b * 2) * MI_Q_BINS;
        for (int i = threadIdx.x; i < MI_Q_BINS; i += 256) {
            const unsigned v = lh[i];
            if (v) atomicAdd(&gh[i], v);
        }
    }
}

// ------------------------------------------------------------------ K12 radix select
// digit layout of a non-negative float's bit pattern (bit 31 = 0): pass 0 -> bits 30..20, pass 1 -> 19..9, pass 2 -> 8..0
__device__ __forceinline__ int q_shift(int pass) { return pass == 0 ? 20 : (pass == 1 ? 9 : 0); }
__device__ __forceinline__ int q_bits(int pass) { return pass == 2 ? 9 : 11; }

// Block-wide: locate the bin of `hist` (MI_Q_BINS entries) that holds 0-based rank r; returns bin and the
// rank inside the bin.  All 256 work-items call it; result broadcast through LDS.
__device__ void q_find_bin(const unsigned* hist, unsigned r, int* sh_scratch, unsigned& bin_out, unsigned& r_out, bool active = true) {
    // `active`: the first 256 work-items of the workgroup scan (8 bins each); larger workgroups pass false for the rest, which only
    // take part in the barriers and receive the broadcast
    unsigned* wsum = reinterpret_cast<unsigned*>(sh_scratch);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned c[8], tot = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { c[k] = active ? hist[tid * 8 + k] : 0u; tot += c[k]; }
    unsigned inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (active && lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave && w < 4; ++w) base += wsum[w];
    unsigned excl = base + inc - tot;
    if (active && r >= excl && r < excl + tot) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (r < excl + c[k]) { wsum[4] = (unsigned)(tid * 8 + k); wsum[5] = r - excl; break; }
            excl += c[k];
        }
    }
    __syncthreads();
    bin_out = wsum[4];
    r_out = wsum[5];
    __syncthreads();
}

template <int PASS>
__global__ __launch_bounds__(256) void quantile_hist_kernel(const mi_quantile_params p) {
    __shared__ unsigned lh[2][MI_Q_BINS];
    __shared__ int scratch[8];
    const int tid = threadIdx.x, b = blockIdx.y;
    // resolve the prefixes chosen by the earlier passes (every workgroup redoes this tiny scan)
    unsigned prefix[2] = {0u, 0u}, rk[2] = {(unsigned)p.k_lo, (unsigned)p.k_hi};
#pragma unroll
    for (int ps = 0; ps < PASS; ++ps) {
#pragma unroll
        for (int sel = 0; sel < 2; ++sel) {
            unsigned bin, rr;
            q_find_bin(p.hist + (((size_t)ps * p.B + b) * 2 + (ps == 0 ? 0 : sel)) * MI_Q_BINS, rk[sel], scratch, bin, rr);      // pass 0: shared slot
            prefix[sel] = (prefix[sel] << q_bits(ps)) | bin;
            rk[sel] = rr;
        }
    }
    for (int i = tid; i < 2 * MI_Q_BINS; i += 256) (&lh[0][0])[i] = 0u;
    __syncthreads();
    const int shift = q_shift(PASS), nb = q_bits(PASS);
    const unsigned mask = (1u << nb) - 1u;
    const float* xb = p.x0 + (size_t)b * p.n;
    auto count = [&](float v) {
        const unsigned key = __float_as_uint(fabsf(v));
        const unsigned hi = PASS == 0 ? 0u : (key >> (shift + nb));
        const unsigned bin = (key >> shift) & mask;
        if (hi == prefix[0]) atomicAdd(&lh[0][bin], 1u);
        if (PASS > 0 && hi == prefix[1]) atomicAdd(&lh[1][bin], 1u);              // pass 0: one histogram serves both order statistics
    };
    if ((p.n & 3) == 0) {
        for (int q = blockIdx.x * 256 + tid; q < (p.n >> 2); q += gridDim.x * 256) {
            const float4 v = mi_ldg4(xb + 4 * q);
            count(v.x); count(v.y); count(v.z); count(v.w);
        }
    } else {
        for (int i = blockIdx.x * 256 + tid; i < p.n; i += gridDim.x * 256) count(xb[i]);
    }
    __syncthreads();
    unsigned* gh = p.hist + (((size_t)PASS * p.B + b) * 2) * MI_Q_BINS;
    for (int i = tid; i < (PASS == 0 ? 1 : 2) * MI_Q_BINS; i += 256) {
        const unsigned v = (&lh[0][0])[i];
        if (v) atomicAdd(&gh[i], v);
    }
}

__global__ __launch_bounds__(256) void quantile_finish_kernel(const mi_quantile_params p) {
    __shared__ int scratch[8];
    const int b = blockIdx.x;
    unsigned prefix[2] = {0u, 0u}, rk[2] = {(unsigned)p.k_lo, (unsigned)p.k_hi};
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {
#pragma unroll
        for (int sel = 0; sel < 2; ++sel) {
            unsigned bin, rr;
            q_find_bin(p.hist + (((size_t)ps * p.B + b) * 2 + (ps == 0 ? 0 : sel)) * MI_Q_BINS, rk[sel], scratch, bin, rr);
            prefix[sel] = (prefix[sel] << q_bits(ps)) | bin;
            rk[sel] = rr;
        }
    }
    if (threadIdx.x == 0) {
        // torch.quantile returns NaN for a row that contains a NaN (not the order statistic): NaN bit patterns sit above infinity,
        // i.e. in the top bins of pass 0 (0x7F9.. are NaN only; arithmetic produces the canonical 0x7FC00000)
        const unsigned* h0 = p.hist + ((size_t)b * 2) * MI_Q_BINS;
        unsigned nan_count = 0;
        for (int k = 0x7F9; k < MI_Q_BINS; ++k) nan_count += h0[k];
        float a = __uint_as_float(prefix[0]), bb = __uint_as_float(prefix[1]);
        if (nan_count) a = bb = __uint_as_float(0x7FC00000u);
        const float d = __fsub_rn(bb, a);
        // ATen lerp: weight < 0.5 ? a + w*d : b - d*(1-w), multiply-add fused
        const float s = (fabsf(p.w) < 0.5f) ? fmaf(p.w, d, a) : fmaf(__fsub_rn(p.w, 1.0f), d, bb);
        p.s_out[b] = s;
        if (p.v_out) { p.v_out[2 * b] = a; p.v_out[2 * b + 1] = bb; }
    }
    if (p.self_cleaning) {       // leave this image's counters zeroed for the next denoising step (no memset launch)
        __syncthreads();
        for (int ps = 0; ps < 3; ++ps) {
            unsigned* gh = p.hist + (((size_t)ps * p.B + b) * 2) * MI_Q_BINS;
            for (int i = threadIdx.x; i < 2 * MI_Q_BINS; i += 256) gh[i] = 0u;
        }
    }
}

// ------------------------------------------------------------------ Philox4x32-10 + Box-Muller
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const unsigned hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const unsigned hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const unsigned n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
// four N(0,1) draws for (seed, sample, stream, quad index)
__device__ __forceinline__ void randn4(unsigned long long seed, unsigned sample, unsigned stream, unsigned quad, float (&z)[4]) {
    unsigned c[4] = {quad, sample, stream, 0x4D494E49u};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    const float u0 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u1 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[2] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u3 = ((float)(c[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    const float a0 = 6.28318530717958647692f * u1, a1 = 6.28318530717958647692f * u3;
    z[0] = r0 * cosf(a0); z[1] = r0 * sinf(a0);
    z[2] = r1 * cosf(a1); z[3] = r1 * sinf(a1);
}

__global__ __launch_bounds__(256) void randn_fill_kernel(float* out, int n, unsigned long long seed, int sample0, int stream_id) {
    const int b = blockIdx.y;
    const int nq = (n + 3) / 4;
    for (int qd = blockIdx.x * 256 + threadIdx.x; qd < nq; qd += gridDim.x * 256) {
        float z[4];
        randn4(seed, (unsigned)(sample0 + b), (unsigned)stream_id, (unsigned)qd, z);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * qd + k < n) out[(size_t)b * n + 4 * qd + k] = z[k];
    }
}

// ------------------------------------------------------------------ K13
__global__ __launch_bounds__(256) void posterior_kernel(const mi_posterior_params p) {
    const int b = blockIdx.y;
    const int t = *p.t_state - p.t_off;
    const float c1 = p.coef[t * 8 + 2], c2 = p.coef[t * 8 + 3], sigma = p.coef[t * 8 + 4];
    const float sq = p.s_q[b];
    const float s = (sq < 1.0f) ? 1.0f : sq;                                     // Imagen.py:320 clamp_(min=1.): a NaN threshold stays NaN, as in torch
    const int k = (p.T - 1) - t;
    const float* nz = p.noise ? p.noise + ((size_t)k * p.B + b) * p.n : nullptr;
    const int nq = (p.n + 3) / 4;
    const bool vec = (p.n & 3) == 0;
    auto one = [&](float x0, float x, float z) {
        x0 = __fdiv_rn(x0 != x0 ? x0 : fminf(fmaxf(x0, -s), s), s);               // Imagen.py:323 (torch.clamp propagates NaN)
        const float mean = __fadd_rn(__fmul_rn(c1, x0), __fmul_rn(c2, x));          // diffusion_model.py:118-121
        return __fadd_rn(mean, __fmul_rn(sigma, z));                               // Imagen.py:370
    };
    for (int qd = blockIdx.x * 256 + threadIdx.x; qd < nq; qd += gridDim.x * 256) {
        float z[4];
        if (nz) {
            if (vec) { const float4 v = mi_ldg4(nz + 4 * qd); z[0] = v.x; z[1] = v.y; z[2] = v.z; z[3] = v.w; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = (4 * qd + e < p.n) ? nz[4 * qd + e] : 0.0f;
            }
        } else {
            randn4(p.seed_dev ? *p.seed_dev : p.seed, (unsigned)(p.sample0 + b), (unsigned)(p.stream_base + k), (unsigned)qd, z);
        }
        if (vec) {
            const size_t o = (size_t)b * p.n + 4 * qd;
            const float4 x0 = mi_ldg4(p.x0 + o), x = mi_ldg4(p.x + o);
            mi_stg4(p.x + o, make_float4(one(x0.x, x.x, z[0]), one(x0.y, x.y, z[1]), one(x0.z, x.z, z[2]), one(x0.w, x.w, z[3])));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * qd + e;
                if (i < p.n) {
                    const size_t o = (size_t)b * p.n + i;
                    p.x[o] = one(p.x0[o], p.x[o], z[e]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ K11 epilogue + K12 + K13 in one launch (small images)
// One workgroup of 1024 work-items per image; every work-item keeps its quads of x0 in registers, the three radix passes run on
// histograms in LDS.  Operation order per element as in cfg_x0_kernel / quantile_*_kernel / posterior_kernel: bit-identical results.
#ifndef SS_THREADS
#define SS_THREADS 1024
#endif
constexpr int SS_NT = SS_THREADS, SS_MAXQ = MI_SAMPLER_SMALL_N / 4 / SS_NT;      // quads per work-item
__global__ __launch_bounds__(SS_NT) void sampler_small_kernel(const mi_cfg_x0_params c, const mi_quantile_params q, const mi_posterior_params pp) {
    __shared__ unsigned lh[2][MI_Q_BINS];
    __shared__ int scratch[8];
    __shared__ unsigned nan_sh;
    const int tid = threadIdx.x, b = blockIdx.x, n = c.n, nq = (n + 3) / 4;
    const int t = *c.t_state - c.t_off;
    const float ca = c.coef[t * 8 + 0], cb = c.coef[t * 8 + 1];
    const float c1 = c.coef[t * 8 + 2], c2 = c.coef[t * 8 + 3], sigma = c.coef[t * 8 + 4];
    const bool vec = (n & 3) == 0;
    const size_t ob = (size_t)b * n, on = (size_t)(b + c.B) * n;
    float x0v[SS_MAXQ][4], xtv[SS_MAXQ][4];
    for (int i = tid; i < 2 * MI_Q_BINS; i += SS_NT) (&lh[0][0])[i] = 0u;
    if (tid == 0) nan_sh = 0u;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SS_MAXQ; ++u) {
        const int qd = tid + u * SS_NT;
        float cc[4], nl[4], pr[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { cc[e] = nl[e] = xtv[u][e] = x0v[u][e] = 0.0f; }
        if (qd < nq) {
            if (vec) {
                const float4 a = mi_ldg4(c.pred2 + ob + 4 * qd), xx = mi_ldg4(c.x_t + ob + 4 * qd);
                const float4 d = c.two ? mi_ldg4(c.pred2 + on + 4 * qd) : a;
                cc[0] = a.x; cc[1] = a.y; cc[2] = a.z; cc[3] = a.w; nl[0] = d.x; nl[1] = d.y; nl[2] = d.z; nl[3] = d.w;
                xtv[u][0] = xx.x; xtv[u][1] = xx.y; xtv[u][2] = xx.z; xtv[u][3] = xx.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * qd + e < n) { cc[e] = c.pred2[ob + 4 * qd + e]; nl[e] = c.two ? c.pred2[on + 4 * qd + e] : cc[e]; xtv[u][e] = c.x_t[ob + 4 * qd + e]; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pr[e] = cc[e];
                if (c.two) pr[e] = __fadd_rn(nl[e], __fmul_rn(__fsub_rn(cc[e], nl[e]), c.cond_scale));       // Unet.py:506
                x0v[u][e] = __fsub_rn(__fmul_rn(ca, xtv[u][e]), __fmul_rn(cb, pr[e]));                      // diffusion_model.py:159-162
                if (4 * qd + e < n) {
                    atomicAdd(&lh[0][__float_as_uint(fabsf(x0v[u][e])) >> 20], 1u);
                    if (c.pred_out) c.pred_out[ob + 4 * qd + e] = pr[e];
                    if (c.x0) c.x0[ob + 4 * qd + e] = x0v[u][e];
                }
            }
        }
    }
    __syncthreads();
    // torch.quantile returns NaN for a row that contains a NaN: NaN patterns sit in the top bins of pass 0 (as quantile_finish_kernel)
    if (tid < MI_Q_BINS - 0x7F9) { const unsigned v = lh[0][0x7F9 + tid]; if (v) atomicAdd(&nan_sh, v); }
    unsigned prefix[2] = {0u, 0u}, rk[2] = {(unsigned)q.k_lo, (unsigned)q.k_hi};
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {
        if (ps > 0) {
            // histogram of the next digit over the elements that carry each selected prefix
            __syncthreads();
            for (int i = tid; i < 2 * MI_Q_BINS; i += SS_NT) (&lh[0][0])[i] = 0u;
            __syncthreads();
            const int shift = q_shift(ps), nb = q_bits(ps);
            const unsigned mask = (1u << nb) - 1u;
#pragma unroll
            for (int u = 0; u < SS_MAXQ; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * (tid + u * SS_NT) + e;
                    if (i < n) {
                        const unsigned key = __float_as_uint(fabsf(x0v[u][e]));
                        const unsigned hi = key >> (shift + nb), bin = (key >> shift) & mask;
                        if (hi == prefix[0]) atomicAdd(&lh[0][bin], 1u);
                        if (hi == prefix[1]) atomicAdd(&lh[1][bin], 1u);
                    }
                }
            __syncthreads();
        }
#pragma unroll
        for (int sel = 0; sel < 2; ++sel) {
            unsigned bin, rr;
            q_find_bin(&lh[ps == 0 ? 0 : sel][0], rk[sel], scratch, bin, rr, tid < 256);
            prefix[sel] = (prefix[sel] << q_bits(ps)) | bin;
            rk[sel] = rr;
        }
    }
    float a = __uint_as_float(prefix[0]), bb = __uint_as_float(prefix[1]);
    if (nan_sh) a = bb = __uint_as_float(0x7FC00000u);
    const float d = __fsub_rn(bb, a);
    const float sq = (fabsf(q.w) < 0.5f) ? fmaf(q.w, d, a) : fmaf(__fsub_rn(q.w, 1.0f), d, bb);      // ATen lerp (fused multiply-add)
    if (tid == 0) {
        if (q.s_out) q.s_out[b] = sq;
        if (q.v_out) { q.v_out[2 * b] = a; q.v_out[2 * b + 1] = bb; }
    }
    const float s = (sq < 1.0f) ? 1.0f : sq;                                     // Imagen.py:320 clamp_(min=1.): a NaN threshold stays NaN
    const int k = (pp.T - 1) - t;
    const float* nz = pp.noise ? pp.noise + ((size_t)k * pp.B + b) * n : nullptr;
#pragma unroll
    for (int u = 0; u < SS_MAXQ; ++u) {
        const int qd = tid + u * SS_NT;
        if (qd >= nq) continue;
        float z[4];
        if (nz) {
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = (4 * qd + e < n) ? nz[4 * qd + e] : 0.0f;
        } else {
            randn4(pp.seed_dev ? *pp.seed_dev : pp.seed, (unsigned)(pp.sample0 + b), (unsigned)(pp.stream_base + k), (unsigned)qd, z);
        }
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x0 = x0v[u][e];
            x0 = __fdiv_rn(x0 != x0 ? x0 : fminf(fmaxf(x0, -s), s), s);                  // Imagen.py:323 (torch.clamp propagates NaN)
            const float mean = __fadd_rn(__fmul_rn(c1, x0), __fmul_rn(c2, xtv[u][e]));     // diffusion_model.py:118-121
            r[e] = __fadd_rn(mean, __fmul_rn(sigma, z[e]));                               // Imagen.py:370
        }
        if (vec) mi_stg4(pp.x + ob + 4 * qd, make_float4(r[0], r[1], r[2], r[3]));
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (4 * qd + e < n) pp.x[ob + 4 * qd + e] = r[e];
        }
    }
}

__global__ void step_advance_kernel(int* t_state, long long* times, int B, int set, int value) {
    const int t = set == 1 ? value : (*t_state - (set == 2 ? value : 1));
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += blockDim.x) times[b] = (long long)t;
    if (threadIdx.x == 0) *t_state = t;
}

__global__ __launch_bounds__(256) void finalize_kernel(const float* x, float* out, long long total, int unnormalize) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const float v = x[i] != x[i] ? x[i] : fminf(fmaxf(x[i], -1.0f), 1.0f);      // torch.clamp propagates NaN
        out[i] = unnormalize ? __fmul_rn(__fadd_rn(v, 1.0f), 0.5f) : v;
    }
}

__global__ __launch_bounds__(256) void lowres_augment_kernel(const float* img, const float* noise, float* out, long long total, float a, float b, int normalize) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        float v = img[i];
        if (noise) v = __fadd_rn(__fmul_rn(a, v), __fmul_rn(b, noise[i]));       // diffusion_model.py:142-147
        if (normalize) v = __fsub_rn(__fmul_rn(v, 2.0f), 1.0f);                  // helpers.py:105-110 via Imagen.py:393
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void resize_kernel(const mi_resize_params p) {
    const int plane = blockIdx.y;
    const float* src = p.in + (size_t)plane * p.Hin * p.Win;
    float* dst = p.out + (size_t)plane * p.Hout * p.Wout;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < p.Hout * p.Wout; o += gridDim.x * 256) {
        const int oy = o / p.Wout, ox = o % p.Wout;
        float acc = 0.0f;
        for (int kx = 0; kx < p.KW; ++kx) {
            const int sx = p.idx_w[ox * p.KW + kx];
            float col = 0.0f;                                   // H pass value at (oy, sx)
            for (int ky = 0; ky < p.KH; ++ky) {
                const float v = __fmul_rn(src[(size_t)p.idx_h[oy * p.KH + ky] * p.Win + sx], p.w_h[oy * p.KH + ky]);
                col = ky == 0 ? v : __fadd_rn(col, v);
            }
            const float v = __fmul_rn(col, p.w_w[ox * p.KW + kx]);
            acc = kx == 0 ? v : __fadd_rn(acc, v);
        }
        dst[o] = acc;
    }
}

#ifndef Q_WGS
#define Q_WGS 16
#endif
inline int grid_for(long long n, int cap = 2048) {
    long long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}


// ------------------------------------------------------------------ K11 epilogue + K12 + K13 in one launch (large images)
// The tail of a denoising step for images too large for one workgroup: G workgroups of 1024 work-items per image, every work-item keeps its
// SG_MAXQ quads of x0 (and x_t) in registers from the guidance combine to the posterior draw; the three radix passes are LDS histograms
// per workgroup, added into per-image histograms with integer agent-scope atomics, and a counter barrier per pass among the G workgroups of
// the image.  Replaces five launches (cfg_x0 + pass 0, pass 1, pass 2, finish, posterior: 71 us at 256^2, B = 32) and two round trips of x0
// through memory.  Same operations in the same order per element: bit-identical to the separate kernels.
// Inter-workgroup protocol: the workgroups of an image are claimed by ticket after they are resident (no cooperative launch, no deadlock
// inside one launch); everything exchanged is touched by agent-scope atomics / sc1 accesses only; the histograms are double-buffered by
// launch parity and the idle copy is zeroed for the next launch; every spin is bounded.
// FAIL-STOP: a workgroup whose wait runs out sets the sticky error word (sync + 8), overwrites ITS part of x_t with NaN and leaves; every
// workgroup that finds the word set -- its peers at their next poll, every workgroup of every later launch on this sync buffer at its
// start -- does the same without waiting.  So a launch that could not complete never leaves a stale or half-written image behind: the
// image (and everything sampled from it) is NaN, which the reference's own pipeline would return for a NaN state as well, and the host
// raises at its next status poll (Imagen: the word is copied to pinned host memory at the end of every call) and re-zeroes the buffer.
// Header of `sync`: [0] u64 ticket | [8] u32 error | [12] u32 knobs: bits 0..30 spin limit (0: SG_SPIN_LIMIT), bit 31 fault injection (tests
// only: the last workgroup of image 0 never arrives at radix pass 1) -- error and knobs are ONE 8-byte load, in flight with the ticket atomic.
constexpr int SG_NT = 1024, SG_MAXQ = 6;
constexpr unsigned SG_SPIN_LIMIT = 1u << 22;
struct sg_layout { long long counters, hist, total; };
__host__ __device__ inline sg_layout sg_sync_layout(int B) {
    sg_layout l;
    l.counters = 64;                                                        // [0] ticket (u64), [8] error word (u32)
    l.hist = l.counters + (((long long)B * 8 + 63) & ~63ll);               // counters: u64 [B]
    l.total = l.hist + (long long)2 * B * 5 * MI_Q_BINS * 4;               // hist: u32 [parity][B][5 = pass 0 | pass 1 x 2 | pass 2 x 2][MI_Q_BINS]
    return l;
}

__global__ __launch_bounds__(SG_NT) void sampler_group_kernel(const mi_cfg_x0_params c, const mi_quantile_params q, const mi_posterior_params pp, char* sync, const int G) {
    __shared__ unsigned lh[2][MI_Q_BINS];
    __shared__ __attribute__((aligned(16))) unsigned hc[2][MI_Q_BINS];
    __shared__ int scratch[8];
    __shared__ unsigned nan_sh;
    __shared__ mi_u64 sTicket;
    __shared__ int sAbort;
    __shared__ unsigned sLimit, sFault;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = c.n, nq = n >> 2;
    unsigned* const errw = reinterpret_cast<unsigned*>(sync + 8);
    if (tid == 0) {
        const mi_u64 hdr = mi_agent_load_u64(reinterpret_cast<const mi_u64*>(sync + 8));      // error | knobs
        sTicket = mi_agent_add_u64(reinterpret_cast<mi_u64*>(sync), 1ull);
        sAbort = (unsigned)hdr != 0u;                           // an earlier launch on this buffer failed: fail-stop, see above
        const unsigned knobs = (unsigned)(hdr >> 32), lim = knobs & 0x7fffffffu;
        sLimit = lim ? lim : SG_SPIN_LIMIT;
        sFault = knobs >> 31;
        nan_sh = 0u;
    }
    for (int i = tid; i < 2 * MI_Q_BINS; i += SG_NT) (&lh[0][0])[i] = 0u;
    __syncthreads();
    const sg_layout lay = sg_sync_layout(c.B);
    const mi_u64 tk = sTicket, Gt = (mi_u64)c.B * (mi_u64)G, seq = tk / Gt;
    const int local = (int)(tk - seq * Gt), b = local / G, g = local - b * G, par = (int)(seq & 1);
    mi_u64* const counter = reinterpret_cast<mi_u64*>(sync + lay.counters) + b;
    unsigned* const H = reinterpret_cast<unsigned*>(sync + lay.hist) + ((size_t)par * c.B + b) * 5 * MI_Q_BINS;
    const mi_buf hbuf = mi_make_buf(H);
    const mi_buf zbuf = mi_make_buf(sync + lay.hist + ((size_t)(par ^ 1) * c.B + b) * 5 * MI_Q_BINS * 4);
    // the idle parity's histograms of this image, zeroed for the next launch (write-through: the next launch's atomics find zeros in memory)
    for (int i = g * SG_NT + tid; i < 5 * MI_Q_BINS / 4; i += G * SG_NT) mi_buf_store_f32x4_sc1(zbuf, (unsigned)i * 16u, (f32x4){0.f, 0.f, 0.f, 0.f});

    const int t = *c.t_state - c.t_off;
    const float ca = c.coef[t * 8 + 0], cb = c.coef[t * 8 + 1];
    const float c1 = c.coef[t * 8 + 2], c2 = c.coef[t * 8 + 3], sigma = c.coef[t * 8 + 4];
    const size_t ob = (size_t)b * n, on = (size_t)(b + c.B) * n;
    const int q0 = g * SG_NT * SG_MAXQ;                        // this workgroup's quads: q0 + tid + u * SG_NT
    // fail-stop: this workgroup's part of the image becomes NaN (never a stale or half-finished x_t)
    auto poison = [&]() {
        const float qn = __uint_as_float(0x7FC00000u);
        for (int u = 0; u < SG_MAXQ; ++u) {
            const int qd = q0 + tid + u * SG_NT;
            if (qd < nq) mi_stg4(pp.x + ob + 4 * qd, make_float4(qn, qn, qn, qn));
        }
    };
    if (sAbort) { poison(); return; }
    float x0v[SG_MAXQ][4], xtv[SG_MAXQ][4];
#pragma unroll
    for (int u = 0; u < SG_MAXQ; ++u) {
        const int qd = q0 + tid + u * SG_NT;
        float cc[4], nl[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { cc[e] = nl[e] = xtv[u][e] = x0v[u][e] = 0.0f; }
        if (qd < nq) {
            const float4 a = mi_ldg4(c.pred2 + ob + 4 * qd), xx = mi_ldg4(c.x_t + ob + 4 * qd);
            const float4 d = c.two ? mi_ldg4(c.pred2 + on + 4 * qd) : a;
            cc[0] = a.x; cc[1] = a.y; cc[2] = a.z; cc[3] = a.w; nl[0] = d.x; nl[1] = d.y; nl[2] = d.z; nl[3] = d.w;
            xtv[u][0] = xx.x; xtv[u][1] = xx.y; xtv[u][2] = xx.z; xtv[u][3] = xx.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pr = cc[e];
                if (c.two) pr = __fadd_rn(nl[e], __fmul_rn(__fsub_rn(cc[e], nl[e]), c.cond_scale));       // Unet.py:506
                x0v[u][e] = __fsub_rn(__fmul_rn(ca, xtv[u][e]), __fmul_rn(cb, pr));                       // diffusion_model.py:159-162
                atomicAdd(&lh[0][__float_as_uint(fabsf(x0v[u][e])) >> 20], 1u);
                if (c.pred_out) c.pred_out[ob + 4 * qd + e] = pr;
                if (c.x0) c.x0[ob + 4 * qd + e] = x0v[u][e];
            }
        }
    }
    // one pass of the radix select across the image's workgroups: add this workgroup's `nh` LDS histograms into the image's, wait for all
    // G workgroups, copy the totals into LDS (hc)
    auto exchange = [&](int phase, int slot0, int nh) -> bool {
        __syncthreads();
        for (int i = tid; i < nh * MI_Q_BINS; i += SG_NT) {
            const unsigned v = (&lh[0][0])[i];
            if (v) atomicAdd(&H[(size_t)slot0 * MI_Q_BINS + i], v);
        }
        mi_drain_vmem();                                   // every wave's atomics are performed before the arrival is counted
        __syncthreads();
        if (tid == 0 && !(sFault == 1u && phase == 1 && b == 0 && g == G - 1)) mi_agent_add_u64(counter, 1ull);
        if (wave == 0) {
            const mi_u64 target = (seq * 3 + (mi_u64)phase + 1) * (mi_u64)G;
            const unsigned limit = sLimit;
            for (unsigned spins = 0;;) {
                const mi_u64 v = mi_agent_load_u64(counter);
                if (__all(v >= target)) break;
                const bool peer_failed = (spins & 63u) == 63u && mi_agent_load_u32(errw) != 0u;
                if (++spins > limit || peer_failed) {
                    if (lane == 0) { if (!peer_failed) mi_agent_store_u32(errw, 0x300u + (unsigned)phase); sAbort = 1; }
                    break;
                }
                mi_sleep();
            }
        }
        __syncthreads();
        if (sAbort) return false;
        for (int i = tid; i < nh * MI_Q_BINS / 4; i += SG_NT)
            reinterpret_cast<f32x4*>(&hc[0][0])[i] = mi_buf_load_f32x4_sc1(hbuf, (unsigned)((slot0 * MI_Q_BINS + 4 * i) * 4));
        __syncthreads();
        return true;
    };
    if (!exchange(0, 0, 1)) { poison(); return; }
    // torch.quantile returns NaN for a row that contains a NaN: NaN patterns sit in the top bins of pass 0 (as quantile_finish_kernel)
    if (tid < MI_Q_BINS - 0x7F9) { const unsigned v = hc[0][0x7F9 + tid]; if (v) atomicAdd(&nan_sh, v); }
    unsigned prefix[2] = {0u, 0u}, rk[2] = {(unsigned)q.k_lo, (unsigned)q.k_hi};
    bool one = true;
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {
        if (ps > 0) {
            __syncthreads();
            for (int i = tid; i < 2 * MI_Q_BINS; i += SG_NT) (&lh[0][0])[i] = 0u;
            __syncthreads();
            const int shift = q_shift(ps), nb = q_bits(ps);
            const unsigned mask = (1u << nb) - 1u;
            one = prefix[0] == prefix[1];          // both ranks in one bin so far (the usual case: neighbours): one histogram serves both
#pragma unroll
            for (int u = 0; u < SG_MAXQ; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (q0 + tid + u * SG_NT < nq) {
                        const unsigned key = __float_as_uint(fabsf(x0v[u][e]));
                        const unsigned hi = key >> (shift + nb), bin = (key >> shift) & mask;
                        if (hi == prefix[0]) atomicAdd(&lh[0][bin], 1u);
                        if (!one && hi == prefix[1]) atomicAdd(&lh[1][bin], 1u);
                    }
                }
            if (!exchange(ps, 2 * ps - 1, one ? 1 : 2)) { poison(); return; }
        }
#pragma unroll
        for (int sel = 0; sel < 2; ++sel) {
            unsigned bin, rr;
            q_find_bin(&hc[one ? 0 : sel][0], rk[sel], scratch, bin, rr, tid < 256);
            prefix[sel] = (prefix[sel] << q_bits(ps)) | bin;
            rk[sel] = rr;
        }
    }
    float a = __uint_as_float(prefix[0]), bb = __uint_as_float(prefix[1]);
    if (nan_sh) a = bb = __uint_as_float(0x7FC00000u);
    const float d = __fsub_rn(bb, a);
    const float sq = (fabsf(q.w) < 0.5f) ? fmaf(q.w, d, a) : fmaf(__fsub_rn(q.w, 1.0f), d, bb);      // ATen lerp (fused multiply-add)
    if (tid == 0 && g == 0) {
        if (q.s_out) q.s_out[b] = sq;
        if (q.v_out) { q.v_out[2 * b] = a; q.v_out[2 * b + 1] = bb; }
    }
    const float s = (sq < 1.0f) ? 1.0f : sq;                                     // Imagen.py:320 clamp_(min=1.): a NaN threshold stays NaN
    const int k = (pp.T - 1) - t;
    const float* nz = pp.noise ? pp.noise + ((size_t)k * pp.B + b) * n : nullptr;
#pragma unroll
    for (int u = 0; u < SG_MAXQ; ++u) {
        const int qd = q0 + tid + u * SG_NT;
        if (qd >= nq) continue;
        float z[4];
        if (nz) { const float4 v = mi_ldg4(nz + 4 * qd); z[0] = v.x; z[1] = v.y; z[2] = v.z; z[3] = v.w; }
        else randn4(pp.seed_dev ? *pp.seed_dev : pp.seed, (unsigned)(pp.sample0 + b), (unsigned)(pp.stream_base + k), (unsigned)qd, z);
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x0 = x0v[u][e];
            x0 = __fdiv_rn(x0 != x0 ? x0 : fminf(fmaxf(x0, -s), s), s);                  // Imagen.py:323 (torch.clamp propagates NaN)
            const float mean = __fadd_rn(__fmul_rn(c1, x0), __fmul_rn(c2, xtv[u][e]));     // diffusion_model.py:118-121
            r[e] = __fadd_rn(mean, __fmul_rn(sigma, z[e]));                               // Imagen.py:370
        }
        mi_stg4(pp.x + ob + 4 * qd, make_float4(r[0], r[1], r[2], r[3]));
    }
}

}  // namespace

extern "C" int mi_cfg_x0_fwd(const mi_cfg_x0_params* p, void* stream) {
    if (p->B <= 0 || p->n <= 0) { mi_set_error("mi_cfg_x0_fwd: empty"); return MI_ERR_INVALID; }
    if (p->x0 && (!p->x_t || !p->coef || !p->t_state)) { mi_set_error("mi_cfg_x0_fwd: x0 needs x_t, coef, t_state"); return MI_ERR_INVALID; }
    if (p->hist0 && !p->x0) { mi_set_error("mi_cfg_x0_fwd: hist0 needs x0"); return MI_ERR_INVALID; }
    if (p->hist0) hipLaunchKernelGGL(HIP_KERNEL_NAME(cfg_x0_kernel<true>), dim3(grid_for(p->n, 128), p->B), dim3(256), 0, (hipStream_t)stream, *p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(cfg_x0_kernel<false>), dim3(grid_for(p->n, 256), p->B), dim3(256), 0, (hipStream_t)stream, *p);
    return mi_check_launch("cfg_x0_kernel");
}

extern "C" int mi_quantile_fwd(const mi_quantile_params* p, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (p->B <= 0 || p->n <= 0 || p->k_lo < 0 || p->k_hi >= p->n || p->k_lo > p->k_hi) { mi_set_error("mi_quantile_fwd: bad ranks"); return MI_ERR_INVALID; }
    if (p->pass0_done && !p->self_cleaning) { mi_set_error("mi_quantile_fwd: pass0_done needs self_cleaning (the memset would erase pass 0)"); return MI_ERR_INVALID; }
    if (!p->self_cleaning && hipMemsetAsync(p->hist, 0, (size_t)3 * p->B * 2 * MI_Q_BINS * sizeof(unsigned), st) != hipSuccess) { mi_set_error("mi_quantile_fwd: memset failed"); return MI_ERR_LAUNCH; }
    const dim3 grid(grid_for((p->n + 15) / 16, Q_WGS), p->B);        // few workgroups per image: every one flushes its histogram with atomics
    if (!p->pass0_done) hipLaunchKernelGGL(HIP_KERNEL_NAME(quantile_hist_kernel<0>), grid, dim3(256), 0, st, *p);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(quantile_hist_kernel<1>), grid, dim3(256), 0, st, *p);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(quantile_hist_kernel<2>), grid, dim3(256), 0, st, *p);
    hipLaunchKernelGGL(quantile_finish_kernel, dim3(p->B), dim3(256), 0, st, *p);
    return mi_check_launch("quantile kernels");
}

extern "C" int mi_sampler_step_small_fwd(const mi_cfg_x0_params* c, const mi_quantile_params* q, const mi_posterior_params* pp, void* stream) {
    if (c->B <= 0 || c->n <= 0 || q->B != c->B || pp->B != c->B || q->n != c->n || pp->n != c->n) { mi_set_error("mi_sampler_step_small_fwd: inconsistent B / n"); return MI_ERR_INVALID; }
    if (c->n > MI_SAMPLER_SMALL_N) { mi_set_error("mi_sampler_step_small_fwd: n = %d > %d", c->n, MI_SAMPLER_SMALL_N); return MI_ERR_UNSUPPORTED; }
    if (!c->x_t || !c->coef || !c->t_state || !pp->x || c->t_state != pp->t_state || c->t_off != pp->t_off || c->coef != pp->coef || c->x_t != pp->x) {
        mi_set_error("mi_sampler_step_small_fwd: needs x_t == x, one coef table and one t_state / t_off for the step"); return MI_ERR_INVALID;
    }
    if (q->k_lo < 0 || q->k_hi >= q->n || q->k_lo > q->k_hi) { mi_set_error("mi_sampler_step_small_fwd: bad ranks"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(sampler_small_kernel, dim3(c->B), dim3(SS_NT), 0, (hipStream_t)stream, *c, *q, *pp);
    return mi_check_launch("sampler_small_kernel");
}

extern "C" int mi_sampler_group_size(int n) {
    if (n <= 0 || (n & 3)) return 0;
    const int g = ((n >> 2) + SG_NT * SG_MAXQ - 1) / (SG_NT * SG_MAXQ);
    return g <= 256 ? g : 0;
}
extern "C" long long mi_sampler_group_sync_bytes(int B, int n) {
    return (B > 0 && mi_sampler_group_size(n) > 0) ? sg_sync_layout(B).total : 0;
}
extern "C" int mi_sampler_step_group_fwd(const mi_cfg_x0_params* c, const mi_quantile_params* q, const mi_posterior_params* pp, void* sync, void* stream) {
    if (c->B <= 0 || c->n <= 0 || q->B != c->B || pp->B != c->B || q->n != c->n || pp->n != c->n) { mi_set_error("mi_sampler_step_group_fwd: inconsistent B / n"); return MI_ERR_INVALID; }
    const int G = mi_sampler_group_size(c->n);
    if (!G) { mi_set_error("mi_sampler_step_group_fwd: n = %d unsupported (a multiple of 4, at most %d)", c->n, 256 * SG_NT * SG_MAXQ * 4); return MI_ERR_UNSUPPORTED; }
    if (!sync) { mi_set_error("mi_sampler_step_group_fwd: sync buffer missing"); return MI_ERR_INVALID; }
    if (!c->x_t || !c->coef || !c->t_state || !pp->x || c->t_state != pp->t_state || c->t_off != pp->t_off || c->coef != pp->coef || c->x_t != pp->x) {
        mi_set_error("mi_sampler_step_group_fwd: needs x_t == x, one coef table and one t_state / t_off for the step"); return MI_ERR_INVALID;
    }
    if (q->k_lo < 0 || q->k_hi >= q->n || q->k_lo > q->k_hi) { mi_set_error("mi_sampler_step_group_fwd: bad ranks"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(sampler_group_kernel, dim3(c->B * G), dim3(SG_NT), 0, (hipStream_t)stream, *c, *q, *pp, (char*)sync, G);
    return mi_check_launch("sampler_group_kernel");
}

extern "C" int mi_posterior_fwd(const mi_posterior_params* p, void* stream) {
    if (p->B <= 0 || p->n <= 0) { mi_set_error("mi_posterior_fwd: empty"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(posterior_kernel, dim3(grid_for((p->n + 3) / 4, 256), p->B), dim3(256), 0, (hipStream_t)stream, *p);
    return mi_check_launch("posterior_kernel");
}

extern "C" int mi_step_advance(int* t_state, int64_t* times, int B, void* stream) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t_state, (long long*)times, B, 0, 0);
    return mi_check_launch("step_advance_kernel");
}
extern "C" int mi_step_advance_by(int* t_state, int64_t* times, int B, int n, void* stream) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t_state, (long long*)times, B, 2, n);
    return mi_check_launch("step_advance_kernel");
}
extern "C" int mi_step_set(int* t_state, int64_t* times, int B, int value, void* stream) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t_state, (long long*)times, B, 1, value);
    return mi_check_launch("step_advance_kernel");
}

extern "C" int mi_randn_fill(float* out, int B, int n, uint64_t seed, int sample0, int stream_id, void* stream) {
    hipLaunchKernelGGL(randn_fill_kernel, dim3(grid_for((n + 3) / 4, 256), B), dim3(256), 0, (hipStream_t)stream, out, n, (unsigned long long)seed, sample0, stream_id);
    return mi_check_launch("randn_fill_kernel");
}

extern "C" int mi_finalize_images(const float* x, float* out, int64_t total, int unnormalize, void* stream) {
    hipLaunchKernelGGL(finalize_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, out, (long long)total, unnormalize);
    return mi_check_launch("finalize_kernel");
}

extern "C" int mi_lowres_augment(const float* img, const float* noise, float* out, int64_t total, float a, float b, int normalize, void* stream) {
    hipLaunchKernelGGL(lowres_augment_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img, noise, out, (long long)total, a, b, normalize);
    return mi_check_launch("lowres_augment_kernel");
}

extern "C" int mi_resize_fwd(const mi_resize_params* p, void* stream) {
    if (p->planes <= 0 || p->KH <= 0 || p->KW <= 0) { mi_set_error("mi_resize_fwd: bad params"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(resize_kernel, dim3(grid_for((long long)p->Hout * p->Wout, 1024), p->planes), dim3(256), 0, (hipStream_t)stream, *p);
    return mi_check_launch("resize_kernel");
}

// ------------------------------------------------------------------ HIP graphs
extern "C" int mi_graph_begin(void* stream) {
    if (hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed) != hipSuccess) { mi_set_error("hipStreamBeginCapture failed"); return MI_ERR_LAUNCH; }
    return MI_OK;
}
extern "C" int mi_graph_end(void* stream, void** graph_exec) {
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture((hipStream_t)stream, &g) != hipSuccess || g == nullptr) { mi_set_error("hipStreamEndCapture failed"); return MI_ERR_LAUNCH; }
    hipGraphExec_t ge = nullptr;
    const hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { mi_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e)); return MI_ERR_LAUNCH; }
    *graph_exec = (void*)ge;
    return MI_OK;
}
extern "C" int mi_graph_launch(void* graph_exec, void* stream) {
    const hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    if (e != hipSuccess) { mi_set_error("hipGraphLaunch failed: %s", hipGetErrorString(e)); return MI_ERR_LAUNCH; }
    return MI_OK;
}
extern "C" int mi_graph_destroy(void* graph_exec) {
    if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    return MI_OK;
}
