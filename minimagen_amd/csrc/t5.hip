// K16: T5 encoder building blocks (minimagen/t5.py:71-84 calls transformers' T5Stack): embedding gather,
// T5LayerNorm (RMS), fp32 MFMA GEMM with fused ReLU / gated-GELU / residual epilogues, and the self-attention core
// (unscaled q.k^T + bucketed relative position bias + key mask, softmax, .v) on v_mfma_f32_16x16x4_f32.
#include "common.hip.h"
#include <cstdlib>

namespace {

__device__ __forceinline__ float gelu_new(float x) {       // transformers' NewGELUActivation (tanh form)
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.0f + tanhf(u));
}

// ---- C = act(A.W^T) [* (A.G^T)] + R ; 64x64 block tile, 4 waves of 32x32 (2x2 MFMA tiles), K step 16 through LDS
template <bool GATED>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ G,
                                                       const float* __restrict__ R, float* __restrict__ Cout, int M, int N, int K, int act) {
    constexpr int BM = 64, BN = 64, BK = 16, LD = BK + 1;
    __shared__ float As[BM][LD], Ws[BN][LD], Gs[GATED ? BN : 1][LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int lr = tid >> 2, lc = (tid & 3) * 4;            // staging: one float4 per work-item and matrix
    f32x4 acc[2][2], accg[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accg[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    for (int k0 = 0; k0 < K; k0 += BK) {
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m0 + lr < M) a4 = *reinterpret_cast<const float4*>(A + (size_t)(m0 + lr) * K + k0 + lc);
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 + lr < N) w4 = *reinterpret_cast<const float4*>(W + (size_t)(n0 + lr) * K + k0 + lc);
        float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (GATED) g4 = *reinterpret_cast<const float4*>(G + (size_t)(n0 + lr) * K + k0 + lc);
        __syncthreads();
        As[lr][lc] = a4.x; As[lr][lc + 1] = a4.y; As[lr][lc + 2] = a4.z; As[lr][lc + 3] = a4.w;
        Ws[lr][lc] = w4.x; Ws[lr][lc + 1] = w4.y; Ws[lr][lc + 2] = w4.z; Ws[lr][lc + 3] = w4.w;
        if (GATED) { Gs[lr][lc] = g4.x; Gs[lr][lc + 1] = g4.y; Gs[lr][lc + 2] = g4.z; Gs[lr][lc + 3] = g4.w; }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            float a[2], b[2], g[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[wm * 32 + i * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                b[j] = Ws[wn * 32 + j * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
                g[j] = GATED ? Gs[wn * 32 + j * 16 + (lane & 15)][kk * 4 + (lane >> 4)] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
                    if (GATED) accg[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], g[j], accg[i][j], 0, 0, 0);
                }
        }
    }
    // D layout: column (n) = lane & 15, row (m) = 4 * (lane >> 4) + r
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + i * 16 + 4 * (lane >> 4) + r, n = n0 + wn * 32 + j * 16 + (lane & 15);
                if (m < M && n < N) {
                    float v = acc[i][j][r];
                    if (act == 1) v = fmaxf(v, 0.0f);
                    else if (act == 2) v = gelu_new(v);
                    else if (act == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                    if (GATED) v *= accg[i][j][r];
                    if (R) v += R[(size_t)m * N + n];
                    Cout[(size_t)m * N + n] = v;
                }
            }
}

// ---- the same GEMM on the f16 matrix-core instruction (16x the fp32 MFMA rate) at fp32 accuracy: every operand is split x = hi + lo (two
// fp16) and multiplied as lo*hi + hi*lo + hi*hi with fp32 accumulation (2^-22 per product, as in the conv / attention kernels).  Range
// safety without any knowledge about the tensors: BLOCK scaling per K slice -- the 64 x 32 tile of A, of W (and of G) is scaled by the power
// of two that brings its largest magnitude into [128, 256) before the split (exact), the slice's products are accumulated in a zeroed
// accumulator and added to the running fp32 sum with the inverse scale (one FMA per accumulator register and slice).  An element 2^-22
// below its tile's maximum still has a normal fp16 hi half; what a smaller element loses is below fp32 rounding of the tile's products.
typedef _Float16 t5_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int t5_scale_exp(float m) {           // k with m * 2^k in [128, 256); 0 for zero / non-finite input
    const int be = (int)((__float_as_uint(m) & 0x7fffffffu) >> 23);
    return (be == 0 || be == 255) ? 0 : 134 - be;
}
__device__ __forceinline__ void t5_split8(const float4& u, const float4& v, float sc, uint4& hi, uint4& lo) {
    const float x[8] = {u.x * sc, u.y * sc, u.z * sc, u.w * sc, v.x * sc, v.y * sc, v.z * sc, v.w * sc};
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const mi_f16x2 h2 = {(_Float16)x[2 * e], (_Float16)x[2 * e + 1]};
        h[e] = __builtin_bit_cast(unsigned, h2);
        l[e] = mi_split_lo2(h[e], x[2 * e], x[2 * e + 1]);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}
// rowsq_in [M][nparts] (may be NULL): partial sums of squares of A's rows -> row m of the product is scaled by rsqrt(sum / K + eps) before the
// activation: T5's RMSNorm in front of a projection, y = (x rsqrt(mean x^2 + eps) w) W^T = rsqrt(..) (x (W diag w)^T), with w folded into W by the host.
// rowsq_out [M][gridDim.x] (may be NULL): this tile's sum of squares of every output row (after the residual) -- what the NEXT projection's RMSNorm needs.
// KS = K steps of 32 per slice (round 6: 2 where K % 64 == 0 -- the slice loop is paced by its barriers and cross-lane maxima, ~1 us per slice
// whatever it multiplies: half the slices for the same work)
template <bool GATED, int KS>
__global__ __launch_bounds__(256) void gemm_f16x3_kernel(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ G,
                                                         const float* __restrict__ R, float* __restrict__ Cout, int M, int N, int K, int act,
                                                         const float* __restrict__ rowsq_in, int nparts, float eps, float* __restrict__ rowsq_out) {
    constexpr int BM = 64, BN = 64, BK = 32 * KS, PITCH = 4 * KS + 1;      // rows of 4 KS 16-byte chunks (8 halves each) + 1 pad chunk: conflict-free ds_read_b128
    __shared__ __attribute__((aligned(16))) uint4 Ah[BM * PITCH], Al[BM * PITCH], Wh[BN * PITCH], Wl[BN * PITCH];
    __shared__ __attribute__((aligned(16))) uint4 Gh[GATED ? BN * PITCH : 1], Gl[GATED ? BN * PITCH : 1];
    __shared__ float smax[3][4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int lr = tid >> 2, lc = tid & 3;                    // staging: row, 8-float chunk (+ 4 per K step)
    f32x4 acc[2][2], accg[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accg[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto amax8 = [](const float4& u, const float4& v) {
        return fmaxf(fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fmaxf(fabsf(u.z), fabsf(u.w))), fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    };
    // the next slice's global loads are issued right after this slice's values have been split into LDS: they fly under the MFMA section
    // (and under the other resident workgroups) instead of in front of every slice
    float4 a0[KS], a1[KS], w0[KS], w1[KS], g0[KS], g1[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { a0[ks] = z4; a1[ks] = z4; w0[ks] = z4; w1[ks] = z4; g0[ks] = z4; g1[ks] = z4; }
    auto load_slice = [&](int k0) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kk = k0 + 32 * ks + 8 * lc;
            if (m0 + lr < M) { const float4* pa = reinterpret_cast<const float4*>(A + (size_t)(m0 + lr) * K + kk); a0[ks] = pa[0]; a1[ks] = pa[1]; }
            if (n0 + lr < N) {
                const float4* pw = reinterpret_cast<const float4*>(W + (size_t)(n0 + lr) * K + kk); w0[ks] = pw[0]; w1[ks] = pw[1];
                if (GATED) { const float4* pg = reinterpret_cast<const float4*>(G + (size_t)(n0 + lr) * K + kk); g0[ks] = pg[0]; g1[ks] = pg[1]; }
            }
        }
    };
    load_slice(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        float ma = 0.f, mw = 0.f, mg = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { ma = fmaxf(ma, amax8(a0[ks], a1[ks])); mw = fmaxf(mw, amax8(w0[ks], w1[ks])); if (GATED) mg = fmaxf(mg, amax8(g0[ks], g1[ks])); }
        ma = mi_wave_max(ma); mw = mi_wave_max(mw); mg = GATED ? mi_wave_max(mg) : 0.0f;
        __syncthreads();                                      // the previous slice's fragments and maxima are consumed
        if (lane == 0) { smax[0][wave] = ma; smax[1][wave] = mw; smax[2][wave] = mg; }
        __syncthreads();
        const int ea = t5_scale_exp(fmaxf(fmaxf(smax[0][0], smax[0][1]), fmaxf(smax[0][2], smax[0][3])));
        const int ew = t5_scale_exp(fmaxf(fmaxf(smax[1][0], smax[1][1]), fmaxf(smax[1][2], smax[1][3])));
        const int eg = GATED ? t5_scale_exp(fmaxf(fmaxf(smax[2][0], smax[2][1]), fmaxf(smax[2][2], smax[2][3]))) : 0;
        uint4 hi, lo;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int o = lr * PITCH + lc + 4 * ks;
            t5_split8(a0[ks], a1[ks], ldexpf(1.0f, ea), hi, lo); Ah[o] = hi; Al[o] = lo;
            t5_split8(w0[ks], w1[ks], ldexpf(1.0f, ew), hi, lo); Wh[o] = hi; Wl[o] = lo;
            if (GATED) { t5_split8(g0[ks], g1[ks], ldexpf(1.0f, eg), hi, lo); Gh[o] = hi; Gl[o] = lo; }
        }
        if (k0 + BK < K) load_slice(k0 + BK);
        __syncthreads();
        const float un = ldexpf(1.0f, -(ea + ew)), ung = ldexpf(1.0f, -(ea + eg));
        f32x4 sl[2][2], slg[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) { sl[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; slg[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            t5_f16x8 ah[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int o = (wm * 32 + i * 16 + (lane & 15)) * PITCH + (lane >> 4) + 4 * ks;
                ah[i] = __builtin_bit_cast(t5_f16x8, Ah[o]);
                al[i] = __builtin_bit_cast(t5_f16x8, Al[o]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int o = (wn * 32 + j * 16 + (lane & 15)) * PITCH + (lane >> 4) + 4 * ks;
                const t5_f16x8 bh = __builtin_bit_cast(t5_f16x8, Wh[o]), bl = __builtin_bit_cast(t5_f16x8, Wl[o]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    sl[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, sl[i][j], 0, 0, 0);
                    sl[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, sl[i][j], 0, 0, 0);
                    sl[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, sl[i][j], 0, 0, 0);
                }
                if (GATED) {
                    const t5_f16x8 gh = __builtin_bit_cast(t5_f16x8, Gh[o]), gl = __builtin_bit_cast(t5_f16x8, Gl[o]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        slg[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], gh, slg[i][j], 0, 0, 0);
                        slg[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], gl, slg[i][j], 0, 0, 0);
                        slg[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], gh, slg[i][j], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[i][j][r] = fmaf(sl[i][j][r], un, acc[i][j][r]);
                    if (GATED) accg[i][j][r] = fmaf(slg[i][j][r], ung, accg[i][j][r]);
                }
    }
    // D layout as in the fp32 kernel above: row (m) = 4 * (lane >> 4) + r, column (n) = lane & 15
    __shared__ float sq[2][BM], rsc[BM];
    if (rowsq_in) {                                        // one work-item per row of the tile adds up the row's partials: rsqrt(mean x^2 + eps)
        if (tid < BM) {
            float t = 0.0f;
            if (m0 + tid < M)
                for (int q = 0; q < nparts; ++q) t += rowsq_in[(size_t)(m0 + tid) * nparts + q];
            rsc[tid] = 1.0f / sqrtf(t / (float)K + eps);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ml = wm * 32 + i * 16 + 4 * (lane >> 4) + r, m = m0 + ml;
            const float rs = rowsq_in ? rsc[ml] : 1.0f;
            float part = 0.0f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 32 + j * 16 + (lane & 15);
                if (m < M && n < N) {
                    float v = acc[i][j][r] * rs;
                    if (act == 1) v = fmaxf(v, 0.0f);
                    else if (act == 2) v = gelu_new(v);
                    else if (act == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                    if (GATED) v *= accg[i][j][r] * rs;
                    if (R) v += R[(size_t)m * N + n];
                    Cout[(size_t)m * N + n] = v;
                    part = fmaf(v, v, part);
                }
            }
            if (rowsq_out) {                                   // (wave-uniform) the 16 lanes of a row group hold its 32 columns of this wave
                part += __shfl_xor(part, 1); part += __shfl_xor(part, 2); part += __shfl_xor(part, 4); part += __shfl_xor(part, 8);
                if ((lane & 15) == 0) sq[wn][ml] = part;
            }
        }
    if (rowsq_out) {
        __syncthreads();
        if (tid < BM && m0 + tid < M) rowsq_out[(size_t)(m0 + tid) * gridDim.x + blockIdx.x] = sq[0][tid] + sq[1][tid];
    }
}

__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* x, const float* w, float* y, int rows, int dim, float eps, const uint8_t* zero_mask) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* xr = x + (size_t)row * dim;
    float s = 0.0f;
    for (int i = tid; i < dim; i += 256) s = fmaf(xr[i], xr[i], s);
    s = mi_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float rs = 1.0f / sqrtf(tot / (float)dim + eps);
    const bool zero = zero_mask && zero_mask[row] == 0;
    for (int i = tid; i < dim; i += 256) y[(size_t)row * dim + i] = zero ? 0.0f : xr[i] * rs * w[i];
}

__global__ __launch_bounds__(256) void embed_rows_kernel(const long long* ids, const float* table, float* out, int rows, int dim, float* rowsq, int nparts) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const float* src = table + (size_t)ids[row] * dim;
    float s = 0.0f;
    for (int i = threadIdx.x; i < dim; i += 256) { const float v = src[i]; out[(size_t)row * dim + i] = v; s = fmaf(v, v, s); }
    if (rowsq) {                    // the row's sum of squares in part 0 of its rowsq record (the layout mi_gemm_rms_f32 reads), the other parts zero
        s = mi_wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x < nparts) rowsq[(size_t)row * nparts + threadIdx.x] = threadIdx.x == 0 ? (red[0] + red[1]) + (red[2] + red[3]) : 0.0f;
    }
}

// one wave = 16 queries of one (batch, head); the whole (<=256)-key score row lives in registers like K9
template <int JT>
__global__ __launch_bounds__(256) void t5_attention_kernel(const float* __restrict__ qkv, const float* __restrict__ bias_tab,
                                                           const uint8_t* __restrict__ key_mask, float* __restrict__ ctx, int L, int heads) {
    constexpr int D = 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int inner = heads * D, ld = 3 * inner;
    const int i0 = (blockIdx.x * 4 + wave) * 16;
    if (i0 >= L) return;
    const int iq = i0 + lq;
    const bool qok = iq < L;
    const int iqc = qok ? iq : L - 1;                       // rows past the sequence end compute (and discard) row L-1's bias
    const float* base = qkv + (size_t)b * L * ld;
    float qf[16];                                            // B operand: Q[i][d = 4kk + lg]
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) qf[kk] = qok ? base[(size_t)iq * ld + h * D + 4 * kk + lg] : 0.0f;
    f32x4 s[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int jr = 16 * jt + lq;                         // A operand: K[j = 16jt + lq][d = 4kk + lg]
        const float* kp = base + (size_t)(jr < L ? jr : 0) * ld + inner + h * D + lg;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[4 * kk], qf[kk], acc, 0, 0, 0);
        s[jt] = acc;
    }
    float m = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + 4 * lg + r;
            float v = -INFINITY;
            if (j < L) {
                v = s[jt][r] + bias_tab[(size_t)h * (2 * L - 1) + (j - iqc) + (L - 1)];
                if (key_mask && key_mask[(size_t)b * L + j] == 0) v += -3.4028234663852886e38f;     // (1 - mask) * finfo.min
            }
            s[jt][r] = v;
            m = fmaxf(m, v);
        }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.0f;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f((s[jt][r] - m) * 1.44269504088896340736f);
            s[jt][r] = e;
            l += e;
        }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    f32x4 o[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) o[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + 4 * lg + r;              // A operand of PV: V[j][d = 16mt + lq]
            const float* vp = base + (size_t)(j < L ? j : 0) * ld + 2 * inner + h * D + lq;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) o[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[16 * mt], s[jt][r], o[mt], 0, 0, 0);
        }
    const float inv = 1.0f / l;
    if (qok) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ctx[((size_t)b * L + iq) * inner + h * D + 16 * mt + 4 * lg + r] = o[mt][r] * inv;
    }
}

}  // namespace

static int gemm_f16x3_launch(const float* A, const float* W, const float* gate, const float* R, float* Cout, int M, int N, int K, int act,
                             const float* rowsq_in, int nparts, float eps, float* rowsq_out, hipStream_t st) {
    const dim3 grid((N + 63) / 64, (M + 63) / 64);
#define MI_GEMM_GO(GATED, KS) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_f16x3_kernel<GATED, KS>), grid, dim3(256), 0, st, A, W, gate, R, Cout, M, N, K, act, rowsq_in, nparts, eps, rowsq_out)
    static const bool ks1 = getenv("MI_GEMM_KS1") != nullptr;       // A/B knob: 32-wide slices everywhere
    // (128-wide slices, measured: wo 54 -> 47 us but qkv 31 -> 35 and wi 37 -> 41: two workgroups per CU instead of three; not kept)
    if ((K % 64) == 0 && !ks1) { if (gate) MI_GEMM_GO(true, 2); else MI_GEMM_GO(false, 2); }
    else { if (gate) MI_GEMM_GO(true, 1); else MI_GEMM_GO(false, 1); }
#undef MI_GEMM_GO
    return mi_check_launch("gemm_f16x3_kernel");
}

extern "C" int mi_gemm_f32(const float* A, const float* W, const float* gate, const float* R, float* Cout, int M, int N, int K, int act, void* stream) {
    if (M <= 0 || N <= 0 || (gate && (N % 64) != 0) || (K % 16) != 0) { mi_set_error("mi_gemm_f32: need M>0, K%%16==0 (and N%%64==0 when gated) (got %d,%d,%d)", M, N, K); return MI_ERR_INVALID; }
    const dim3 grid((N + 63) / 64, (M + 63) / 64);
    // K in multiples of 32 (every T5 / attention projection): 3-term fp16 products on the f16 matrix-core instruction, block-scaled per K slice;
    // otherwise (and with MI_GEMM_EXACT_F32 set in the environment, for A/B measurements) the exact-fp32 MFMA kernel
    static const bool exact = getenv("MI_GEMM_EXACT_F32") != nullptr;
    if ((K % 32) == 0 && !exact) {
        return gemm_f16x3_launch(A, W, gate, R, Cout, M, N, K, act, nullptr, 0, 0.0f, nullptr, (hipStream_t)stream);
    }
    if (gate) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_f32_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, A, W, gate, R, Cout, M, N, K, act);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_f32_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, A, W, gate, R, Cout, M, N, K, act);
    return mi_check_launch("gemm_f32_kernel");
}

extern "C" int mi_gemm_rms_f32(const float* A, const float* W, const float* gate, const float* R, float* Cout, int M, int N, int K, int act,
                               const float* rowsq_in, int nparts, float eps, float* rowsq_out, void* stream) {
    if (M <= 0 || N <= 0 || (gate && (N % 64) != 0) || (K % 32) != 0 || (rowsq_in && nparts <= 0)) { mi_set_error("mi_gemm_rms_f32: need M>0, K%%32==0, nparts>0 with rowsq_in (and N%%64==0 when gated) (got %d,%d,%d)", M, N, K); return MI_ERR_INVALID; }
    return gemm_f16x3_launch(A, W, gate, R, Cout, M, N, K, act, rowsq_in, nparts, eps, rowsq_out, (hipStream_t)stream);
}

extern "C" int mi_rmsnorm(const float* x, const float* w, float* y, int rows, int dim, float eps, const uint8_t* zero_mask, void* stream) {
    if (rows <= 0 || dim <= 0) { mi_set_error("mi_rmsnorm: empty"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, w, y, rows, dim, eps, zero_mask);
    return mi_check_launch("rmsnorm_kernel");
}

extern "C" int mi_embed_rows(const int64_t* ids, const float* table, float* out, int rows, int dim, void* stream) {
    if (rows <= 0) { mi_set_error("mi_embed_rows: empty"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(embed_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const long long*)ids, table, out, rows, dim, (float*)nullptr, 0);
    return mi_check_launch("embed_rows_kernel");
}

extern "C" int mi_embed_rows_sq(const int64_t* ids, const float* table, float* out, float* rowsq, int nparts, int rows, int dim, void* stream) {
    if (rows <= 0 || nparts <= 0 || nparts > 256 || !rowsq) { mi_set_error("mi_embed_rows_sq: empty / bad rowsq record"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(embed_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const long long*)ids, table, out, rows, dim, rowsq, nparts);
    return mi_check_launch("embed_rows_kernel");
}

extern "C" int mi_t5_attention(const float* qkv, const float* bias_tab, const uint8_t* key_mask, float* ctx, int B, int L, int heads, void* stream) {
    if (B <= 0 || L <= 0 || L > 256) { mi_set_error("mi_t5_attention: sequence length %d outside (0, 256] (t5.py MAX_LENGTH)", L); return MI_ERR_INVALID; }
    const dim3 grid((L + 63) / 64, heads, B);
    hipStream_t st = (hipStream_t)stream;
    if (L <= 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(t5_attention_kernel<4>), grid, dim3(256), 0, st, qkv, bias_tab, key_mask, ctx, L, heads);
    else if (L <= 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(t5_attention_kernel<8>), grid, dim3(256), 0, st, qkv, bias_tab, key_mask, ctx, L, heads);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(t5_attention_kernel<16>), grid, dim3(256), 0, st, qkv, bias_tab, key_mask, ctx, L, heads);
    return mi_check_launch("t5_attention_kernel");
}
