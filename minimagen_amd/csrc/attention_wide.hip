// Wide-channel attention (Unet() default, Base, Super: C > 32): the unfolded forms of CrossAttention (layers.py:220-251), the
// multi-query Attention (layers.py:52-104) and ChanFeedForward (layers.py:148-161) as token-major building blocks:
//   LayerNorm tokens (attention.hip) -> projections (mi_gemm_f32, t5.hip) -> mi_flash_attn_fwd -> projection -> mi_tokens_to_nchw_fwd.
// The folded kernels of attention.hip pay K = C per score, which only wins while C <= 32; from 64 channels up the reference's own
// factorisation (dim_head 64) is the cheaper one.  Everything here is exact fp32 on v_mfma_f32_16x16x4_f32 (no operand splits).
#include "common.hip.h"
#include <type_traits>
#include <cstdlib>

namespace {

// ---- flash attention, dim_head 64: one wave = 16 queries of one (batch, head); context walked in 64-row chunks staged through LDS
// (shared by the workgroup's four waves) with an online softmax.  Scores S^T = K . Q^T (swapped operands): a query's scores sit in one
// lane (+ the three lanes 16 apart), and the C/D layout of S IS the B layout of O^T += V^T . P^T for the k-order j = 4k + r, so P never
// moves (same register trick as attention.hip).
__global__ __launch_bounds__(256) void flash_attn_kernel(const mi_flash_attn_params p) {
    constexpr int D = 64, KS = 66, VS = 68;            // LDS row strides: conflict-free A-operand reads of K (lanes = rows) and V (lanes = dims)
    __shared__ float Ks[64 * KS], Vs[64 * VS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z, kvh = p.kv_heads == 1 ? 0 : h;
    const int inner = p.heads * D;
    const int tok = (blockIdx.x * 4 + wave) * 16 + lq;
    const int tokc = tok < p.HW ? tok : p.HW - 1;
    const int nnull = p.null_k ? 1 : 0, J = nnull + p.n0 + p.n1;
    // Q as the B operand: this lane supplies dims 4kk + lg of query lq, pre-scaled (softmax scale * log2 e: scores in log2 units)
    float q[16];
    {
        const float* qr = p.q + ((size_t)b * p.HW + tokc) * inner + h * D;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) q[kk] = qr[4 * kk + lg] * p.q_scale;
    }
    float m = -INFINITY, l = 0.0f;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int j0 = 0; j0 < J; j0 += 64) {
        __syncthreads();                          // the previous chunk is no longer read
        {   // stage 64 context rows x 64 dims of K and V: work-item -> (row, 16-dim quarter)
            const int row = tid >> 2, d0 = (tid & 3) * 16, jj = j0 + row;
            const float* ksrc = nullptr;
            const float* vsrc = nullptr;
            if (jj < J) {
                if (jj < nnull) { ksrc = p.null_k; vsrc = p.null_v; }
                else if (jj - nnull < p.n0) { const size_t o_ = (size_t)b * p.bs0 + (size_t)(jj - nnull) * p.ld0 + kvh * D; ksrc = p.k0 + o_; vsrc = p.v0 + o_; }
                else { const size_t o_ = (size_t)b * p.bs1 + (size_t)(jj - nnull - p.n0) * p.ld1 + kvh * D; ksrc = p.k1 + o_; vsrc = p.v1 + o_; }
            }
#pragma unroll
            for (int e = 0; e < 16; e += 4) {
                float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = k4;
                if (ksrc) { k4 = *reinterpret_cast<const float4*>(ksrc + d0 + e); v4 = *reinterpret_cast<const float4*>(vsrc + d0 + e); }
                float* kd = &Ks[row * KS + d0 + e];
                kd[0] = k4.x; kd[1] = k4.y; kd[2] = k4.z; kd[3] = k4.w;
                *reinterpret_cast<float4*>(&Vs[row * VS + d0 + e]) = v4;
            }
        }
        __syncthreads();
        f32x4 s[4];
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[(16 * jt + lq) * KS + 4 * kk + lg], q[kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (j0 + 16 * jt + 4 * lg + r >= J) acc[r] = -INFINITY;
                mx = fmaxf(mx, acc[r]);
            }
            s[jt] = acc;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(m, mx);                      // finite: every chunk holds at least one live row
        const float alpha = __builtin_amdgcn_exp2f(m - mn);  // m = -inf on the first chunk -> 0
        m = mn;
        l *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pe = __builtin_amdgcn_exp2f(s[jt][r] - mn);
                l += pe;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[(16 * jt + 4 * lg + r) * VS + 16 * dt + lq], pe, o[dt], 0, 0, 0);
            }
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float linv = 1.0f / l;
    if (tok < p.HW) {
        float* orow = p.out + ((size_t)b * p.HW + tok) * inner + h * D;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<float4*>(orow + 16 * dt + 4 * lg) = make_float4(o[dt][0] * linv, o[dt][1] * linv, o[dt][2] * linv, o[dt][3] * linv);
    }
}

// ---- the same attention on the f16 matrix-core instruction (v_mfma_f32_16x16x32_f16, 16x the fp32 MFMA rate) at fp32 accuracy: Q, K, V and P are
// split x = hi + lo (two fp16) and multiplied as lo*hi + hi*lo + hi*hi with fp32 accumulation (2^-22 per product).  Range safety as in
// gemm_f16x3_kernel (t5.hip): per 64-row chunk K and V are scaled by the power of two that brings their largest magnitude into [128, 256)
// before the split (exact), Q once per wave; the scores are unscaled in fp32 before the softmax, a chunk's P.V goes through a zeroed
// accumulator and is added to the running output with the inverse scale.  K = 32 per instruction: QK^T contracts 32 of the 64 head dims,
// P.V a PAIR of score tiles -- the lane that holds S[j = 16t + 4lg + r][query lq] (t = 2hf, 2hf + 1) in the C/D layout supplies exactly
// those eight P values as its B operand, and V^T is staged in LDS with the context rows of a 32-row half permuted to that order
// (position 8lg + 4t' + r), so that a lane's A operand is one aligned 16-byte read.
typedef _Float16 fw_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int fw_scale_exp(float m) {           // k with m * 2^k in [128, 256); 0 for zero / non-finite input
    const int be = (int)((__float_as_uint(m) & 0x7fffffffu) >> 23);
    return (be == 0 || be == 255) ? 0 : 134 - be;
}
__device__ __forceinline__ void fw_split8(const float (&x)[8], uint4& hi, uint4& lo) {
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
// NW waves per workgroup: waves 0-3 are the four 16-query groups of one head, every further group of four waves another head of the SAME 64
// queries -- for the multi-query Attention (one shared k / v head) a staged, split and transposed K / V chunk then serves NW / 4 heads.
template <int NW>
__global__ __launch_bounds__(64 * NW) void flash_attn_f16x3_kernel(const mi_flash_attn_params p) {
    constexpr int D = 64, CP = 9;                        // LDS rows of 8 16-byte chunks (64 halves) + 1 pad chunk: conflict-free ds_read_b128
    constexpr int NH = NW / 4, EPT = 64 / NW, PPR = NW;  // heads per workgroup; head dims per work-item and pieces per row when staging
    __shared__ __attribute__((aligned(16))) uint4 KsH[64 * CP], KsL[64 * CP];      // [context row j][head dim d]
    __shared__ __attribute__((aligned(16))) uint4 VtH[64 * CP], VtL[64 * CP];      // [head dim d][permuted context row]
    __shared__ float smax[2][NW];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int h = blockIdx.y * NH + (wave >> 2), b = blockIdx.z, kvh = p.kv_heads == 1 ? 0 : h;
    const int inner = p.heads * D;
    const int tok = (blockIdx.x * 4 + (wave & 3)) * 16 + lq;
    const int tokc = tok < p.HW ? tok : p.HW - 1;
    const int nnull = p.null_k ? 1 : 0, J = nnull + p.n0 + p.n1;
    // Q as the B operand: this lane supplies dims 32hf + 8lg + e of query lq, pre-scaled by q_scale (scores in log2 units) and by 2^eq
    fw_f16x8 qh[2], ql[2];
    int eq;
    {
        const float* qr = p.q + ((size_t)b * p.HW + tokc) * inner + h * D;
        float qv[2][8];
        float mq = 0.0f;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const float4 a = *reinterpret_cast<const float4*>(qr + 32 * hf + 8 * lg), c = *reinterpret_cast<const float4*>(qr + 32 * hf + 8 * lg + 4);
            const float t[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) { qv[hf][e] = t[e] * p.q_scale; mq = fmaxf(mq, fabsf(qv[hf][e])); }
        }
        eq = fw_scale_exp(mi_wave_max(mq));
        const float sq = ldexpf(1.0f, eq);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = qv[hf][e] * sq;
            uint4 hi, lo;
            fw_split8(t, hi, lo);
            qh[hf] = __builtin_bit_cast(fw_f16x8, hi);
            ql[hf] = __builtin_bit_cast(fw_f16x8, lo);
        }
    }
    float m = -INFINITY, l = 0.0f;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // staging role: work-item -> (context row, EPT head dims); the row's position in the permuted order of V^T
    const int srow = tid / PPR, sd0 = (tid % PPR) * EPT;
    const int spos = (srow & 32) | (((srow >> 2) & 3) << 3) | (((srow >> 4) & 1) << 2) | (srow & 3);

    // this work-item's piece of the K / V chunk starting at context row j0, straight from the fp32 tensors; the NEXT chunk's loads are issued
    // right after the current one has been staged, so that they fly under the chunk's matrix-core work instead of in front of it
    float kf[EPT], vf[EPT];
    auto load_chunk = [&](int j0) {
        const int jj = j0 + srow;
        const float* ksrc = nullptr;
        const float* vsrc = nullptr;
        if (jj < J) {
            if (jj < nnull) { ksrc = p.null_k; vsrc = p.null_v; }
            else if (jj - nnull < p.n0) { const size_t o_ = (size_t)b * p.bs0 + (size_t)(jj - nnull) * p.ld0 + kvh * D; ksrc = p.k0 + o_; vsrc = p.v0 + o_; }
            else { const size_t o_ = (size_t)b * p.bs1 + (size_t)(jj - nnull - p.n0) * p.ld1 + kvh * D; ksrc = p.k1 + o_; vsrc = p.v1 + o_; }
        }
#pragma unroll
        for (int e = 0; e < EPT; e += 4) {
            float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = k4;
            if (ksrc) { k4 = *reinterpret_cast<const float4*>(ksrc + sd0 + e); v4 = *reinterpret_cast<const float4*>(vsrc + sd0 + e); }
            kf[e] = k4.x; kf[e + 1] = k4.y; kf[e + 2] = k4.z; kf[e + 3] = k4.w;
            vf[e] = v4.x; vf[e + 1] = v4.y; vf[e + 2] = v4.z; vf[e + 3] = v4.w;
        }
    };
    load_chunk(0);
    for (int j0 = 0; j0 < J; j0 += 64) {
        float mk = 0.0f, mv = 0.0f;
#pragma unroll
        for (int e = 0; e < EPT; ++e) { mk = fmaxf(mk, fabsf(kf[e])); mv = fmaxf(mv, fabsf(vf[e])); }
        mk = mi_wave_max(mk); mv = mi_wave_max(mv);
        __syncthreads();                          // the previous chunk's fragments and maxima are no longer read
        if (lane == 0) { smax[0][wave] = mk; smax[1][wave] = mv; }
        __syncthreads();
        float mka = 0.0f, mva = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { mka = fmaxf(mka, smax[0][w]); mva = fmaxf(mva, smax[1][w]); }
        const int ek = fw_scale_exp(mka), ev = fw_scale_exp(mva);
        {
            const float sk = ldexpf(1.0f, ek), sv = ldexpf(1.0f, ev);
            _Float16* ksh = reinterpret_cast<_Float16*>(KsH);
            _Float16* ksl = reinterpret_cast<_Float16*>(KsL);
#pragma unroll
            for (int e = 0; e < EPT; e += 4) {        // K row-major: 4 halves (8 bytes) of hi and of lo per step
                unsigned hb[2], lb[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float x0 = kf[e + 2 * q] * sk, x1 = kf[e + 2 * q + 1] * sk;
                    const mi_f16x2 h2 = {(_Float16)x0, (_Float16)x1};
                    hb[q] = __builtin_bit_cast(unsigned, h2);
                    lb[q] = mi_split_lo2(hb[q], x0, x1);
                }
                *reinterpret_cast<uint2*>(ksh + srow * (8 * CP) + sd0 + e) = make_uint2(hb[0], hb[1]);
                *reinterpret_cast<uint2*>(ksl + srow * (8 * CP) + sd0 + e) = make_uint2(lb[0], lb[1]);
            }
            _Float16* vth = reinterpret_cast<_Float16*>(VtH);
            _Float16* vtl = reinterpret_cast<_Float16*>(VtL);
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const float x = vf[e] * sv;
                const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
                vth[(sd0 + e) * (8 * CP) + spos] = hi;
                vtl[(sd0 + e) * (8 * CP) + spos] = lo;
            }
        }
        __syncthreads();
        if (j0 + 64 < J) load_chunk(j0 + 64);
        const float us = ldexpf(1.0f, -(ek + eq)), uv = ldexpf(1.0f, -ev);
        f32x4 s[4];
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const fw_f16x8 kh = __builtin_bit_cast(fw_f16x8, KsH[(16 * jt + lq) * CP + 4 * hf + lg]), kl = __builtin_bit_cast(fw_f16x8, KsL[(16 * jt + lq) * CP + 4 * hf + lg]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[hf], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[hf], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[hf], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[r] *= us;
                if (j0 + 16 * jt + 4 * lg + r >= J) acc[r] = -INFINITY;
                mx = fmaxf(mx, acc[r]);
            }
            s[jt] = acc;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(m, mx);                      // finite: every chunk holds at least one live row
        const float alpha = __builtin_amdgcn_exp2f(m - mn);  // m = -inf on the first chunk -> 0
        m = mn;
        l = mi_mul_rounded(l, alpha);                  // (the prepared-K/V kernel below skips this when alpha == 1: same bits either way)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
        fw_f16x8 ph[2], pl[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            float pe[8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { pe[4 * t + r] = __builtin_amdgcn_exp2f(s[2 * hf + t][r] - mn); l += pe[4 * t + r]; }
            uint4 hi, lo;
            fw_split8(pe, hi, lo);
            ph[hf] = __builtin_bit_cast(fw_f16x8, hi);
            pl[hf] = __builtin_bit_cast(fw_f16x8, lo);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4 sl = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const fw_f16x8 vh = __builtin_bit_cast(fw_f16x8, VtH[(16 * dt + lq) * CP + 4 * hf + lg]), vl = __builtin_bit_cast(fw_f16x8, VtL[(16 * dt + lq) * CP + 4 * hf + lg]);
                sl = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[hf], sl, 0, 0, 0);
                sl = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl[hf], sl, 0, 0, 0);
                sl = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph[hf], sl, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] = fmaf(sl[r], uv, o[dt][r]);
        }
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float linv = 1.0f / l;
    if (tok < p.HW) {
        float* orow = p.out + ((size_t)b * p.HW + tok) * inner + h * D;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<float4*>(orow + 16 * dt + 4 * lg) = make_float4(o[dt][0] * linv, o[dt][1] * linv, o[dt][2] * linv, o[dt][3] * linv);
    }
}

// ---- multi-query attention with the K / V operands prepared ONCE per launch (p.kv_prep): flash_kv_prep_kernel splits, scales and transposes
// every 64-row context chunk into the operand image the matrix loop reads -- flash_attn_f16x3_kernel's values, octet-major (below): [K hi | K lo | V^T hi | V^T lo], 4 x 8 x 64
// 16-byte chunks, plus the chunk's two block-scaling exponents -- and flash_attn_mq_kernel copies a chunk global -> LDS by LDS-DMA into a
// double buffer, the next chunk under the current one's matrix work, one barrier per chunk.  In the staged form every workgroup (64 queries)
// re-did that preparation for all 65 chunks of a 4096-token context: two reductions, the split, 2-byte scattered LDS writes and three barriers
// per chunk, 64 times per image -- 11.5 us per chunk and workgroup against ~3 us of LDS reads + matrix work.  Same arithmetic, same bits.
// the prepared image of a 64-row chunk: four operand arrays [K hi | K lo | V^T hi | V^T lo], each [octet of the contracted index 8][row 64] 16-byte chunks
// (octet-major: the lanes of two neighbouring octets that a ds_read_b128 serves together -- {0-3, 12-15, 20-27}, ... -- then fall on disjoint banks;
// the row-major pitch-9 layout of the self-staging kernel is a 2-way conflict on this machine's lane groups: 46 % of the LDS cycles, PMC)
constexpr int FW_PL = 8 * 64, FW_CHUNK16 = 4 * FW_PL;          // 16-byte chunks per array / per prepared context chunk (32 768 bytes)

__global__ __launch_bounds__(256) void flash_kv_prep_kernel(const mi_flash_attn_params p, const int nchunk) {
    constexpr int EPT = 16, PPR = 4;
    __shared__ __attribute__((aligned(16))) uint4 img[FW_CHUNK16];          // KsH | KsL | VtH | VtL
    __shared__ float smax[2][4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z, j0 = 64 * c;       // kvh: which k / v head (0 for the multi-query form)
    const int KVH = gridDim.y;
    const int nnull = p.null_k ? 1 : 0, J = nnull + p.n0 + p.n1;
    const int srow = tid / PPR, sd0 = (tid % PPR) * EPT;
    const int spos = (srow & 32) | (((srow >> 2) & 3) << 3) | (((srow >> 4) & 1) << 2) | (srow & 3);
    float kf[EPT], vf[EPT];
    {
        const int jj = j0 + srow;
        const float* ksrc = nullptr;
        const float* vsrc = nullptr;
        if (jj < J) {
            if (jj < nnull) { ksrc = p.null_k; vsrc = p.null_v; }
            else if (jj - nnull < p.n0) { const size_t o_ = (size_t)b * p.bs0 + (size_t)(jj - nnull) * p.ld0 + kvh * 64; ksrc = p.k0 + o_; vsrc = p.v0 + o_; }
            else { const size_t o_ = (size_t)b * p.bs1 + (size_t)(jj - nnull - p.n0) * p.ld1 + kvh * 64; ksrc = p.k1 + o_; vsrc = p.v1 + o_; }
        }
#pragma unroll
        for (int e = 0; e < EPT; e += 4) {
            float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = k4;
            if (ksrc) { k4 = *reinterpret_cast<const float4*>(ksrc + sd0 + e); v4 = *reinterpret_cast<const float4*>(vsrc + sd0 + e); }
            kf[e] = k4.x; kf[e + 1] = k4.y; kf[e + 2] = k4.z; kf[e + 3] = k4.w;
            vf[e] = v4.x; vf[e + 1] = v4.y; vf[e + 2] = v4.z; vf[e + 3] = v4.w;
        }
    }
    float mk = 0.0f, mv = 0.0f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) { mk = fmaxf(mk, fabsf(kf[e])); mv = fmaxf(mv, fabsf(vf[e])); }
    mk = mi_wave_max(mk); mv = mi_wave_max(mv);
    for (int i = tid; i < FW_CHUNK16; i += 256) img[i] = make_uint4(0u, 0u, 0u, 0u);      // (the pad chunks are copied too: keep them defined)
    if (lane == 0) { smax[0][wave] = mk; smax[1][wave] = mv; }
    __syncthreads();
    const float mka = fmaxf(fmaxf(smax[0][0], smax[0][1]), fmaxf(smax[0][2], smax[0][3])), mva = fmaxf(fmaxf(smax[1][0], smax[1][1]), fmaxf(smax[1][2], smax[1][3]));
    const int ek = fw_scale_exp(mka), ev = fw_scale_exp(mva);
    {
        const float sk = ldexpf(1.0f, ek), sv = ldexpf(1.0f, ev);
        _Float16* ksh = reinterpret_cast<_Float16*>(img);
        _Float16* ksl = reinterpret_cast<_Float16*>(img + FW_PL);
        _Float16* vth = reinterpret_cast<_Float16*>(img + 2 * FW_PL);
        _Float16* vtl = reinterpret_cast<_Float16*>(img + 3 * FW_PL);
#pragma unroll
        for (int e = 0; e < EPT; e += 4) {
            unsigned hb[2], lb[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float x0 = kf[e + 2 * q] * sk, x1 = kf[e + 2 * q + 1] * sk;
                const mi_f16x2 h2 = {(_Float16)x0, (_Float16)x1};
                hb[q] = __builtin_bit_cast(unsigned, h2);
                lb[q] = mi_split_lo2(hb[q], x0, x1);
            }
            const int ko = ((((sd0 + e) >> 3) * 64 + srow) << 3) + ((sd0 + e) & 7);       // chunk (octet of d, row), halves within
            *reinterpret_cast<uint2*>(ksh + ko) = make_uint2(hb[0], hb[1]);
            *reinterpret_cast<uint2*>(ksl + ko) = make_uint2(lb[0], lb[1]);
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const float x = vf[e] * sv;
            const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
            const int vo = (((spos >> 3) * 64 + sd0 + e) << 3) + (spos & 7);             // chunk (octet of the permuted context row, d)
            vth[vo] = hi;
            vtl[vo] = lo;
        }
    }
    __syncthreads();
    const size_t img_i = ((size_t)b * KVH + kvh) * nchunk + c;
    uint4* dst = reinterpret_cast<uint4*>(p.kv_prep) + img_i * FW_CHUNK16;
    for (int i = tid; i < FW_CHUNK16; i += 256) dst[i] = img[i];
    if (tid == 0) {
        int* ex = reinterpret_cast<int*>(reinterpret_cast<uint4*>(p.kv_prep) + (size_t)p.B * KVH * nchunk * FW_CHUNK16) + img_i * 2;
        ex[0] = ek; ex[1] = ev;
    }
}

// PERHEAD (k / v per head, the wide presets' cross-attention): a workgroup = NW x QT x 16 queries of ONE head, which share that head's chunks
template <int NW, int QT, int WPS, bool PERHEAD = false>
__global__ __launch_bounds__(64 * NW, WPS) void flash_attn_mq_kernel(const mi_flash_attn_params p, const int nchunk) {
    // a workgroup = 64 queries x NH heads; a wave = QT 16-query tiles of one head: every K / V fragment read from LDS feeds QT x 3 matrix
    // instructions (with QT = 1 and 16 waves the LDS reads -- each wave reads the whole chunk -- took longer than the matrix work)
    constexpr int D = 64, WPH = 4 / QT, NH = NW / WPH;
    __shared__ __attribute__((aligned(16))) uint4 kv[2][FW_CHUNK16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int h = PERHEAD ? (int)blockIdx.y : (int)blockIdx.y * NH + wave / WPH, b = blockIdx.z;
    const int inner = p.heads * D;
    const int nnull = p.null_k ? 1 : 0, J = nnull + p.n0 + p.n1;
    const int KVH = PERHEAD ? p.heads : 1;
    const size_t img0 = ((size_t)b * KVH + (PERHEAD ? h : 0)) * nchunk;
    const uint4* const prep = reinterpret_cast<const uint4*>(p.kv_prep) + img0 * FW_CHUNK16;
    const int* const exps = reinterpret_cast<const int*>(reinterpret_cast<const uint4*>(p.kv_prep) + (size_t)p.B * KVH * nchunk * FW_CHUNK16) + img0 * 2;
    auto issue_chunk = [&](int c, int buf) {
        for (int r = wave; r < FW_CHUNK16 / 64; r += NW) {           // one 1 KB row (64 lanes x 16 bytes) per instruction
            const uint4* src = prep + (size_t)c * FW_CHUNK16 + r * 64 + lane;
#if defined(HIPEMU)
            kv[buf][r * 64 + lane] = *src;
#else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)&kv[buf][r * 64], 16, 0, 0);
#endif
        }
    };
    issue_chunk(0, 0);
    // Q as the B operand (as flash_attn_f16x3_kernel), one block-scaling exponent per 16-query tile
    fw_f16x8 qh[QT][2], ql[QT][2];
    int eq[QT], tok[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        tok[t] = (PERHEAD ? (int)blockIdx.x * (NW * QT) + wave * QT + t : (int)blockIdx.x * 4 + (wave % WPH) * QT + t) * 16 + lq;
        const int tokc = tok[t] < p.HW ? tok[t] : p.HW - 1;
        const float* qr = p.q + ((size_t)b * p.HW + tokc) * inner + h * D;
        float qv[2][8];
        float mq = 0.0f;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const float4 a = *reinterpret_cast<const float4*>(qr + 32 * hf + 8 * lg), c4 = *reinterpret_cast<const float4*>(qr + 32 * hf + 8 * lg + 4);
            const float w[8] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) { qv[hf][e] = w[e] * p.q_scale; mq = fmaxf(mq, fabsf(qv[hf][e])); }
        }
        eq[t] = fw_scale_exp(mi_wave_max(mq));
        const float sq = ldexpf(1.0f, eq[t]);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            float w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = qv[hf][e] * sq;
            uint4 hi, lo;
            fw_split8(w, hi, lo);
            qh[t][hf] = __builtin_bit_cast(fw_f16x8, hi);
            ql[t][hf] = __builtin_bit_cast(fw_f16x8, lo);
        }
    }
    float m[QT], l[QT];
    f32x4 o[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m[t] = -INFINITY; l[t] = 0.0f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // one context chunk; MASKED only for the last one (the only chunk that can hold rows >= J).  The block scale `us` is a power of two, so
    // max(s) * us and fma(s, us, -max) are the values max(s * us) and s * us - max of the self-staging kernel, without the multiplies.
    auto chunk = [&](const int c, auto masked) {
        const int j0 = 64 * c, buf = c & 1;
#if !defined(HIPEMU)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of chunk c has landed in LDS ...
#endif
        __syncthreads();                                  // ... everybody's has, and nobody reads chunk c - 1's buffer any more
        if (c + 1 < nchunk) issue_chunk(c + 1, buf ^ 1);
        const uint4* const KsH = kv[buf], * const KsL = kv[buf] + FW_PL, * const VtH = kv[buf] + 2 * FW_PL, * const VtL = kv[buf] + 3 * FW_PL;
        const int ek = exps[2 * c], ev = exps[2 * c + 1];
        const float uv = ldexpf(1.0f, -ev);
        f32x4 s[QT][4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
            for (int t = 0; t < QT; ++t) s[t][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const fw_f16x8 kh = __builtin_bit_cast(fw_f16x8, KsH[(4 * hf + lg) * 64 + 16 * jt + lq]), kl = __builtin_bit_cast(fw_f16x8, KsL[(4 * hf + lg) * 64 + 16 * jt + lq]);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    s[t][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[t][hf], s[t][jt], 0, 0, 0);
                    s[t][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[t][hf], s[t][jt], 0, 0, 0);
                    s[t][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[t][hf], s[t][jt], 0, 0, 0);
                }
            }
        }
        fw_f16x8 ph[QT][2], pl[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const float us = ldexpf(1.0f, -(ek + eq[t]));
            float mx = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (decltype(masked)::value && j0 + 16 * jt + 4 * lg + r >= J) s[t][jt][r] = -INFINITY;
                    mx = fmaxf(mx, s[t][jt][r]);
                }
            mx *= us;
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(m[t], mx);
            if (__any(mn != m[t])) {                        // (else every lane's alpha is exp2(0) = 1)
                const float alpha = __builtin_amdgcn_exp2f(m[t] - mn);
                m[t] = mn;
                l[t] = mi_mul_rounded(l[t], alpha);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[t][dt][r] *= alpha;
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float pe[8];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { pe[4 * u + r] = __builtin_amdgcn_exp2f(fmaf(s[t][2 * hf + u][r], us, -mn)); l[t] += pe[4 * u + r]; }
                uint4 hi, lo;
                fw_split8(pe, hi, lo);
                ph[t][hf] = __builtin_bit_cast(fw_f16x8, hi);
                pl[t][hf] = __builtin_bit_cast(fw_f16x8, lo);
            }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4 sl[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t) sl[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const fw_f16x8 vh = __builtin_bit_cast(fw_f16x8, VtH[(4 * hf + lg) * 64 + 16 * dt + lq]), vl = __builtin_bit_cast(fw_f16x8, VtL[(4 * hf + lg) * 64 + 16 * dt + lq]);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    sl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[t][hf], sl[t], 0, 0, 0);
                    sl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl[t][hf], sl[t], 0, 0, 0);
                    sl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph[t][hf], sl[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[t][dt][r] = fmaf(sl[t][r], uv, o[t][dt][r]);
        }
    };
    for (int c = 0; c + 1 < nchunk; ++c) chunk(c, std::false_type{});
    chunk(nchunk - 1, std::true_type{});
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float lt = l[t];
        lt += __shfl_xor(lt, 16);
        lt += __shfl_xor(lt, 32);
        const float linv = 1.0f / lt;
        if (tok[t] < p.HW) {
            float* orow = p.out + ((size_t)b * p.HW + tok[t]) * inner + h * D;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<float4*>(orow + 16 * dt + 4 * lg) = make_float4(o[t][dt][0] * linv, o[t][dt][1] * linv, o[t][dt][2] * linv, o[t][dt][3] * linv);
        }
    }
}

// (Tried and removed, in the git history: the same kernel with the two waves of every SIMD half a chunk apart -- waves 0-3 in the matrix block
// [PV(i - 1), QK^T(i)] while waves 4-7 run the softmax of chunk i - 1, a barrier at every hand-over, bit-identical.  3.23 ms against 3.19 ms per
// 4096-token launch: on this machine the matrix time and the VALU issue time of a SIMD's waves add up whichever wave they come from
// (96 matrix instructions x 16 cycles + ~270 VALU x 4 per wave and chunk = the measured 6 000 cycles per chunk and wave pair);
// profiles/r05_wide_flash_kv_prep_ab.txt.)

// ---- tokens [B][HW][C] -> NCHW, with an optional LayerNorm over C (gamma, beta) in front, a residual NCHW tensor added and the next
// GroupNorm's partial statistics (64-token tiles) emitted: to_out.1 + residual of both attentions, and the tail of ChanFeedForward
// A workgroup = 64 tokens.  Token rows are read channel-contiguous (a wave per token row: coalesced), the 64 x 64 (token, channel) blocks go
// through an LDS transpose (pitch 65) and leave token-contiguous per channel (a wave per channel: coalesced NCHW stores, wave-level statistics).
__global__ __launch_bounds__(256) void tokens_to_nchw_kernel(const mi_tokens_to_nchw_params p) {
    __shared__ float sMean[64], sRstd[64];
    __shared__ float tileS[64][65];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int tok0 = tile * 64;
    const float* base = p.tokens + (size_t)b * p.HW * p.C;
    if (p.gamma) {
        // per-token moments, a wave per token: mean, then the variance of the centred values (the reference's torch.var, layers.py:341-343)
        // (four rows at a time: their loads are in flight together -- a row's two passes are a dependent chain of memory round trips, and 16 of
        // them in sequence were a third of the kernel's time on the small images)
        for (int t0 = wave; t0 < 64; t0 += 16) {
            const float* row[4];
            float sm[4], mean[4], v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = t0 + 4 * k, tk = tok0 + t < p.HW ? tok0 + t : p.HW - 1;
                row[k] = base + (size_t)tk * p.C;
                sm[k] = 0.f; v[k] = 0.f;
            }
            for (int c = lane; c < p.C; c += 64) {
#pragma unroll
                for (int k = 0; k < 4; ++k) sm[k] += row[k][c];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) mean[k] = mi_wave_sum(sm[k]) / (float)p.C;
            for (int c = lane; c < p.C; c += 64) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float d = row[k][c] - mean[k]; v[k] = fmaf(d, d, v[k]); }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float vs = mi_wave_sum(v[k]);
                if (lane == 0) { sMean[t0 + 4 * k] = mean[k]; sRstd[t0 + 4 * k] = 1.0f / sqrtf(vs / (float)p.C + p.eps); }
            }
        }
    }
    const int nt = (p.HW + 63) / 64;
    const int br = p.res.data ? mi_row_of(b, p.res.bmod) : 0;
    const int tok = tok0 + lane;
    const bool ok = tok < p.HW;
    // blockIdx.z: which share of the 64-channel blocks (small token counts -- 16 x 16 images -- would otherwise leave half of the CUs without a
    // workgroup: 4 tiles x 32 images); every share computes the token moments for itself
    const int nblk = (p.C + 63) / 64, per = (nblk + (int)gridDim.z - 1) / (int)gridDim.z;
    const int c_lo = (int)blockIdx.z * per * 64, c_hi = c_lo + per * 64 < p.C ? c_lo + per * 64 : p.C;
    for (int c0 = c_lo; c0 < c_hi; c0 += 64) {
        __syncthreads();                                  // moments published / the previous block's reads are done
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int t = wave * 16 + i;
            tileS[t][lane] = (tok0 + t < p.HW && c0 + lane < p.C) ? base[(size_t)(tok0 + t) * p.C + c0 + lane] : 0.f;
        }
        __syncthreads();
        for (int i = 0; i < 16; ++i) {
            const int c = c0 + wave * 16 + i;
            if (c >= p.C) break;                          // wave-uniform
            float y = 0.f;
            if (ok) {
                y = tileS[lane][wave * 16 + i];
                if (p.gamma) y = (y - sMean[lane]) * sRstd[lane] * p.gamma[c] + (p.beta ? p.beta[c] : 0.0f);
                if (p.res.data) y += p.res.data[((size_t)br * p.C + c) * p.HW + tok] * p.res.scale;
                p.out[((size_t)b * p.C + c) * p.HW + tok] = y;
            }
            if (p.out_stats) {
                double s_, q_;
                mi_stat_reduce64(y, ok, s_, q_);
                if (lane == 0) {
                    p.out_stats[((size_t)(b * p.C + c) * nt + tile) * 2] = s_;
                    p.out_stats[((size_t)(b * p.C + c) * nt + tile) * 2 + 1] = q_;
                }
            }
        }
    }
}

// ---- LayerNorm over the last dimension of [rows][dim] (ChanLayerNorm of ChanFeedForward in token layout: gamma only).  A wave per row, no LDS and no
// barriers (a 256-thread workgroup per 128..512-wide row left half of its lanes idle between four barriers: 152 us for the 131072 rows of the 64 x 64 level)
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ y, int rows, int dim, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * dim;
    float s = 0.f;
    for (int i = lane; i < dim; i += 64) s += xr[i];
    const float mean = mi_wave_sum(s) / (float)dim;
    float v = 0.f;
    for (int i = lane; i < dim; i += 64) { const float d = xr[i] - mean; v = fmaf(d, d, v); }
    const float rstd = 1.0f / sqrtf(mi_wave_sum(v) / (float)dim + eps);
    for (int i = lane; i < dim; i += 64) y[(size_t)row * dim + i] = (xr[i] - mean) * rstd * gamma[i] + (beta ? beta[i] : 0.0f);
}

}  // namespace

extern "C" int mi_flash_attn_fwd(const mi_flash_attn_params* p, void* stream) {
    if (p->B <= 0 || p->HW <= 0 || p->heads <= 0 || (p->kv_heads != 1 && p->kv_heads != p->heads)) { mi_set_error("mi_flash_attn_fwd: bad shape"); return MI_ERR_INVALID; }
    if (p->n0 + p->n1 + (p->null_k ? 1 : 0) <= 0 || (p->null_k && !p->null_v)) { mi_set_error("mi_flash_attn_fwd: empty context"); return MI_ERR_INVALID; }
    if ((p->ld0 & 3) || (p->n1 && (p->ld1 & 3))) { mi_set_error("mi_flash_attn_fwd: row strides must be multiples of 4 floats"); return MI_ERR_INVALID; }
    // 3-term fp16 products on the f16 matrix-core instruction (block-scaled per chunk); MI_FLASH_EXACT_F32 in the environment selects the
    // exact-fp32 MFMA kernel (A/B measurements)
    static const bool exact = getenv("MI_FLASH_EXACT_F32") != nullptr;
    static const bool one_head = getenv("MI_FLASH_ONE_HEAD") != nullptr;
    if (exact) hipLaunchKernelGGL(flash_attn_kernel, dim3((p->HW + 63) / 64, p->heads, p->B), dim3(256), 0, (hipStream_t)stream, *p);
    else if (p->kv_prep && !one_head && ((p->kv_heads == 1 && (p->heads & 3) == 0) || p->kv_heads == p->heads)) {
        // prepared K / V (flash_kv_prep_kernel): multi-query (four heads of 64 queries per workgroup) or one k / v head per head (256 queries of a head)
        const int nnull = p->null_k ? 1 : 0, nchunk = (nnull + p->n0 + p->n1 + 63) / 64;
        if (p->kv_prep_bytes < mi_flash_kv_prep_bytes(p->B * p->kv_heads, nnull + p->n0 + p->n1)) { mi_set_error("mi_flash_attn_fwd: kv_prep buffer too small"); return MI_ERR_INVALID; }
        hipLaunchKernelGGL(flash_kv_prep_kernel, dim3(nchunk, p->kv_heads, p->B), dim3(256), 0, (hipStream_t)stream, *p, nchunk);
        static const int qt = getenv("MI_FLASH_MQ_QT") ? atoi(getenv("MI_FLASH_MQ_QT")) : 2;         // 16-query tiles per wave (A/B knob: 1 = 16 waves of one tile each)
        if (!(p->kv_heads == 1 && (p->heads & 3) == 0)) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(flash_attn_mq_kernel<8, 2, 2, true>), dim3((p->HW + 255) / 256, p->heads, p->B), dim3(512), 0, (hipStream_t)stream, *p, nchunk);
        } else {
            const dim3 grid((p->HW + 63) / 64, p->heads / 4, p->B);
            if (qt == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(flash_attn_mq_kernel<16, 1, 4>), grid, dim3(1024), 0, (hipStream_t)stream, *p, nchunk);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(flash_attn_mq_kernel<8, 2, 2>), grid, dim3(512), 0, (hipStream_t)stream, *p, nchunk);
        }
    }
    else if (p->kv_heads == 1 && (p->heads & 3) == 0 && !one_head) {    // multi-query: heads of the same 64 queries share every staged K / V chunk
        static const int two_heads = getenv("MI_FLASH_TWO_HEADS") ? atoi(getenv("MI_FLASH_TWO_HEADS")) : 0;
        if (two_heads) hipLaunchKernelGGL(HIP_KERNEL_NAME(flash_attn_f16x3_kernel<8>), dim3((p->HW + 63) / 64, p->heads / 2, p->B), dim3(512), 0, (hipStream_t)stream, *p);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(flash_attn_f16x3_kernel<16>), dim3((p->HW + 63) / 64, p->heads / 4, p->B), dim3(1024), 0, (hipStream_t)stream, *p);
    }
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(flash_attn_f16x3_kernel<4>), dim3((p->HW + 63) / 64, p->heads, p->B), dim3(256), 0, (hipStream_t)stream, *p);
    return mi_check_launch("flash_attn_kernel");
}

extern "C" long long mi_flash_kv_prep_bytes(int B, int J) {
    const long long nchunk = (J + 63) / 64;
    return (long long)B * nchunk * (FW_CHUNK16 * 16 + 8);
}

extern "C" int mi_tokens_to_nchw_fwd(const mi_tokens_to_nchw_params* p, void* stream) {
    if (p->B <= 0 || p->HW <= 0 || p->C <= 0) { mi_set_error("mi_tokens_to_nchw_fwd: empty problem"); return MI_ERR_INVALID; }
    const int nt = (p->HW + 63) / 64, nblk = (p->C + 63) / 64;
    int split = 1;
    while (split < nblk && (long long)nt * p->B * split < 512) split *= 2;
    hipLaunchKernelGGL(tokens_to_nchw_kernel, dim3(nt, p->B, split), dim3(256), 0, (hipStream_t)stream, *p);
    return mi_check_launch("tokens_to_nchw_kernel");
}

extern "C" int mi_ln_rows_fwd(const float* x, const float* gamma, const float* beta, float* y, int rows, int dim, float eps, void* stream) {
    if (rows <= 0 || dim <= 0) { mi_set_error("mi_ln_rows_fwd: empty problem"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(ln_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, rows, dim, eps);
    return mi_check_launch("ln_rows_kernel");
}
