// Wide-channel attention (Unet() default, Base, Super: C > 32): the unfolded forms of CrossAttention (layers.py:220-251), the
// multi-query Attention (layers.py:52-104) and ChanFeedForward (layers.py:148-161) as token-major building blocks:
//   LayerNorm tokens (attention.hip) -> projections (mi_gemm_f32, t5.hip) -> mi_flash_attn_fwd -> projection -> mi_tokens_to_nchw_fwd.
// The folded kernels of attention.hip pay K = C per score, which only wins while C <= 32; from 64 channels up the reference's own
// factorisation (dim_head 64) is the cheaper one.  Everything here is exact fp32 on v_mfma_f32_16x16x4_f32 (no operand splits).
#include "common.hip.h"

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

// ---- tokens [B][HW][C] -> NCHW, with an optional LayerNorm over C (gamma, beta) in front, a residual NCHW tensor added and the next
// GroupNorm's partial statistics (64-token tiles) emitted: to_out.1 + residual of both attentions, and the tail of ChanFeedForward
__global__ __launch_bounds__(256) void tokens_to_nchw_kernel(const mi_tokens_to_nchw_params p) {
    __shared__ float sMean[64], sRstd[64];
    const int tid = threadIdx.x, tok_l = tid & 63, cq = tid >> 6;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int tok = tile * 64 + tok_l;
    const bool ok = tok < p.HW;
    const float* trow = p.tokens + ((size_t)b * p.HW + (ok ? tok : p.HW - 1)) * p.C;
    if (p.gamma) {
        // per-token moments: the four work-items of a token each take a quarter of the channels
        float s = 0.f, q = 0.f;
        for (int c = cq; c < p.C; c += 4) { const float v = trow[c]; s += v; q = fmaf(v, v, q); }
        __shared__ float ps[4][64], pq[4][64];
        ps[cq][tok_l] = s; pq[cq][tok_l] = q;
        __syncthreads();
        if (cq == 0) {
            const float S = (ps[0][tok_l] + ps[1][tok_l]) + (ps[2][tok_l] + ps[3][tok_l]);
            const float mean = S / (float)p.C;
            // two-pass variance (the reference's torch.var over the centred values, layers.py:341-343)
            float v = 0.f;
            for (int c = 0; c < p.C; ++c) { const float d = trow[c] - mean; v = fmaf(d, d, v); }
            sMean[tok_l] = mean;
            sRstd[tok_l] = 1.0f / sqrtf(v / (float)p.C + p.eps);
        }
        __syncthreads();
    }
    const int nt = (p.HW + 63) / 64;
    const int br = p.res.data ? mi_row_of(b, p.res.bmod) : 0;
    for (int c = cq; c < p.C; c += 4) {              // a wave = the 64 tokens of one channel: coalesced NCHW writes, wave-level statistics
        float y = 0.f;
        if (ok) {
            y = trow[c];
            if (p.gamma) y = (y - sMean[tok_l]) * sRstd[tok_l] * p.gamma[c] + (p.beta ? p.beta[c] : 0.0f);
            if (p.res.data) y += p.res.data[((size_t)br * p.C + c) * p.HW + tok] * p.res.scale;
            p.out[((size_t)b * p.C + c) * p.HW + tok] = y;
        }
        if (p.out_stats) {
            const float s = mi_wave_sum(y), q = mi_wave_sum(y * y);
            if (tok_l == 0) {
                p.out_stats[((size_t)(b * p.C + c) * nt + tile) * 2] = s;
                p.out_stats[((size_t)(b * p.C + c) * nt + tile) * 2 + 1] = q;
            }
        }
    }
}

// ---- LayerNorm over the last dimension of [rows][dim] (ChanLayerNorm of ChanFeedForward in token layout: gamma only)
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ y, int dim, float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* xr = x + (size_t)row * dim;
    float s = 0.f;
    for (int i = tid; i < dim; i += 256) s += xr[i];
    s = mi_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)dim;
    __syncthreads();
    float v = 0.f;
    for (int i = tid; i < dim; i += 256) { const float d = xr[i] - mean; v = fmaf(d, d, v); }
    v = mi_wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)dim + eps);
    for (int i = tid; i < dim; i += 256) y[(size_t)row * dim + i] = (xr[i] - mean) * rstd * gamma[i] + (beta ? beta[i] : 0.0f);
}

}  // namespace

extern "C" int mi_flash_attn_fwd(const mi_flash_attn_params* p, void* stream) {
    if (p->B <= 0 || p->HW <= 0 || p->heads <= 0 || (p->kv_heads != 1 && p->kv_heads != p->heads)) { mi_set_error("mi_flash_attn_fwd: bad shape"); return MI_ERR_INVALID; }
    if (p->n0 + p->n1 + (p->null_k ? 1 : 0) <= 0 || (p->null_k && !p->null_v)) { mi_set_error("mi_flash_attn_fwd: empty context"); return MI_ERR_INVALID; }
    if ((p->ld0 & 3) || (p->n1 && (p->ld1 & 3))) { mi_set_error("mi_flash_attn_fwd: row strides must be multiples of 4 floats"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(flash_attn_kernel, dim3((p->HW + 63) / 64, p->heads, p->B), dim3(256), 0, (hipStream_t)stream, *p);
    return mi_check_launch("flash_attn_kernel");
}

extern "C" int mi_tokens_to_nchw_fwd(const mi_tokens_to_nchw_params* p, void* stream) {
    if (p->B <= 0 || p->HW <= 0 || p->C <= 0) { mi_set_error("mi_tokens_to_nchw_fwd: empty problem"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(tokens_to_nchw_kernel, dim3((p->HW + 63) / 64, p->B), dim3(256), 0, (hipStream_t)stream, *p);
    return mi_check_launch("tokens_to_nchw_kernel");
}

extern "C" int mi_ln_rows_fwd(const float* x, const float* gamma, const float* beta, float* y, int rows, int dim, float eps, void* stream) {
    if (rows <= 0 || dim <= 0) { mi_set_error("mi_ln_rows_fwd: empty problem"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(ln_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, dim, eps);
    return mi_check_launch("ln_rows_kernel");
}
