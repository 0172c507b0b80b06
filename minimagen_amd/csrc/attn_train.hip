// Training path: the core of the FOLDED cross-attention (layers.py:220-251 with keys / values mapped into the token's
// channel space, layers.CrossAttention._forward_folded) forward and backward without ever materialising the [tokens x heads x context] score
// tensor:      out_i = sum_h sum_j softmax_j(q_i . kf_hj) vf_hj ,          q [B][n][C], kf / vf [B][H][J][C] (C = 8 / 16 / 32), mask [B][J].
// The problem is tiny per score (C multiply-adds) and has no reuse the matrix cores could exploit at C = 16, so these are fp32 VALU kernels:
//   folded_attn_fwd_kernel    a work-item per token, kf / vf of one (row, head) in LDS read as broadcasts, online softmax over blocks of 8
//                             context rows; saves the logsumexp and the head's own output per (token, head)
//   folded_attn_dq_kernel     a work-item per token: recomputes the probabilities, D = dO . O_h, dq = sum_h sum_j p (dP - D) kf_hj; saves D
//   folded_attn_dkv_kernel    a work-item per context row j of one (row, head, token chunk): dkf_j = sum_i dS_ij q_i, dvf_j = sum_i p_ij dO_i
//                             over the chunk's tokens (staged in LDS, broadcast reads); chunk partials are added by the caller (fixed order)
// exp through the hardware exp2 on log2(e)-scaled scores (fp32 throughout).
#include "common.hip.h"

namespace {

constexpr float AT_LOG2E = 1.44269504088896340736f;
constexpr int AT_CAP = 6144;        // floats of one (row, head)'s keys (and of its values) in LDS: J * C <= 6144 (C = 16: 384 context rows)
constexpr int AT_JMAX = 1024;

template <int CC>
__device__ __forceinline__ float dotc(const float (&a)[CC], const float* b) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CC; ++c) s = fmaf(a[c], b[c], s);
    return s;
}

// grid (ceil(n / 256), B), block 256
template <int CC>
__global__ __launch_bounds__(256) void folded_attn_fwd_kernel(mi_folded_attn_params p) {
    __shared__ float ks[AT_CAP], vs[AT_CAP], live[AT_JMAX];          // kf [J][CC], vf [J][CC] of one (row, head); live context rows
    const int b = blockIdx.y, tok = blockIdx.x * 256 + threadIdx.x;
    const bool ok = tok < p.n;
    float q[CC], o[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) { q[c] = ok ? p.q[((size_t)b * p.n + tok) * CC + c] * AT_LOG2E : 0.f; o[c] = 0.f; }
    for (int j = threadIdx.x; j < p.J; j += 256) live[j] = (p.mask == nullptr || p.mask[(size_t)b * p.J + j]) ? 1.f : 0.f;
    for (int h = 0; h < p.H; ++h) {
        __syncthreads();
        const size_t base = ((size_t)b * p.H + h) * p.J * CC;
        for (int i = threadIdx.x; i < p.J * CC; i += 256) { ks[i] = p.kf[base + i]; vs[i] = p.vf[base + i]; }
        __syncthreads();
        // online softmax over blocks of 8 context rows: every score is computed once; the running sums are rescaled once per block
        float m = -INFINITY, l = 0.f, acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = 0.f;
        for (int j0 = 0; j0 < p.J; j0 += 8) {
            float sc[8];
            float mb = m;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int j = j0 + t;
                sc[t] = (j < p.J && live[j] != 0.f) ? dotc<CC>(q, ks + j * CC) : -INFINITY;
                mb = fmaxf(mb, sc[t]);
            }
            if (mb == -INFINITY) continue;              // nothing live so far (wave-uniform: the mask is per image)
            const float alpha = __builtin_amdgcn_exp2f(m - mb);      // m = -inf -> 0
            m = mb;
            l *= alpha;
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] *= alpha;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int j = (j0 + t < p.J) ? j0 + t : p.J - 1;
                const float pj = __builtin_amdgcn_exp2f(sc[t] - m);          // exp2(-inf) = 0 for masked / out-of-range rows
                l += pj;
#pragma unroll
                for (int c = 0; c < CC; ++c) acc[c] = fmaf(pj, vs[j * CC + c], acc[c]);
            }
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int c = 0; c < CC; ++c) { acc[c] *= inv; o[c] += acc[c]; }
        if (ok && p.oh) {                                  // the head's own output: the backward's D = dO . O_h needs no extra pass
#pragma unroll
            for (int c = 0; c < CC; ++c) p.oh[(((size_t)b * p.n + tok) * p.H + h) * CC + c] = acc[c];
        }
        if (ok) p.lse[((size_t)b * p.n + tok) * p.H + h] = m + log2f(l);       // log2 domain: p = exp2(s * log2e - lse)
    }
    if (ok) {
#pragma unroll
        for (int c = 0; c < CC; ++c) p.out[((size_t)b * p.n + tok) * CC + c] = o[c];
    }
}

// grid (ceil(n / 256), B), block 256
template <int CC>
__global__ __launch_bounds__(256) void folded_attn_dq_kernel(mi_folded_attn_params p) {
    __shared__ float ks[AT_CAP], vs[AT_CAP], live[AT_JMAX];
    const int b = blockIdx.y, tok = blockIdx.x * 256 + threadIdx.x;
    const bool ok = tok < p.n;
    const size_t row = (size_t)b * p.n + (ok ? tok : 0);
    float q[CC], g[CC], dq[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) { q[c] = p.q[row * CC + c] * AT_LOG2E; g[c] = p.dout[row * CC + c]; dq[c] = 0.f; }
    for (int j = threadIdx.x; j < p.J; j += 256) live[j] = (p.mask == nullptr || p.mask[(size_t)b * p.J + j]) ? 1.f : 0.f;
    for (int h = 0; h < p.H; ++h) {
        __syncthreads();
        const size_t base = ((size_t)b * p.H + h) * p.J * CC;
        for (int i = threadIdx.x; i < p.J * CC; i += 256) { ks[i] = p.kf[base + i]; vs[i] = p.vf[base + i]; }
        __syncthreads();
        const float lse = p.lse[row * p.H + h];
        float D = 0.f;
        if (p.oh) {                                        // D = sum_j p_j dP_j = dO . (sum_j p_j vf_j) = dO . O_h
            D = dotc<CC>(g, p.oh + (row * p.H + h) * CC);
        } else {
            for (int j = 0; j < p.J; ++j) {
                if (live[j] == 0.f) continue;
                const float pj = __builtin_amdgcn_exp2f(dotc<CC>(q, ks + j * CC) - lse);
                D = fmaf(pj, dotc<CC>(g, vs + j * CC), D);
            }
        }
        for (int j = 0; j < p.J; ++j) {
            if (live[j] == 0.f) continue;
            const float pj = __builtin_amdgcn_exp2f(dotc<CC>(q, ks + j * CC) - lse);
            const float ds = pj * (dotc<CC>(g, vs + j * CC) - D);
#pragma unroll
            for (int c = 0; c < CC; ++c) dq[c] = fmaf(ds, ks[j * CC + c], dq[c]);
        }
        if (ok) p.dsum[row * p.H + h] = D;
    }
    if (ok) {
#pragma unroll
        for (int c = 0; c < CC; ++c) p.dq[row * CC + c] = dq[c];
    }
}

// grid (nchunk, H, B), block 64 * ceil(J / 64): a work-item per context row
template <int CC>
__global__ __launch_bounds__(1024) void folded_attn_dkv_kernel(mi_folded_attn_params p) {
    __shared__ float qs[64 * CC], gs[64 * CC], ls[64], ds_[64];
    const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z, j = threadIdx.x;
    const bool row = j < p.J && (p.mask == nullptr || p.mask[(size_t)b * p.J + j]);
    float k[CC], v[CC], dk[CC], dv[CC];
    const size_t base = (((size_t)b * p.H + h) * p.J + (j < p.J ? j : 0)) * CC;
#pragma unroll
    for (int c = 0; c < CC; ++c) { k[c] = p.kf[base + c] * AT_LOG2E; v[c] = p.vf[base + c]; dk[c] = 0.f; dv[c] = 0.f; }
    const int per = (p.n + gridDim.x - 1) / gridDim.x;
    const int i0 = chunk * per, i1 = (i0 + per < p.n) ? i0 + per : p.n;
    for (int t0 = i0; t0 < i1; t0 += 64) {
        const int nt = (i1 - t0 < 64) ? i1 - t0 : 64;
        __syncthreads();
        for (int i = threadIdx.x; i < nt * CC; i += blockDim.x) {
            qs[i] = p.q[((size_t)b * p.n + t0) * CC + i];
            gs[i] = p.dout[((size_t)b * p.n + t0) * CC + i];
        }
        for (int i = threadIdx.x; i < nt; i += blockDim.x) {
            ls[i] = p.lse[((size_t)b * p.n + t0 + i) * p.H + h];
            ds_[i] = p.dsum[((size_t)b * p.n + t0 + i) * p.H + h];
        }
        __syncthreads();
        if (row) {
            for (int i = 0; i < nt; ++i) {
                const float pj = __builtin_amdgcn_exp2f(dotc<CC>(k, qs + i * CC) - ls[i]);
                const float dsv = pj * (dotc<CC>(v, gs + i * CC) - ds_[i]);
#pragma unroll
                for (int c = 0; c < CC; ++c) { dk[c] = fmaf(dsv, qs[i * CC + c], dk[c]); dv[c] = fmaf(pj, gs[i * CC + c], dv[c]); }
            }
        }
    }
    if (j < p.J) {
        const size_t o = ((((size_t)chunk * gridDim.z + b) * p.H + h) * p.J + j) * CC;
#pragma unroll
        for (int c = 0; c < CC; ++c) { p.dkf[o + c] = dk[c]; p.dvf[o + c] = dv[c]; }
    }
}

template <int CC>
int launch_folded(const mi_folded_attn_params& p, int which, hipStream_t st) {
    const dim3 gtok((p.n + 255) / 256, p.B);
    if (which == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(folded_attn_fwd_kernel<CC>), gtok, dim3(256), 0, st, p);
    else if (which == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(folded_attn_dq_kernel<CC>), gtok, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(folded_attn_dkv_kernel<CC>), dim3(p.nchunk, p.H, p.B), dim3(64 * ((p.J + 63) / 64)), 0, st, p);
    return mi_check_launch(which == 0 ? "folded_attn_fwd_kernel" : (which == 1 ? "folded_attn_dq_kernel" : "folded_attn_dkv_kernel"));
}

int folded_dispatch(const mi_folded_attn_params* q, int which, void* stream) {
    if (!q || q->B <= 0 || q->n <= 0 || q->H <= 0 || q->J <= 0 || !q->q || !q->kf || !q->vf || !q->lse) { mi_set_error("mi_folded_attn: bad arguments"); return MI_ERR_INVALID; }
    if (q->J > AT_JMAX || q->J * q->C > AT_CAP) { mi_set_error("mi_folded_attn: context of %d rows x %d channels does not fit (J * C <= %d)", q->J, q->C, AT_CAP); return MI_ERR_UNSUPPORTED; }
    if (which == 0 && !q->out) { mi_set_error("mi_folded_attn_fwd: out missing"); return MI_ERR_INVALID; }
    if (which >= 1 && (!q->dout || !q->dsum)) { mi_set_error("mi_folded_attn_bwd: dout / dsum missing"); return MI_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    switch (q->C) {
        case 8: return launch_folded<8>(*q, which, st);
        case 16: return launch_folded<16>(*q, which, st);
        case 32: return launch_folded<32>(*q, which, st);
    }
    mi_set_error("mi_folded_attn: C = %d not instantiated (8, 16, 32)", q->C);
    return MI_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int mi_folded_attn_fwd(const mi_folded_attn_params* p, void* stream) { return folded_dispatch(p, 0, stream); }

extern "C" int mi_folded_attn_bwd(const mi_folded_attn_params* p, void* stream) {
    if (p && (!p->dq || !p->dkf || !p->dvf || p->nchunk <= 0)) { mi_set_error("mi_folded_attn_bwd: dq / dkf / dvf / nchunk missing"); return MI_ERR_INVALID; }
    const int rc = folded_dispatch(p, 1, stream);
    return rc ? rc : folded_dispatch(p, 2, stream);
}
