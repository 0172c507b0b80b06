// Training path: the core of the FOLDED cross-attention (layers.py:220-251 with keys / values mapped into the token's
// channel space, layers.CrossAttention._forward_folded) forward and backward without ever materialising the [tokens x heads x context] score
// tensor:      out_i = sum_h sum_j softmax_j(q_i . kf_hj) vf_hj ,          q [B][n][C], kf / vf [B][H][J][C] (C = 8 / 16 / 32), mask [B][J].
// The problem is tiny per score (C multiply-adds).  C = 8 / 32 (and MI_FOLDED_ATTN_VALU=1, for A/B runs): fp32 VALU kernels --
//   folded_attn_fwd_kernel    a work-item per token, kf / vf of one (row, head) in LDS read as broadcasts, online softmax over blocks of 8
//                             context rows; saves the logsumexp and the head's own output per (token, head)
//   folded_attn_dq_kernel     a work-item per token: recomputes the probabilities, D = dO . O_h, dq = sum_h sum_j p (dP - D) kf_hj; saves D
//   folded_attn_dkv_kernel    a work-item per context row j of one (row, head, token chunk): dkf_j = sum_i dS_ij q_i, dvf_j = sum_i p_ij dO_i
//                             over the chunk's tokens (staged in LDS, broadcast reads); chunk partials are added by the caller (fixed order)
// exp through the hardware exp2 on log2(e)-scaled scores (fp32 throughout).
// C = 16 (the BASELINE U-Nets): the same three kernels on v_mfma_f32_16x16x4_f32 (exact fp32 products), further down: folded_attn_fwd_mfma_kernel,
// folded_attn_dq_mfma_kernel, folded_attn_dkv_mfma_kernel.
#include "common.hip.h"
#include <cstdlib>

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


// ---- matrix cores for the forward at C = 16 (the BASELINE U-Nets' bottleneck cross-attention): exact-fp32 products on v_mfma_f32_16x16x4_f32, the
// arithmetic class of the VALU kernel above at twice its multiply-add rate and without its per-score LDS traffic.  A wave owns TT tiles of 16 tokens;
// per tile of 16 context rows:    S^T[j][i] = sum_c kf[j][c] q[i][c]     A = kf[j0 + lq][4 lg + kb], B = q[i0 + lq][4 lg + kb], kb = 0 .. 3
// -- the contraction index is (kb, lg) -> c = 4 lg + kb, so either operand is ONE 16-byte read per lane -- leaving lane (lq, lg) with the scores of
// token i0 + lq against rows j0 + 4 lg + r.  Those four registers are, as they lie, the B operands of    out^T[c'][i] += sum_j vf[j][c'] P^T[j][i]
// (instruction r contracts over j = j0 + 4 lg + r; A = vf[j0 + 4 lg + r][lq] from the [J/4][16][4] copy of vf: one 16-byte read), so the
// probabilities never move between lanes; the per-token running maximum is shared by the four lanes of a token with two cross-lane steps per tile,
// the running sum stays a per-lane partial until the head is done.  Online softmax as above (rescale once per 16 rows).
constexpr int ATM_KP = 20;                   // floats per kf row in LDS: the 16-byte reads of 16 consecutive rows spread over all banks

template <int TT, int NJT_MAX, int NW>
__global__ __launch_bounds__(64 * NW) void folded_attn_fwd_mfma_kernel(mi_folded_attn_params p) {
    constexpr int JP = 16 * NJT_MAX;
    __shared__ __attribute__((aligned(16))) float ks[JP * ATM_KP], vt[JP * 16], bias[JP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    constexpr int NT = 64 * NW;
    const int b = blockIdx.y, tok0 = (blockIdx.x * NW + wave) * (16 * TT);
    const int njt = (p.J + 15) >> 4;
    for (int j = tid; j < 16 * njt; j += NT) bias[j] = (j < p.J && (p.mask == nullptr || p.mask[(size_t)b * p.J + j])) ? 0.f : -INFINITY;
    float4 q4[TT];
    f32x4 otot[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const int i = tok0 + 16 * t + lq;
        const float4 u = mi_ldg4(p.q + ((size_t)b * p.n + (i < p.n ? i : p.n - 1)) * 16 + 4 * lg);
        q4[t] = make_float4(u.x * AT_LOG2E, u.y * AT_LOG2E, u.z * AT_LOG2E, u.w * AT_LOG2E);
        otot[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int h = 0; h < p.H; ++h) {
        __syncthreads();
        const size_t base = ((size_t)b * p.H + h) * p.J * 16;
        for (int e = tid; e < 16 * njt * 4; e += NT) {                   // one 16-byte chunk (row j, channels 4 c4 ..) per work-item and turn
            const int j = e >> 2, c4 = e & 3;
            float4 k = make_float4(0.f, 0.f, 0.f, 0.f), v = k;
            if (j < p.J) { k = mi_ldg4(p.kf + base + (size_t)j * 16 + 4 * c4); v = mi_ldg4(p.vf + base + (size_t)j * 16 + 4 * c4); }
            *reinterpret_cast<float4*>(&ks[j * ATM_KP + 4 * c4]) = k;
            float* d = &vt[((j >> 2) * 16 + 4 * c4) * 4 + (j & 3)];      // vt[j / 4][c][j % 4]
            d[0] = v.x; d[4] = v.y; d[8] = v.z; d[12] = v.w;
        }
        __syncthreads();
        // NP token tiles side by side: two independent (scores -> maximum -> exponentials -> P V) chains per wave hide each other's matrix-pipe
        // and cross-lane latencies (two workgroups per CU = two waves per SIMD do not)
        constexpr int NP = TT >= 2 ? 2 : 1;
#pragma unroll
        for (int t0 = 0; t0 < TT; t0 += NP) {
            if (tok0 + 16 * t0 >= p.n) break;                                // wave-uniform
            float m[NP], l[NP];
            f32x4 acc[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) { m[u] = -INFINITY; l[u] = 0.f; acc[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            for (int jt = 0; jt < njt; ++jt) {
                const float4 ka = *reinterpret_cast<const float4*>(&ks[(16 * jt + lq) * ATM_KP + 4 * lg]);
                const float4 bb = *reinterpret_cast<const float4*>(&bias[16 * jt + 4 * lg]);
                const float4 va = *reinterpret_cast<const float4*>(&vt[((4 * jt + lg) * 16 + lq) * 4]);
                const float kv[4] = {ka.x, ka.y, ka.z, ka.w};
                f32x4 sc[NP];
#pragma unroll
                for (int u = 0; u < NP; ++u) sc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int u = 0; u < NP; ++u) {
                        const float qv = kb == 0 ? q4[t0 + u].x : (kb == 1 ? q4[t0 + u].y : (kb == 2 ? q4[t0 + u].z : q4[t0 + u].w));
                        sc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[kb], qv, sc[u], 0, 0, 0);
                    }
                float pr[NP][4];
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const float s0 = sc[u][0] + bb.x, s1 = sc[u][1] + bb.y, s2 = sc[u][2] + bb.z, s3 = sc[u][3] + bb.w;
                    float tm = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
                    tm = fmaxf(tm, __shfl_xor(tm, 16));
                    tm = fmaxf(tm, __shfl_xor(tm, 32));
                    const float mn = fmaxf(m[u], tm);
                    const float mu = (mn == -INFINITY) ? 0.f : mn;           // nothing live so far for this token: every exponent below is -inf -> 0
                    const float alpha = __builtin_amdgcn_exp2f(m[u] - mu);
                    m[u] = mn;
                    pr[u][0] = __builtin_amdgcn_exp2f(s0 - mu); pr[u][1] = __builtin_amdgcn_exp2f(s1 - mu);
                    pr[u][2] = __builtin_amdgcn_exp2f(s2 - mu); pr[u][3] = __builtin_amdgcn_exp2f(s3 - mu);
                    l[u] = fmaf(l[u], alpha, (pr[u][0] + pr[u][1]) + (pr[u][2] + pr[u][3]));
                    acc[u][0] *= alpha; acc[u][1] *= alpha; acc[u][2] *= alpha; acc[u][3] *= alpha;
                }
                const float vv[4] = {va.x, va.y, va.z, va.w};
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int u = 0; u < NP; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[r], pr[u][r], acc[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int t = t0 + u;
                float lt = l[u];
                lt += __shfl_xor(lt, 16);
                lt += __shfl_xor(lt, 32);
                const float inv = 1.0f / lt;
                const int i = tok0 + 16 * t + lq;
                const float4 oh4 = make_float4(acc[u][0] * inv, acc[u][1] * inv, acc[u][2] * inv, acc[u][3] * inv);     // out^T rows c' = 4 lg + r of token i
                otot[t][0] += oh4.x; otot[t][1] += oh4.y; otot[t][2] += oh4.z; otot[t][3] += oh4.w;
                if (i < p.n) {
                    if (p.oh) mi_stg4(p.oh + (((size_t)b * p.n + i) * p.H + h) * 16 + 4 * lg, oh4);
                    if (lg == 0) p.lse[((size_t)b * p.n + i) * p.H + h] = m[u] + log2f(lt);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const int i = tok0 + 16 * t + lq;
        if (i < p.n) mi_stg4(p.out + ((size_t)b * p.n + i) * 16 + 4 * lg, make_float4(otot[t][0], otot[t][1], otot[t][2], otot[t][3]));
    }
}

template <int TT, int NW>
int launch_folded_fwd_mfma(const mi_folded_attn_params& p, hipStream_t st) {
    const dim3 grid((p.n + 16 * NW * TT - 1) / (16 * NW * TT), p.B);
    if (p.J <= 16 * 17) hipLaunchKernelGGL(HIP_KERNEL_NAME(folded_attn_fwd_mfma_kernel<TT, 17, NW>), grid, dim3(64 * NW), 0, st, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(folded_attn_fwd_mfma_kernel<TT, 24, NW>), grid, dim3(64 * NW), 0, st, p);
    return mi_check_launch("folded_attn_fwd_mfma_kernel");
}


// ---- the token-side backward on the matrix cores (C = 16, per-head outputs saved by the forward): per tile of 16 context rows three products of the
// forward's shapes -- S^T = kf q^T (probabilities p = exp2(S^T - lse)), dP^T = vf dO^T, and dq^T[c][i] += sum_j kf[j][c] dS^T[j][i] with
// dS = p (dP - D), D_i = dO_i . O_hi -- the four dS registers of a lane being the B operands of the third as they lie.
template <int TT, int NJT_MAX, int NW>
__global__ __launch_bounds__(64 * NW) void folded_attn_dq_mfma_kernel(mi_folded_attn_params p) {
    constexpr int JP = 16 * NJT_MAX, NT = 64 * NW;
    __shared__ __attribute__((aligned(16))) float ks[JP * ATM_KP], vs[JP * ATM_KP], kt[JP * 16], bias[JP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int b = blockIdx.y, tok0 = (blockIdx.x * NW + wave) * (16 * TT);
    const int njt = (p.J + 15) >> 4;
    for (int j = tid; j < 16 * njt; j += NT) bias[j] = (j < p.J && (p.mask == nullptr || p.mask[(size_t)b * p.J + j])) ? 0.f : -INFINITY;
    float4 q4[TT], g4[TT];
    f32x4 dqa[TT];
    size_t rowi[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const int i = tok0 + 16 * t + lq;
        rowi[t] = (size_t)b * p.n + (i < p.n ? i : p.n - 1);
        const float4 u = mi_ldg4(p.q + rowi[t] * 16 + 4 * lg);
        q4[t] = make_float4(u.x * AT_LOG2E, u.y * AT_LOG2E, u.z * AT_LOG2E, u.w * AT_LOG2E);
        g4[t] = mi_ldg4(p.dout + rowi[t] * 16 + 4 * lg);
        dqa[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int h = 0; h < p.H; ++h) {
        __syncthreads();
        const size_t base = ((size_t)b * p.H + h) * p.J * 16;
        for (int e = tid; e < 16 * njt * 4; e += NT) {
            const int j = e >> 2, c4 = e & 3;
            float4 k = make_float4(0.f, 0.f, 0.f, 0.f), v = k;
            if (j < p.J) { k = mi_ldg4(p.kf + base + (size_t)j * 16 + 4 * c4); v = mi_ldg4(p.vf + base + (size_t)j * 16 + 4 * c4); }
            *reinterpret_cast<float4*>(&ks[j * ATM_KP + 4 * c4]) = k;
            *reinterpret_cast<float4*>(&vs[j * ATM_KP + 4 * c4]) = v;
            float* d = &kt[((j >> 2) * 16 + 4 * c4) * 4 + (j & 3)];      // kt[j / 4][c][j % 4]
            d[0] = k.x; d[4] = k.y; d[8] = k.z; d[12] = k.w;
        }
        __syncthreads();
        constexpr int NP = TT >= 2 ? 2 : 1;
#pragma unroll
        for (int t0 = 0; t0 < TT; t0 += NP) {
            if (tok0 + 16 * t0 >= p.n) break;                                // wave-uniform
            float lse[NP], D[NP];
            f32x4 acc[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                lse[u] = p.lse[rowi[t0 + u] * p.H + h];
                const float4 o = mi_ldg4(p.oh + (rowi[t0 + u] * p.H + h) * 16 + 4 * lg);
                const float4 g = g4[t0 + u];
                float d = fmaf(g.x, o.x, fmaf(g.y, o.y, fmaf(g.z, o.z, g.w * o.w)));
                d += __shfl_xor(d, 16);
                d += __shfl_xor(d, 32);
                D[u] = d;
                acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            for (int jt = 0; jt < njt; ++jt) {
                const float4 ka = *reinterpret_cast<const float4*>(&ks[(16 * jt + lq) * ATM_KP + 4 * lg]);
                const float4 va = *reinterpret_cast<const float4*>(&vs[(16 * jt + lq) * ATM_KP + 4 * lg]);
                const float4 bb = *reinterpret_cast<const float4*>(&bias[16 * jt + 4 * lg]);
                const float4 ta = *reinterpret_cast<const float4*>(&kt[((4 * jt + lg) * 16 + lq) * 4]);
                const float kv[4] = {ka.x, ka.y, ka.z, ka.w}, vv[4] = {va.x, va.y, va.z, va.w}, tv[4] = {ta.x, ta.y, ta.z, ta.w};
                f32x4 sc[NP], dp[NP];
#pragma unroll
                for (int u = 0; u < NP; ++u) { sc[u] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int u = 0; u < NP; ++u) {
                        const float4 qq = q4[t0 + u], gg = g4[t0 + u];
                        const float qv = kb == 0 ? qq.x : (kb == 1 ? qq.y : (kb == 2 ? qq.z : qq.w));
                        const float gv = kb == 0 ? gg.x : (kb == 1 ? gg.y : (kb == 2 ? gg.z : gg.w));
                        sc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[kb], qv, sc[u], 0, 0, 0);
                        dp[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[kb], gv, dp[u], 0, 0, 0);
                    }
                float ds[NP][4];
                const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int u = 0; u < NP; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pj = __builtin_amdgcn_exp2f(sc[u][r] + bv[r] - lse[u]);      // masked / padded rows: exp2(-inf) = 0
                        ds[u][r] = pj * (dp[u][r] - D[u]);
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int u = 0; u < NP; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(tv[r], ds[u][r], acc[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int t = t0 + u, i = tok0 + 16 * t + lq;
                dqa[t][0] += acc[u][0]; dqa[t][1] += acc[u][1]; dqa[t][2] += acc[u][2]; dqa[t][3] += acc[u][3];
                if (i < p.n && lg == 0) p.dsum[rowi[t] * p.H + h] = D[u];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const int i = tok0 + 16 * t + lq;
        if (i < p.n) mi_stg4(p.dq + rowi[t] * 16 + 4 * lg, make_float4(dqa[t][0], dqa[t][1], dqa[t][2], dqa[t][3]));
    }
}

template <int TT, int NW>
int launch_folded_dq_mfma(const mi_folded_attn_params& p, hipStream_t st) {
    const dim3 grid((p.n + 16 * NW * TT - 1) / (16 * NW * TT), p.B);
    if (p.J <= 16 * 17) hipLaunchKernelGGL(HIP_KERNEL_NAME(folded_attn_dq_mfma_kernel<TT, 17, NW>), grid, dim3(64 * NW), 0, st, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(folded_attn_dq_mfma_kernel<TT, 24, NW>), grid, dim3(64 * NW), 0, st, p);
    return mi_check_launch("folded_attn_dq_mfma_kernel");
}


// ---- the context-side backward on the matrix cores (C = 16): a workgroup per (token chunk, head, image) as above; its NW waves own the tiles of 16
// context rows (tile jt -> wave jt % NW: kf / vf fragments and the tile's dkf^T / dvf^T accumulators stay in registers) and walk the chunk's tokens
// together, 64 at a time through LDS.  Per (token tile, context tile):   S[i][j] = q_i . kf_j  and  dP[i][j] = dO_i . vf_j  (A = the token side: rows
// i0 + lq, channels 4 lg + kb), p = exp2(S - lse_i), dS = p (dP - D_i) for the lane's tokens i0 + 4 lg + r, then
//     dkf^T[c][j] += sum_i q[i][c] dS[i][j],   dvf^T[c][j] += sum_i dO[i][c] p[i][j]      (A = q / dO of token i0 + 4 lg + r, channel lq: the [i/4][16][4] copies)
// with the four dS / p registers as B operands as they lie.
template <int NJT_MAX, int NW>
__global__ __launch_bounds__(64 * NW) void folded_attn_dkv_mfma_kernel(mi_folded_attn_params p) {
    constexpr int NT = 64 * NW, TPW = (NJT_MAX + NW - 1) / NW;           // context tiles per wave
    __shared__ __attribute__((aligned(16))) float qs[64 * ATM_KP], gs[64 * ATM_KP], qt[64 * 16], gt[64 * 16], ls[64], dsm[64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int njt = (p.J + 15) >> 4;
    const size_t base = ((size_t)b * p.H + h) * p.J * 16;
    float4 kb4[TPW], vb4[TPW];
    float bj[TPW];
    f32x4 dk[TPW], dv[TPW];
#pragma unroll
    for (int w = 0; w < TPW; ++w) {
        const int j = 16 * (wave + NW * w) + lq;
        const bool in = j < p.J;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 k = in ? mi_ldg4(p.kf + base + (size_t)j * 16 + 4 * lg) : z;
        kb4[w] = make_float4(k.x * AT_LOG2E, k.y * AT_LOG2E, k.z * AT_LOG2E, k.w * AT_LOG2E);
        vb4[w] = in ? mi_ldg4(p.vf + base + (size_t)j * 16 + 4 * lg) : z;
        bj[w] = (in && (p.mask == nullptr || p.mask[(size_t)b * p.J + j])) ? 0.f : -INFINITY;
        dk[w] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[w] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int per = (p.n + gridDim.x - 1) / gridDim.x;
    const int i0 = chunk * per, i1 = (i0 + per < p.n) ? i0 + per : p.n;
    for (int t0 = i0; t0 < i1; t0 += 64) {
        __syncthreads();
        for (int e = tid; e < 64 * 4; e += NT) {                            // one 16-byte chunk (token, channels 4 c4 ..) of q and of dO per turn
            const int i = e >> 2, c4 = e & 3;
            float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), gv = qv;
            if (t0 + i < i1) { qv = mi_ldg4(p.q + ((size_t)b * p.n + t0 + i) * 16 + 4 * c4); gv = mi_ldg4(p.dout + ((size_t)b * p.n + t0 + i) * 16 + 4 * c4); }
            *reinterpret_cast<float4*>(&qs[i * ATM_KP + 4 * c4]) = qv;
            *reinterpret_cast<float4*>(&gs[i * ATM_KP + 4 * c4]) = gv;
            float* dq_ = &qt[((i >> 2) * 16 + 4 * c4) * 4 + (i & 3)];
            dq_[0] = qv.x; dq_[4] = qv.y; dq_[8] = qv.z; dq_[12] = qv.w;
            float* dg_ = &gt[((i >> 2) * 16 + 4 * c4) * 4 + (i & 3)];
            dg_[0] = gv.x; dg_[4] = gv.y; dg_[8] = gv.z; dg_[12] = gv.w;
        }
        for (int i = tid; i < 64; i += NT) {
            const bool in = t0 + i < i1;
            ls[i] = in ? p.lse[((size_t)b * p.n + t0 + i) * p.H + h] : INFINITY;      // a token beyond the chunk: p = exp2(S - inf) = 0
            dsm[i] = in ? p.dsum[((size_t)b * p.n + t0 + i) * p.H + h] : 0.f;
        }
        __syncthreads();
        const int nit = (i1 - t0 + 15) >> 4 < 4 ? (i1 - t0 + 15) >> 4 : 4;
        for (int it = 0; it < nit; ++it) {
            const float4 qa = *reinterpret_cast<const float4*>(&qs[(16 * it + lq) * ATM_KP + 4 * lg]);
            const float4 ga = *reinterpret_cast<const float4*>(&gs[(16 * it + lq) * ATM_KP + 4 * lg]);
            const float4 qta = *reinterpret_cast<const float4*>(&qt[((4 * it + lg) * 16 + lq) * 4]);
            const float4 gta = *reinterpret_cast<const float4*>(&gt[((4 * it + lg) * 16 + lq) * 4]);
            const float4 l4 = *reinterpret_cast<const float4*>(&ls[16 * it + 4 * lg]);
            const float4 d4 = *reinterpret_cast<const float4*>(&dsm[16 * it + 4 * lg]);
            const float qv[4] = {qa.x, qa.y, qa.z, qa.w}, gv[4] = {ga.x, ga.y, ga.z, ga.w}, qtv[4] = {qta.x, qta.y, qta.z, qta.w}, gtv[4] = {gta.x, gta.y, gta.z, gta.w};
            const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int w = 0; w < TPW; ++w) {
                if (wave + NW * w >= njt) break;                            // wave-uniform
                const float kv[4] = {kb4[w].x, kb4[w].y, kb4[w].z, kb4[w].w}, vv[4] = {vb4[w].x, vb4[w].y, vb4[w].z, vb4[w].w};
                f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[kb], kv[kb], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[kb], vv[kb], dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pj = __builtin_amdgcn_exp2f(sc[r] + bj[w] - lv[r]);
                    const float ds = pj * (dp[r] - dvv[r]);
                    dk[w] = __builtin_amdgcn_mfma_f32_16x16x4f32(qtv[r], ds, dk[w], 0, 0, 0);
                    dv[w] = __builtin_amdgcn_mfma_f32_16x16x4f32(gtv[r], pj, dv[w], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int w = 0; w < TPW; ++w) {
        const int j = 16 * (wave + NW * w) + lq;                             // D rows c = 4 lg + r of context row j
        if (j < p.J) {
            const size_t o = ((((size_t)chunk * gridDim.z + b) * p.H + h) * p.J + j) * 16 + 4 * lg;
            mi_stg4(p.dkf + o, make_float4(dk[w][0], dk[w][1], dk[w][2], dk[w][3]));
            mi_stg4(p.dvf + o, make_float4(dv[w][0], dv[w][1], dv[w][2], dv[w][3]));
        }
    }
}

int launch_folded_dkv_mfma(const mi_folded_attn_params& p, hipStream_t st) {
    const dim3 grid(p.nchunk, p.H, p.B);
    if (p.J <= 16 * 12) hipLaunchKernelGGL(HIP_KERNEL_NAME(folded_attn_dkv_mfma_kernel<12, 4>), grid, dim3(256), 0, st, p);
    else if (p.J <= 16 * 18) hipLaunchKernelGGL(HIP_KERNEL_NAME(folded_attn_dkv_mfma_kernel<18, 6>), grid, dim3(384), 0, st, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(folded_attn_dkv_mfma_kernel<24, 8>), grid, dim3(512), 0, st, p);
    return mi_check_launch("folded_attn_dkv_mfma_kernel");
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
    const bool valu_fwd = getenv("MI_FOLDED_ATTN_VALU") != nullptr;    // (read per call: the GPU test flips it in-process)          // A/B knob: the fp32 VALU kernels at C = 16 too
    if (which == 0 && q->C == 16 && q->J <= 16 * 24 && !valu_fwd) {
        // 256 tokens per workgroup where that still leaves two workgroups per CU: eight waves of two token tiles share one staged (kf, vf) -- four
        // waves per SIMD instead of two; smaller problems: four waves of two / one tile(s)
        const long long wg256 = (long long)((q->n + 255) / 256) * q->B;
        if (wg256 >= 512) return launch_folded_fwd_mfma<2, 8>(*q, st);        // (four waves of four tiles: 265-270 against 235-240 us at the SR shape)
        if (2 * wg256 >= 512) return launch_folded_fwd_mfma<2, 4>(*q, st);
        return launch_folded_fwd_mfma<1, 4>(*q, st);
    }
    if (which == 1 && q->C == 16 && q->J <= 16 * 24 && q->oh && !valu_fwd) {
        const long long wg256 = (long long)((q->n + 255) / 256) * q->B;
        if (wg256 >= 512) return launch_folded_dq_mfma<2, 8>(*q, st);
        if (2 * wg256 >= 512) return launch_folded_dq_mfma<2, 4>(*q, st);
        return launch_folded_dq_mfma<1, 4>(*q, st);
    }
    if (which == 2 && q->C == 16 && q->J <= 16 * 24 && !valu_fwd) return launch_folded_dkv_mfma(*q, st);
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
