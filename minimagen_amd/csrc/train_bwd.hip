// Training path: the pointwise half of Block's backward (layers.py:131-145: GroupNorm -> [scale/shift] -> SiLU),
// fused so that the normalised / activated tensors are never materialised:
//   channel_stats_kernel      per-(image, channel) sum and sum of squares of a tensor no HIP producer left statistics for
//   block_bwd_sums_kernel     U = sum g * xhat, V = sum g over a (image, channel) row chunk, g = dA * silu'(y2)          (reads x, dA)
//   block_bwd_apply_kernel    dx = rstd * (k * g - mean_g(dxhat) - xhat * mean_g(dxhat * xhat))                      (reads x, dA)
//   block_bwd_params_kernel   dgamma, dbeta, dscale, dshift from U, V (one wave per channel)
//   pack_conv3_kernel         a 3x3 conv weight (or its adjoint) -> row-paired fp16 hi|lo fragments + the direct-conv layout, on the device
// with y1 = gamma * xhat + beta, y2 = y1 * (scale + 1) + shift, a = silu(y2), k = gamma * (scale + 1).
#include "common.hip.h"

namespace {

// 1024 work-items per (image, channel) row, 16-byte loads where the row allows (round 6: 256 work-items with 4-byte loads left one
// four-wave workgroup per CU on the 256-row launches of the SR training step -- latency-bound at 20 us average)
__global__ __launch_bounds__(1024) void channel_stats_kernel(const float* x, double* stats, int HW) {
    __shared__ double red[32];
    const size_t row = blockIdx.x;
    const mi_gptr<const float> p = mi_global(x) + row * (size_t)HW;
    double S = 0.0, Q = 0.0;                      // fp64 from the first element: the kernel is memory-bound, and an fp32 sum of squares
    if ((HW & 3) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0) {      // loses the variance of a tensor with a large mean (common.hip.h)
        const float* xr = x + row * (size_t)HW;
        for (int i = threadIdx.x; i < (HW >> 2); i += 1024) {
            const float4 u = mi_ldg4(xr + 4 * (size_t)i);
            const double a = (double)u.x, b = (double)u.y, c = (double)u.z, d = (double)u.w;
            S += (a + b) + (c + d);
            Q = fma(a, a, Q); Q = fma(b, b, Q); Q = fma(c, c, Q); Q = fma(d, d, Q);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += 1024) {
            const double v = (double)p[i];
            S += v; Q = fma(v, v, Q);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { S += __shfl_xor(S, o); Q += __shfl_xor(Q, o); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[2 * wave] = S; red[2 * wave + 1] = Q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { s += red[2 * w]; q += red[2 * w + 1]; }
        stats[2 * row] = s;
        stats[2 * row + 1] = q;
    }
}

struct RowCoef { float mu, r, A, Bc, k; };

__device__ __forceinline__ RowCoef row_coef(const mi_block_bwd_params& p, int b, int c) {
    RowCoef o;
    mi_group_moments(p.x_stats, p.nt, p.C, p.groups, p.HW, p.eps, b, c, o.mu, o.r);
    const float ga = p.gamma[c], be = p.beta[c];
    float sc1 = 1.0f, sh = 0.0f;
    if (p.ss) { sc1 = p.ss[(size_t)b * p.ss_stride + p.ss_off + c] + 1.0f; sh = p.ss[(size_t)b * p.ss_stride + p.ss_off + p.C + c]; }
    o.k = ga * sc1;
    o.A = o.r * o.k;
    o.Bc = (be - o.mu * o.r * ga) * sc1 + sh;
    return o;
}

// grid (B*C, nchunk)
__global__ __launch_bounds__(256) void block_bwd_sums_kernel(mi_block_bwd_params p) {
    __shared__ double red[8];
    const int row = blockIdx.x, b = row / p.C, c = row % p.C;
    const RowCoef k = row_coef(p, b, c);
    const int per = (p.HW + gridDim.y - 1) / gridDim.y;
    const int i0 = blockIdx.y * per, i1 = (i0 + per < p.HW) ? i0 + per : p.HW;
    const mi_gptr<const float> x = mi_global(p.x) + (size_t)row * p.HW;
    const mi_gptr<const float> da = mi_global(p.da) + (size_t)row * p.HW;
    float u = 0.f, v = 0.f;
    double U = 0.0, V = 0.0;
    int n = 0;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const float xv = x[i];
        const float g = da[i] * mi_silu_grad(fmaf(xv, k.A, k.Bc));
        u = fmaf(g, (xv - k.mu) * k.r, u);
        v += g;
        if (++n == 64) { U += (double)u; V += (double)v; u = v = 0.f; n = 0; }
    }
    U += (double)u; V += (double)v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { U += __shfl_xor(U, o); V += __shfl_xor(V, o); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[2 * wave] = U; red[2 * wave + 1] = V; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = p.uv + ((size_t)row * gridDim.y + blockIdx.y) * 2;
        o[0] = (float)((red[0] + red[2]) + (red[4] + red[6]));
        o[1] = (float)((red[1] + red[3]) + (red[5] + red[7]));
    }
}

// grid (B*C, nchunk); p.nchunk = the chunk count the sums were written with
__global__ __launch_bounds__(256) void block_bwd_apply_kernel(mi_block_bwd_params p) {
    __shared__ float sM[2];
    const int row = blockIdx.x, b = row / p.C, c = row % p.C;
    const RowCoef k = row_coef(p, b, c);
    if (threadIdx.x < 64) {         // group means of dxhat and dxhat * xhat from the rows' sums
        const int cpg = p.C / p.groups, g = c / cpg, lane = threadIdx.x;
        double m1 = 0.0, m2 = 0.0;
        for (int i = lane; i < cpg * p.nchunk; i += 64) {
            const int cc = g * cpg + i / p.nchunk;
            float sc1 = 1.0f;
            if (p.ss) sc1 = p.ss[(size_t)b * p.ss_stride + p.ss_off + cc] + 1.0f;
            const float kc = p.gamma[cc] * sc1;
            const float* uv = p.uv + (((size_t)b * p.C + cc) * p.nchunk + i % p.nchunk) * 2;
            m2 += (double)kc * (double)uv[0];
            m1 += (double)kc * (double)uv[1];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
        const double n = (double)cpg * (double)p.HW;
        if (lane == 0) { sM[0] = (float)(m1 / n); sM[1] = (float)(m2 / n); }
    }
    __syncthreads();
    const float M1 = sM[0], M2 = sM[1];
    const int per = (p.HW + gridDim.y - 1) / gridDim.y;
    const int i0 = blockIdx.y * per, i1 = (i0 + per < p.HW) ? i0 + per : p.HW;
    const mi_gptr<const float> x = mi_global(p.x) + (size_t)row * p.HW;
    const mi_gptr<const float> da = mi_global(p.da) + (size_t)row * p.HW;
    float* dx = p.dx + (size_t)row * p.HW;
    double S = 0.0, Q = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const float xv = x[i];
        const float g = da[i] * mi_silu_grad(fmaf(xv, k.A, k.Bc));
        const float xh = (xv - k.mu) * k.r;
        const float d = k.r * (k.k * g - M1 - xh * M2);
        dx[i] = d;
        S += (double)d; Q = fma((double)d, (double)d, Q);
    }
    if (p.dx_stats) {               // (sum, sum of squares) of this chunk of dx: the statistics the NEXT backward's data-gradient conv scales by
        __shared__ double red[8];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { S += __shfl_xor(S, o); Q += __shfl_xor(Q, o); }
        const int wave = threadIdx.x >> 6;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { red[2 * wave] = S; red[2 * wave + 1] = Q; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double* o = p.dx_stats + ((size_t)row * gridDim.y + blockIdx.y) * 2;
            o[0] = (red[0] + red[2]) + (red[4] + red[6]);
            o[1] = (red[1] + red[3]) + (red[5] + red[7]);
        }
    }
}

// one wave per channel: lanes stride over the (image, chunk) pairs, fixed-order shuffle tree
__global__ __launch_bounds__(256) void block_bwd_params_kernel(mi_block_bwd_params p) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= p.C) return;
    const float ga = p.gamma[c], be = p.beta[c];
    double dg = 0.0, db = 0.0;
    for (int b = lane; b < p.B; b += 64) {
        double U = 0.0, V = 0.0;
        const float* uv = p.uv + ((size_t)b * p.C + c) * p.nchunk * 2;
        for (int j = 0; j < p.nchunk; ++j) { U += (double)uv[2 * j]; V += (double)uv[2 * j + 1]; }
        float sc1 = 1.0f;
        if (p.ss) {
            sc1 = p.ss[(size_t)b * p.ss_stride + p.ss_off + c] + 1.0f;
            p.dss[(size_t)b * 2 * p.C + c] = (float)((double)ga * U + (double)be * V);        // d scale
            p.dss[(size_t)b * 2 * p.C + p.C + c] = (float)V;                                 // d shift
        }
        dg += (double)sc1 * U;
        db += (double)sc1 * V;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { dg += __shfl_xor(dg, o); db += __shfl_xor(db, o); }
    if (lane == 0) { p.dgamma[c] = (float)dg; p.dbeta[c] = (float)db; }
}

constexpr int PK_PERM[4] = {0, 2, 1, 3};        // packing.RP_PERM

// fragments [ko][3][nj][64][8 hi | 8 lo] (packing.pack_conv_weight_rp, 3x3 form) and generic [Cin][3][3][cout_pad] of W or of its adjoint
__device__ __forceinline__ void pack_conv3_body(const float* w, int Cout, int Cin, int adjoint, int exp, _Float16* frag, float* generic, int cout_pad, int bx, int nbx) {
    // logical weight L[co][ci][ky][kx]: adjoint ? w[ci][co][2-ky][2-kx] (w stored [Cin_l = Cout_w ...]) : w[co][ci][ky][kx]
    const int LCo = adjoint ? Cin : Cout, LCi = adjoint ? Cout : Cin;      // logical output / input channels
    auto Lw = [&](int co, int ci, int ky, int kx) -> float {
        if (co >= LCo || ci >= LCi) return 0.f;
        return adjoint ? w[(((size_t)ci * Cin + co) * 3 + (2 - ky)) * 3 + (2 - kx)] : w[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx];
    };
    const int nj = (LCo + 7) / 8, ko = (LCi + 7) / 8;
    const int nfrag = ko * 3 * nj * 64 * 8;
    const int ngen = LCi * 9 * cout_pad;
    const float scale = ldexpf(1.0f, exp);
    for (int idx = bx * 256 + threadIdx.x; idx < nfrag + ngen; idx += nbx * 256) {
        if (idx < nfrag) {
            const int e = idx & 7, lane = (idx >> 3) & 63;
            int rest = idx >> 9;
            const int jt = rest % nj; rest /= nj;
            const int s = rest % 3, k = rest / 3;
            const int lq = lane & 15, lg = lane >> 4;
            const int ky = PK_PERM[lg] - (lq >> 3);
            float v = 0.f;
            if (ky >= 0 && ky <= 2) v = Lw(8 * jt + (lq & 7), 8 * k + e, ky, s) * scale;
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            _Float16* o = frag + ((size_t)(idx >> 3)) * 16;
            o[e] = hi;
            o[8 + e] = lo;
        } else {
            const int j = idx - nfrag;
            const int co = j % cout_pad, kx = (j / cout_pad) % 3, ky = (j / (cout_pad * 3)) % 3, ci = j / (cout_pad * 9);
            generic[j] = Lw(co, ci, ky, kx);
        }
    }
}

__global__ __launch_bounds__(256) void pack_conv3_kernel(const float* w, int Cout, int Cin, int adjoint, int exp, _Float16* frag, float* generic, int cout_pad) {
    pack_conv3_body(w, Cout, Cin, adjoint, exp, frag, generic, cout_pad, blockIdx.x, gridDim.x);
}

// grid (blocks, n): descriptor blockIdx.y
__global__ __launch_bounds__(256) void pack_conv3_multi_kernel(const mi_pack_conv3_desc* descs) {
    const mi_pack_conv3_desc d = descs[blockIdx.y];
    pack_conv3_body(d.w, d.Cout, d.Cin, d.adjoint, d.exp, (_Float16*)d.frag, d.generic, d.cout_pad, blockIdx.x, gridDim.x);
}

}  // namespace

extern "C" int mi_chan_stats_fwd(const float* x, double* stats, int rows, int HW, void* stream) {
    if (!x || !stats || rows <= 0 || HW <= 0) { mi_set_error("mi_chan_stats_fwd: bad arguments"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(channel_stats_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, x, stats, HW);
    return mi_check_launch("channel_stats_kernel");
}

extern "C" int mi_block_bwd(const mi_block_bwd_params* q, void* stream) {
    if (!q || q->B <= 0 || q->C <= 0 || q->HW <= 0 || q->groups <= 0 || (q->C % q->groups) || q->nchunk <= 0 || q->nchunk > 65535 || !q->x || !q->da || !q->x_stats || q->nt <= 0
        || !q->gamma || !q->beta || !q->uv || !q->dx || !q->dgamma || !q->dbeta || (q->ss && !q->dss)) {
        mi_set_error("mi_block_bwd: bad arguments");
        return MI_ERR_INVALID;
    }
    const mi_block_bwd_params p = *q;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(p.B * p.C, p.nchunk);
    hipLaunchKernelGGL(block_bwd_sums_kernel, grid, dim3(256), 0, st, p);
    int rc = mi_check_launch("block_bwd_sums_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(block_bwd_apply_kernel, grid, dim3(256), 0, st, p);
    rc = mi_check_launch("block_bwd_apply_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(block_bwd_params_kernel, dim3((p.C + 3) / 4), dim3(256), 0, st, p);
    return mi_check_launch("block_bwd_params_kernel");
}

extern "C" long long mi_pack_conv3_floats(int Cout, int Cin, int adjoint, int cout_pad, int which) {
    const int LCo = adjoint ? Cin : Cout, LCi = adjoint ? Cout : Cin;
    if (which == 0) return (long long)((LCi + 7) / 8) * 3 * ((LCo + 7) / 8) * 64 * 16;      // fp16 elements of the fragments
    return (long long)LCi * 9 * cout_pad;                                                   // fp32 elements of the direct-conv layout
}

extern "C" int mi_pack_conv3(const float* w, int Cout, int Cin, int adjoint, int exp, void* frag, float* generic, int cout_pad, void* stream) {
    const int LCo = adjoint ? Cin : Cout;
    if (!w || !frag || !generic || Cout <= 0 || Cin <= 0 || cout_pad < LCo) { mi_set_error("mi_pack_conv3: bad arguments"); return MI_ERR_INVALID; }
    const long long n = mi_pack_conv3_floats(Cout, Cin, adjoint, cout_pad, 0) / 2 + mi_pack_conv3_floats(Cout, Cin, adjoint, cout_pad, 1);
    const int blocks = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_conv3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, adjoint, exp, (_Float16*)frag, generic, cout_pad);
    return mi_check_launch("pack_conv3_kernel");
}

extern "C" int mi_pack_conv3_multi(const mi_pack_conv3_desc* descs, int n, int blocks, void* stream) {
    if (!descs || n <= 0 || n > 65535 || blocks <= 0) { mi_set_error("mi_pack_conv3_multi: bad arguments"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(pack_conv3_multi_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, descs);
    return mi_check_launch("pack_conv3_multi_kernel");
}
