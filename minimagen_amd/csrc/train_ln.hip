// Training path: LayerNorm over the last dimension, forward and backward -- the reference's `LayerNorm` (layers.py:333-343)
// and the nn.LayerNorm members of the conditioning stack (Unet.py:107-112, layers.py:137) inside Imagen.forward -> loss.backward()
// (Imagen.py:512-573).  The big launches of the BASELINE U-Nets are the token LayerNorms around the bottleneck cross-attention: 131 072 rows of 16
// channels at B = 32 -- a work-item per row there (64 contiguous bytes per lane, a wave reads 4 KB), a wave per row for wider rows.
//   forward:   y = (x - mean) rstd gamma + beta,   stat[row] = (mean, rstd)          (biased variance, two passes in registers)
//   backward:  g = dy gamma,  dx = rstd (g - mean(g) - xh mean(g xh)),  xh = (x - mean) rstd
//              dgamma = sum_rows dy xh,  dbeta = sum_rows dy   -- per-workgroup partials added in a fixed order by a second launch (deterministic)
#include "common.hip.h"

namespace {

constexpr int LN_NT = 256, LN_MAXD = 1024;

// ---- a work-item per row (dim <= 32, dim % 4 == 0)
template <int D>
__global__ __launch_bounds__(LN_NT) void ln_fwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, float2* __restrict__ stat, int rows, float eps) {
    const int row = blockIdx.x * LN_NT + threadIdx.x;
    if (row >= rows) return;
    float v[D];
#pragma unroll
    for (int c = 0; c < D; c += 4) { const float4 t = *reinterpret_cast<const float4*>(x + (size_t)row * D + c); v[c] = t.x; v[c + 1] = t.y; v[c + 2] = t.z; v[c + 3] = t.w; }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) s += v[c];
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) { const float d = v[c] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(q / (float)D + eps);
    if (stat) stat[row] = make_float2(mean, rstd);
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        float4 o;
        o.x = (v[c] - mean) * rstd * gamma[c] + (beta ? beta[c] : 0.f);
        o.y = (v[c + 1] - mean) * rstd * gamma[c + 1] + (beta ? beta[c + 1] : 0.f);
        o.z = (v[c + 2] - mean) * rstd * gamma[c + 2] + (beta ? beta[c + 2] : 0.f);
        o.w = (v[c + 3] - mean) * rstd * gamma[c + 3] + (beta ? beta[c + 3] : 0.f);
        *reinterpret_cast<float4*>(y + (size_t)row * D + c) = o;
    }
}

// backward, a work-item per row; every work-item walks rows_per of them and keeps its dgamma / dbeta sums in registers; the workgroup's sums go to
// partial[wg][2][D] (lanes first, then the four waves through LDS)
template <int D>
__global__ __launch_bounds__(LN_NT) void ln_bwd_rows_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float2* __restrict__ stat, float* __restrict__ dx, float* __restrict__ partial, int rows, int rows_per) {
    __shared__ float red[LN_NT / 64][2 * D];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float dg[D], db[D], gm[D];
#pragma unroll
    for (int c = 0; c < D; ++c) { dg[c] = 0.f; db[c] = 0.f; gm[c] = gamma[c]; }
    const int base = blockIdx.x * LN_NT * rows_per;
    for (int k = 0; k < rows_per; ++k) {
        const int row = base + k * LN_NT + tid;
        if (row >= rows) break;
        float g[D], xh[D];
        const float2 st = stat[row];
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            const float4 a = *reinterpret_cast<const float4*>(dy + (size_t)row * D + c), b = *reinterpret_cast<const float4*>(x + (size_t)row * D + c);
            g[c] = a.x; g[c + 1] = a.y; g[c + 2] = a.z; g[c + 3] = a.w;
            xh[c] = (b.x - st.x) * st.y; xh[c + 1] = (b.y - st.x) * st.y; xh[c + 2] = (b.z - st.x) * st.y; xh[c + 3] = (b.w - st.x) * st.y;
        }
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            dg[c] = fmaf(g[c], xh[c], dg[c]);
            db[c] += g[c];
            g[c] *= gm[c];
            c1 += g[c];
            c2 = fmaf(g[c], xh[c], c2);
        }
        c1 /= (float)D; c2 /= (float)D;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            float4 o;
            o.x = st.y * (g[c] - c1 - xh[c] * c2); o.y = st.y * (g[c + 1] - c1 - xh[c + 1] * c2);
            o.z = st.y * (g[c + 2] - c1 - xh[c + 2] * c2); o.w = st.y * (g[c + 3] - c1 - xh[c + 3] * c2);
            *reinterpret_cast<float4*>(dx + (size_t)row * D + c) = o;
        }
    }
    if (!partial) return;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        float a = dg[c], b = db[c];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (lane == 0) { red[wave][c] = a; red[wave][D + c] = b; }
    }
    __syncthreads();
    if (tid < 2 * D) partial[(size_t)blockIdx.x * 2 * D + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// ---- a wave per row (any dim <= 1024): lane l holds columns l, l + 64, ...
__global__ __launch_bounds__(LN_NT) void ln_fwd_wave_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, float2* __restrict__ stat, int rows, int dim, float eps) {
    constexpr int PER = LN_MAXD / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * (LN_NT / 64) + wave;
    if (row >= rows) return;
    float v[PER];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int c = lane + 64 * k; v[k] = c < dim ? x[(size_t)row * dim + c] : 0.f; s += v[k]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) { const float d = (lane + 64 * k < dim) ? v[k] - mean : 0.f; q = fmaf(d, d, q); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)dim + eps);
    if (stat && lane == 0) stat[row] = make_float2(mean, rstd);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = lane + 64 * k;
        if (c < dim) y[(size_t)row * dim + c] = (v[k] - mean) * rstd * gamma[c] + (beta ? beta[c] : 0.f);
    }
}

__global__ __launch_bounds__(LN_NT) void ln_bwd_wave_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float2* __restrict__ stat, float* __restrict__ dx, float* __restrict__ partial, int rows, int dim, int rows_per) {
    constexpr int PER = LN_MAXD / 64, NW = LN_NT / 64;
    __shared__ float red[NW][2 * LN_MAXD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float dg[PER], db[PER], gm[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) { dg[k] = 0.f; db[k] = 0.f; gm[k] = (lane + 64 * k < dim) ? gamma[lane + 64 * k] : 0.f; }
    const int base = blockIdx.x * NW * rows_per;
    for (int r = 0; r < rows_per; ++r) {
        const int row = base + r * NW + wave;
        if (row >= rows) break;
        const float2 st = stat[row];
        float g[PER], xh[PER];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = lane + 64 * k;
            const bool ok = c < dim;
            const float d = ok ? dy[(size_t)row * dim + c] : 0.f;
            xh[k] = ok ? (x[(size_t)row * dim + c] - st.x) * st.y : 0.f;
            dg[k] = fmaf(d, xh[k], dg[k]);
            db[k] += d;
            g[k] = d * gm[k];
            c1 += g[k];
            c2 = fmaf(g[k], xh[k], c2);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { c1 += __shfl_xor(c1, o); c2 += __shfl_xor(c2, o); }
        c1 /= (float)dim; c2 /= (float)dim;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = lane + 64 * k;
            if (c < dim) dx[(size_t)row * dim + c] = st.y * (g[k] - c1 - xh[k] * c2);
        }
    }
    if (!partial) return;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = lane + 64 * k;
        if (c < dim) { red[wave][c] = dg[k]; red[wave][dim + c] = db[k]; }
    }
    __syncthreads();
    for (int i = tid; i < 2 * dim; i += LN_NT) partial[(size_t)blockIdx.x * 2 * dim + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// dgamma[c] = sum_wg partial[wg][0][c], dbeta[c] = sum_wg partial[wg][1][c]: a workgroup per output, strided partial sums then a fixed tree (the
// same order every run).  (A work-item per output walking the <= 512 partials one after the other took 82 us per launch: latency.)
__global__ __launch_bounds__(LN_NT) void ln_bwd_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dgamma, float* __restrict__ dbeta, int nwg, int dim) {
    __shared__ float red[LN_NT];
    const int i = blockIdx.x, tid = threadIdx.x;
    float a = 0.f;
    for (int w = tid; w < nwg; w += LN_NT) a += partial[(size_t)w * 2 * dim + i];
    red[tid] = a;
    __syncthreads();
    for (int o = LN_NT / 2; o >= 1; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        if (i < dim) dgamma[i] = red[0];
        else if (dbeta) dbeta[i - dim] = red[0];
    }
}

bool ln_rows_form(int dim) { return dim == 8 || dim == 16 || dim == 32; }

}  // namespace

extern "C" int mi_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stat, int rows, int dim, float eps, void* stream) {
    if (rows <= 0 || dim <= 0 || dim > LN_MAXD || !x || !gamma || !y) { mi_set_error("mi_layernorm_fwd: need rows > 0, 0 < dim <= %d (got %d, %d)", LN_MAXD, rows, dim); return MI_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    float2* s2 = reinterpret_cast<float2*>(stat);
    const dim3 g((rows + LN_NT - 1) / LN_NT);
    if (dim == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_fwd_rows_kernel<8>), g, dim3(LN_NT), 0, st, x, gamma, beta, y, s2, rows, eps);
    else if (dim == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_fwd_rows_kernel<16>), g, dim3(LN_NT), 0, st, x, gamma, beta, y, s2, rows, eps);
    else if (dim == 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_fwd_rows_kernel<32>), g, dim3(LN_NT), 0, st, x, gamma, beta, y, s2, rows, eps);
    else hipLaunchKernelGGL(ln_fwd_wave_kernel, dim3((rows + LN_NT / 64 - 1) / (LN_NT / 64)), dim3(LN_NT), 0, st, x, gamma, beta, y, s2, rows, dim, eps);
    return mi_check_launch("mi_layernorm_fwd");
}

/* workgroups of the backward launch = rows of `partial` ([nwg][2][dim] floats) */
extern "C" int mi_layernorm_bwd_nwg(int rows, int dim) {
    if (rows <= 0 || dim <= 0 || dim > LN_MAXD) return 0;
    const int per_wg = ln_rows_form(dim) ? LN_NT : LN_NT / 64;
    int nwg = (rows + per_wg - 1) / per_wg;
    return nwg > 512 ? 512 : nwg;                                      // (each workgroup then walks ceil(rows / (512 per_wg)) row groups)
}

extern "C" int mi_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* stat, float* dx, float* partial, float* dgamma, float* dbeta,
                                int rows, int dim, void* stream) {
    const int nwg = mi_layernorm_bwd_nwg(rows, dim);
    if (nwg == 0 || !dy || !x || !gamma || !stat || !dx || (dgamma && !partial)) { mi_set_error("mi_layernorm_bwd: bad arguments (rows %d, dim %d)", rows, dim); return MI_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    const float2* s2 = reinterpret_cast<const float2*>(stat);
    float* part = dgamma ? partial : nullptr;
    if (ln_rows_form(dim)) {
        const int rows_per = (rows + nwg * LN_NT - 1) / (nwg * LN_NT);
        if (dim == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_bwd_rows_kernel<8>), dim3(nwg), dim3(LN_NT), 0, st, dy, x, gamma, s2, dx, part, rows, rows_per);
        else if (dim == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_bwd_rows_kernel<16>), dim3(nwg), dim3(LN_NT), 0, st, dy, x, gamma, s2, dx, part, rows, rows_per);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_bwd_rows_kernel<32>), dim3(nwg), dim3(LN_NT), 0, st, dy, x, gamma, s2, dx, part, rows, rows_per);
    } else {
        const int nw = LN_NT / 64, rows_per = (rows + nwg * nw - 1) / (nwg * nw);
        hipLaunchKernelGGL(ln_bwd_wave_kernel, dim3(nwg), dim3(LN_NT), 0, st, dy, x, gamma, s2, dx, part, rows, dim, rows_per);
    }
    int rc = mi_check_launch("mi_layernorm_bwd");
    if (rc != MI_OK || !dgamma) return rc;
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(2 * dim), dim3(LN_NT), 0, st, partial, dgamma, dbeta, nwg, dim);
    return mi_check_launch("mi_layernorm_bwd (reduce)");
}
