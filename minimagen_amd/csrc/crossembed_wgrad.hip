// Weight / bias gradients of CrossEmbedLayer (layers.py:254-305: parallel k = 3, 7, 15 convs of the image, stride 1; training path, SURVEY
// 8(f) rank 3).  All members share the input window, so ONE correlation over the largest kernel's K x K taps serves every member:
//     G[co][ci][ky][kx] = sum_{b,y,x} dY[b][co][y][x] * X[b][ci][y + ky - K/2][x + kx - K/2]          (co over ALL members' channels)
// and a smaller member's gradient is the centre k x k window of its channels' G.  As a split-K GEMM on v_mfma_f32_16x16x4_f32:
// M = the (<= 16) output channels -- the k = 3 / 7 members ride in rows the MFMA would pad anyway --, N = (ci, kx) flattened (Cin * K <= 96:
// up to six 16-column blocks), K = pixels, one accumulator tile per (ky, N block); the four waves of a workgroup split the vertical taps.
// A workgroup walks 8 x 32 pixel tiles (X window with halo and the dY tile in LDS), writes its partial G once; a second kernel adds the
// partials in a fixed order (deterministic) straight into the members' [cout][Cin][k][k] gradients.
// PAIR (<= 8 output channels in all, the reference's dim_scales (4, 2, 2)): the MFMA's rows 8 .. 15 would be padding, so they carry the SAME
// channels one vertical tap further: with the X row rho as B operand, rows 0 .. 7 take dY row rho - ky (tap ky = 2 jp) and rows 8 .. 15 dY row
// rho - ky - 1 (tap ky + 1) -- both of the tile's own dY rows, zero outside them.  A tile then costs ceil(K / 2) x (TH + 1) x N blocks matrix
// instructions per four pixels of a row instead of K x TH (72 against 120 at K = 15); each wave keeps two tap pairs (half the accumulators).
#include "common.hip.h"

namespace {

constexpr int CW_TH = 8, CW_TW = 32, CW_KMAX = 15, CW_MAXC = 6, CW_NB = 6;
constexpr int CW_ROWS = CW_TH + CW_KMAX - 1, CW_XP = CW_TW + CW_KMAX - 1 + 1;       // 22 rows, pitch 47
constexpr int CW_DYP = CW_TH * CW_TW + 1;                                          // 257
constexpr int CW_DYR = 34, CW_DYC = 292;      // paired form: dY row pitch = -2 and channel pitch = 4 (mod 32 banks): the 32 lanes of a pass hit 32 banks
constexpr int CW_KY_PER_WAVE = 4;

struct CeWgradArgs {
    const float* x; const float* dy; float* partial; float* partial_db;
    int B, Cin, Ctot, H, W, K, nb, tiles_x, tiles_y;
};

template <bool PAIR>
__global__ __launch_bounds__(256) void crossembed_wgrad_partial_kernel(CeWgradArgs p) {
    __shared__ float x_s[CW_MAXC * CW_ROWS * CW_XP];              // 24.8 KB
    __shared__ float dy_s[PAIR ? 8 * CW_DYC : 16 * CW_DYP];        // 16.4 KB (PAIR: 9.3 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, kk = lane >> 4;
    const int K = p.K, halo = K >> 1;
    const long long HW = (long long)p.H * p.W;
    const mi_gptr<const float> gx = mi_global(p.x);
    const mi_gptr<const float> gdy = mi_global(p.dy);
    int boff[CW_NB];                                              // LDS offset of this lane's (ci, kx) column in each N block
#pragma unroll
    for (int nb = 0; nb < CW_NB; ++nb) {
        const int n = nb * 16 + lc;
        const bool ok = nb < p.nb && n < p.Cin * K;
        boff[nb] = ok ? (n / K) * CW_ROWS * CW_XP + (n % K) : 0;
    }
    f32x4 acc[CW_KY_PER_WAVE][CW_NB];
#pragma unroll
    for (int a = 0; a < CW_KY_PER_WAVE; ++a)
#pragma unroll
        for (int nb = 0; nb < CW_NB; ++nb) acc[a][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float sdb[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) sdb[c] = 0.f;
    const int ky0 = wave * CW_KY_PER_WAVE;

    const int ntiles = p.B * p.tiles_y * p.tiles_x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / (p.tiles_y * p.tiles_x);
        const int ty = (t / p.tiles_x) % p.tiles_y, tx = t % p.tiles_x;
        const int y0 = ty * CW_TH, x0 = tx * CW_TW;
        __syncthreads();
        {
            const int col = tid & 31, row = tid >> 5;
            const int y = y0 + row, x = x0 + col;
            const bool in = y < p.H && x < p.W;
            const long long base = (long long)b * p.Ctot * HW + (long long)y * p.W + x;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float v = (in && c < p.Ctot) ? gdy[base + c * HW] : 0.f;
                if (PAIR) { if (c < 8) dy_s[c * CW_DYC + row * CW_DYR + col] = v; }
                else dy_s[c * CW_DYP + row * CW_TW + col] = v;
                sdb[c] += v;
            }
        }
        const int rows = CW_TH + K - 1, cols = CW_TW + K - 1;
        for (int idx = tid; idx < p.Cin * rows * cols; idx += 256) {
            const int c = idx / (rows * cols), r = (idx / cols) % rows, cc = idx % cols;
            const int y = y0 + r - halo, x = x0 + cc - halo;
            const bool in = y >= 0 && y < p.H && x >= 0 && x < p.W;
            x_s[(c * CW_ROWS + r) * CW_XP + cc] = in ? gx[((long long)b * p.Cin + c) * HW + (long long)y * p.W + x] : 0.f;
        }
        __syncthreads();
        if (PAIR) {
            // this wave's tap pairs jp = 2 wave + a (a = 0, 1): X rows rho with a live dY row for either tap of a pair are 2 jp .. 2 jp + TH
            const int half = lc >> 3, cdy = (lc & 7) * CW_DYC;
            for (int rr = 0; rr < CW_TH + 3; ++rr) {
                const int rho = 4 * wave + rr;
                if (rho >= CW_TH + K - 1) break;
#pragma unroll 2
                for (int s = 0; s < CW_TW / 4; ++s) {
                    const int px = 4 * s + kk;
                    const float* row = x_s + rho * CW_XP + px;
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const int r0 = rr - 2 * a;                          // dY row of the even tap (wave-uniform); the odd tap's is r0 - 1
                        if (r0 >= 0 && r0 <= CW_TH && 4 * wave + 2 * a < K) {
                            const int r = r0 - half;
                            const float av = (r >= 0 && r < CW_TH) ? dy_s[cdy + r * CW_DYR + px] : 0.f;
#pragma unroll
                            for (int nb = 0; nb < CW_NB; ++nb)
                                if (nb < p.nb) acc[a][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, row[boff[nb]], acc[a][nb], 0, 0, 0);
                        }
                    }
                }
            }
        } else
        for (int r = 0; r < CW_TH; ++r) {
#pragma unroll 2
            for (int s = 0; s < CW_TW / 4; ++s) {
                const int px = 4 * s + kk;
                const float av = dy_s[lc * CW_DYP + r * CW_TW + px];
#pragma unroll
                for (int a = 0; a < CW_KY_PER_WAVE; ++a) {
                    if (ky0 + a < K) {                             // wave-uniform
                        const float* row = x_s + (r + ky0 + a) * CW_XP + px;
#pragma unroll
                        for (int nb = 0; nb < CW_NB; ++nb)
                            if (nb < p.nb) acc[a][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, row[boff[nb]], acc[a][nb], 0, 0, 0);
                    }
                }
            }
        }
    }
    // partial G of this workgroup: [ky][nb][co 16][n 16]
    float* out = p.partial + (long long)blockIdx.x * K * p.nb * 256;
    if (PAIR) {                                                    // D rows 0 .. 7: (co, tap 2 jp), rows 8 .. 15: (co, tap 2 jp + 1); rows co >= 8 of the partial are never read
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int nb = 0; nb < CW_NB; ++nb)
                if (nb < p.nb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = 4 * kk + i, ky = 4 * wave + 2 * a + (m >> 3);
                        if (ky < K) out[(ky * p.nb + nb) * 256 + (m & 7) * 16 + lc] = acc[a][nb][i];
                    }
    } else {
#pragma unroll
    for (int a = 0; a < CW_KY_PER_WAVE; ++a)
        if (ky0 + a < K)
#pragma unroll
            for (int nb = 0; nb < CW_NB; ++nb)
                if (nb < p.nb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) out[((ky0 + a) * p.nb + nb) * 256 + (4 * kk + i) * 16 + lc] = acc[a][nb][i];
    }
    __syncthreads();
    float* red = x_s;                                              // [256][17]
#pragma unroll
    for (int c = 0; c < 16; ++c) red[tid * 17 + c] = sdb[c];
    __syncthreads();
    if (tid < 16) {
        float s = 0.f;
        for (int k = 0; k < 256; ++k) s += red[k * 17 + tid];
        p.partial_db[(long long)blockIdx.x * 16 + tid] = s;
    }
}

struct CeReduceArgs {
    const float* partial; const float* partial_db;
    float* dw[3]; float* db[3];
    int n_kernels, ksize[3], cout[3], Cin, K, nb, nwg;
};

// 16 output elements x 16 slices of the partials per workgroup (as conv_wgrad_reduce_kernel)
__global__ __launch_bounds__(256) void crossembed_wgrad_reduce_kernel(CeReduceArgs p) {
    __shared__ float red[16][17];
    const int e16 = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + e16;
    // which member / element
    int member = -1, cbase = 0, local = idx;
    for (int i = 0; i < p.n_kernels; ++i) {
        const int n = p.cout[i] * p.Cin * p.ksize[i] * p.ksize[i];
        if (member < 0) {
            if (local < n) member = i;
            else { local -= n; cbase += p.cout[i]; }
        }
    }
    float s = 0.f;
    float* dst = nullptr;
    if (member >= 0) {
        const int k = p.ksize[member], off = (p.K - k) >> 1;
        const int kx = local % k, ky = (local / k) % k, ci = (local / (k * k)) % p.Cin, co = local / (k * k * p.Cin);
        const int n = ci * p.K + kx + off;
        const long long e = ((long long)(ky + off) * p.nb + n / 16) * 256 + (cbase + co) * 16 + n % 16;
        const long long stride = (long long)p.K * p.nb * 256;
        float s0 = 0.f, s1 = 0.f;
        int w = slice;
        for (; w + 16 < p.nwg; w += 32) { s0 += p.partial[w * stride + e]; s1 += p.partial[(w + 16) * stride + e]; }
        if (w < p.nwg) s0 += p.partial[w * stride + e];
        s = s0 + s1;
        dst = p.dw[member] + local;
    } else {                                                       // bias gradients after the weights: local = the channel over all members
        int c = local, m = 0;
        while (m < p.n_kernels && c >= p.cout[m]) { c -= p.cout[m]; ++m; }
        if (m < p.n_kernels && p.db[m]) {
            for (int w = slice; w < p.nwg; w += 16) s += p.partial_db[(long long)w * 16 + local];
            dst = p.db[m] + c;
        }
    }
    red[slice][e16] = s;
    __syncthreads();
    if (slice == 0 && dst) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][e16];
        *dst = t;
    }
}

}  // namespace

extern "C" long long mi_crossembed_wgrad_workspace(int Cin, int kmax, int nwg) {
    if (Cin <= 0 || kmax <= 0 || nwg <= 0) return -1;
    const long long nb = (Cin * kmax + 15) / 16;
    return (long long)nwg * (kmax * nb * 256 + 16);
}

extern "C" int mi_crossembed_wgrad(const mi_crossembed_wgrad_params* q, void* stream) {
    if (!q || q->B <= 0 || q->Cin <= 0 || q->H <= 0 || q->W <= 0 || q->nwg <= 0 || q->n_kernels <= 0 || q->n_kernels > 3 || !q->x || !q->dy || !q->partial) {
        mi_set_error("mi_crossembed_wgrad: bad arguments");
        return MI_ERR_INVALID;
    }
    int K = 0, ctot = 0;
    for (int i = 0; i < q->n_kernels; ++i) {
        if (q->ksize[i] <= 0 || !(q->ksize[i] & 1) || q->cout[i] <= 0 || !q->dw[i]) { mi_set_error("mi_crossembed_wgrad: kernel sizes must be odd, every member needs dw"); return MI_ERR_INVALID; }
        K = q->ksize[i] > K ? q->ksize[i] : K;
        ctot += q->cout[i];
    }
    if (K > CW_KMAX || ctot > 16 || q->Cin > CW_MAXC || (q->Cin * K + 15) / 16 > CW_NB) {
        mi_set_error("mi_crossembed_wgrad: supports kernel sizes <= %d, <= 16 output channels, Cin * kmax <= %d (got k %d, %d channels, Cin %d)", CW_KMAX, 16 * CW_NB, K, ctot, q->Cin);
        return MI_ERR_UNSUPPORTED;
    }
    CeWgradArgs p;
    p.x = q->x; p.dy = q->dy; p.B = q->B; p.Cin = q->Cin; p.Ctot = ctot; p.H = q->H; p.W = q->W; p.K = K; p.nb = (q->Cin * K + 15) / 16;
    p.tiles_x = (q->W + CW_TW - 1) / CW_TW; p.tiles_y = (q->H + CW_TH - 1) / CW_TH;
    p.partial = q->partial;
    p.partial_db = q->partial + (long long)q->nwg * K * p.nb * 256;
    if ((long long)p.B * p.tiles_x * p.tiles_y > 0x7fffffffLL) { mi_set_error("mi_crossembed_wgrad: problem too large"); return MI_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    if (ctot <= 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_wgrad_partial_kernel<true>), dim3(q->nwg), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_wgrad_partial_kernel<false>), dim3(q->nwg), dim3(256), 0, st, p);
    int rc = mi_check_launch("crossembed_wgrad_partial_kernel");
    if (rc) return rc;
    CeReduceArgs r;
    r.partial = q->partial; r.partial_db = p.partial_db; r.n_kernels = q->n_kernels; r.Cin = q->Cin; r.K = K; r.nb = p.nb; r.nwg = q->nwg;
    int n = 0;
    for (int i = 0; i < 3; ++i) {
        r.dw[i] = i < q->n_kernels ? q->dw[i] : nullptr; r.db[i] = i < q->n_kernels ? q->db[i] : nullptr;
        r.ksize[i] = i < q->n_kernels ? q->ksize[i] : 0; r.cout[i] = i < q->n_kernels ? q->cout[i] : 0;
        if (i < q->n_kernels) n += q->cout[i] * q->Cin * q->ksize[i] * q->ksize[i];
    }
    n += ctot;
    hipLaunchKernelGGL(crossembed_wgrad_reduce_kernel, dim3((n + 15) / 16), dim3(256), 0, st, r);
    return mi_check_launch("crossembed_wgrad_reduce_kernel");
}
