// Weight / bias gradient of the 3x3 stride-1 convolutions (training path): the backward of `Block.project`
// (layers.py:126, 145) w.r.t. its parameters,
//     dW[co][ci][ky][kx] = sum_{b,y,x} dY[b][co][y][x] * A[b][ci][y+ky-1][x+kx-1],      db[co] = sum_{b,y,x} dY[b][co][y][x]
// as a split-K GEMM on v_mfma_f32_16x16x4_f32 (exact fp32 products): M = 16 output channels, N = 16 input channels, K = pixels, one
// accumulator tile per tap.  A workgroup walks 8 x 32 pixel tiles (persistent over the batch), stages dY and the haloed A window in LDS
// pixel-major (pitch 17: the transposing write and the [4 px][16 ch] operand reads are both conflict-free), every wave takes two tile rows;
// the 9 x 16 x 16 partial of a workgroup goes to a workspace and a second kernel adds the partials in a fixed order (deterministic, no atomics).
// The data-gradient is the forward kernel itself on the transposed, flipped weights (minimagen_amd/train_ops.py).
#include "common.hip.h"

namespace {

constexpr int WG_TH = 8, WG_TW = 32, WG_PITCH = 17;
constexpr int WG_AW = WG_TW + 2, WG_AH = WG_TH + 2;
constexpr int WG_DY_FLOATS = WG_TH * WG_TW * WG_PITCH;            // 4352
constexpr int WG_A_FLOATS = WG_AH * WG_AW * WG_PITCH;             // 5780
constexpr int WG_BLK = 9 * 256;                                   // one (16 co x 16 ci) block of partials: [tap][co][ci]

struct WgradArgs {
    const float* a; const float* dy; float* partial; float* partial_db;
    int B, Cin, Cout, H, W, nciB, ncoB, tiles_x, tiles_y;
    // fused activation of the A operand (a_stats != NULL): a = SiLU(GroupNorm(x) * (scale + 1) + shift), computed while the tile is staged
    const double* a_stats; int a_nt; const float* gamma; const float* beta; int groups; float eps; const float* ss; int ss_stride, ss_off;
};

// PACK (Cin <= 8): the N dimension holds TWO taps x 8 input channels (lanes 0-7 of a column group: tap 2m, lanes 8-15: tap 2m + 1), so the nine taps
// take five MFMAs per 4-pixel step instead of nine; the partial block is then [m][co][half * 8 + ci]
template <bool PACK>
__global__ __launch_bounds__(256) void conv_wgrad_partial_kernel(WgradArgs p) {
    __shared__ float lds[WG_DY_FLOATS + WG_A_FLOATS];             // 40.5 KB; reused for the cross-wave reduction (4 x 2304 floats)
    __shared__ float aff[16][2];                                  // y2 = x * aff[c][0] + aff[c][1] of the current image's 16 input channels
    int aff_b = -1;
    float* dy_s = lds;
    float* a_s = lds + WG_DY_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 15, kk = lane >> 4;
    const int coB = blockIdx.y / p.nciB, ciB = blockIdx.y % p.nciB;
    const int co0 = coB * 16, ci0 = ciB * 16;
    const bool do_db = (ciB == 0) && p.partial_db != nullptr;
    const mi_gptr<const float> ga = mi_global(p.a);
    const mi_gptr<const float> gdy = mi_global(p.dy);
    const long long HW = (long long)p.H * p.W;

    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float sdb[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) sdb[c] = 0.f;

    int poff[5];                                                  // PACK: window offset (ky * AW + kx) of this lane's tap in each of the five MFMAs
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const int tap = 2 * m + (lc >> 3);
        poff[m] = tap < 9 ? (tap / 3) * WG_AW + tap % 3 : 0;
    }
    const int ntiles = p.B * p.tiles_y * p.tiles_x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / (p.tiles_y * p.tiles_x);
        const int ty = (t / p.tiles_x) % p.tiles_y, tx = t % p.tiles_x;
        const int y0 = ty * WG_TH, x0 = tx * WG_TW;
        __syncthreads();                                          // the previous tile's operand reads are done
        if (p.a_stats && b != aff_b) {                            // uniform: a new image -> its channels' affine (one wave per 4 channels)
            aff_b = b;
            for (int j = 0; j < 4; ++j) {
                const int c = ci0 + wave * 4 + j;
                float A = 0.f, Bc = 0.f;
                if (c < p.Cin) {
                    float mu, r;
                    mi_group_moments(p.a_stats, p.a_nt, p.Cin, p.groups, p.H * p.W, p.eps, b, c, mu, r);
                    float sc1 = 1.0f, sh = 0.0f;
                    if (p.ss) { sc1 = p.ss[(size_t)b * p.ss_stride + p.ss_off + c] + 1.0f; sh = p.ss[(size_t)b * p.ss_stride + p.ss_off + p.Cin + c]; }
                    A = r * p.gamma[c] * sc1;
                    Bc = (p.beta[c] - mu * r * p.gamma[c]) * sc1 + sh;
                }
                if (lane == 0) { aff[wave * 4 + j][0] = A; aff[wave * 4 + j][1] = Bc; }
            }
            __syncthreads();
        }
        {
            const int col = tid & 31, row = tid >> 5;
            const int y = y0 + row, x = x0 + col;
            const bool in = y < p.H && x < p.W;
            const long long base = ((long long)b * p.Cout + co0) * HW + (long long)y * p.W + x;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float v = (in && co0 + c < p.Cout) ? gdy[base + c * HW] : 0.f;
                dy_s[(row * WG_TW + col) * WG_PITCH + c] = v;
                sdb[c] += v;
            }
        }
        for (int idx = tid; idx < 16 * WG_AH * WG_AW; idx += 256) {
            const int c = idx / (WG_AH * WG_AW), px = idx % (WG_AH * WG_AW);
            const int y = y0 + px / WG_AW - 1, x = x0 + px % WG_AW - 1;
            const bool in = y >= 0 && y < p.H && x >= 0 && x < p.W && ci0 + c < p.Cin;
            float v = in ? ga[((long long)b * p.Cin + ci0 + c) * HW + (long long)y * p.W + x] : 0.f;
            if (p.a_stats && in) v = mi_silu(fmaf(v, aff[c][0], aff[c][1]));          // zero padding stays zero: the conv pads the ACTIVATED tensor
            a_s[px * WG_PITCH + c] = v;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = wave * 2 + rr;
#pragma unroll 2
            for (int s = 0; s < WG_TW / 4; ++s) {
                const int px = 4 * s + kk;
                const float av = dy_s[(r * WG_TW + px) * WG_PITCH + lc];
                if constexpr (PACK) {
#pragma unroll
                    for (int m = 0; m < 5; ++m) {
                        const float bv = a_s[(r * WG_AW + px + poff[m]) * WG_PITCH + (lc & 7)];
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[m], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float bv = a_s[((r + ky) * WG_AW + px + kx) * WG_PITCH + lc];
                            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[ky * 3 + kx], 0, 0, 0);
                        }
                }
            }
        }
    }
    // ---- workgroup partial: sum over the four waves, then one [tap][co][ci] block to the workspace
    __syncthreads();
    float* red = lds;                                             // [wave][tap][co][ci]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(wave * 9 + t) * 256 + (4 * kk + i) * 16 + lc] = acc[t][i];
    __syncthreads();
    float* out = p.partial + ((long long)blockIdx.x * gridDim.y + blockIdx.y) * WG_BLK;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int e = tid + 256 * j;
        out[e] = (red[e] + red[WG_BLK + e]) + (red[2 * WG_BLK + e] + red[3 * WG_BLK + e]);
    }
    if (do_db) {                                                  // uniform per workgroup
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) red[tid * WG_PITCH + c] = sdb[c];
        __syncthreads();
        if (tid < 16) {
            float s = 0.f;
            for (int k = 0; k < 256; ++k) s += red[k * WG_PITCH + tid];
            p.partial_db[((long long)blockIdx.x * p.ncoB + coB) * 16 + tid] = s;
        }
    }
}

// 16 output elements x 16 slices of the partials per workgroup: every thread adds its slice in a fixed order, then a fixed-order LDS tree
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* partial, const float* partial_db, float* dw, float* db, int Cin, int Cout,
                                                                int nciB, int ncoB, int nwg, int pack) {
    __shared__ float red[16][17];
    const int e16 = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + e16;
    const int nw = Cout * Cin * 9;
    float s = 0.f;
    if (idx < nw) {
        const int tap = idx % 9, ci = (idx / 9) % Cin, co = idx / (9 * Cin);
        const long long blk = (long long)(co / 16) * nciB + ci / 16;
        const long long e = pack ? (tap >> 1) * 256 + (co % 16) * 16 + (tap & 1) * 8 + ci : tap * 256 + (co % 16) * 16 + ci % 16;
        const long long stride = (long long)ncoB * nciB * WG_BLK;
        const float* src = partial + blk * WG_BLK + e;
        float s0 = 0.f, s1 = 0.f;
        int w = slice;
        for (; w + 16 < nwg; w += 32) { s0 += src[w * stride]; s1 += src[(w + 16) * stride]; }
        if (w < nwg) s0 += src[w * stride];
        s = s0 + s1;
    } else if (db != nullptr && idx < nw + Cout) {
        const int co = idx - nw;
        for (int w = slice; w < nwg; w += 16) s += partial_db[((long long)w * ncoB + co / 16) * 16 + co % 16];
    }
    red[slice][e16] = s;
    __syncthreads();
    if (slice == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][e16];
        if (idx < nw) dw[idx] = t;
        else if (db != nullptr && idx < nw + Cout) db[idx - nw] = t;
    }
}

}  // namespace

extern "C" long long mi_conv_wgrad_workspace(int Cin, int Cout, int nwg) {
    if (Cin <= 0 || Cout <= 0 || nwg <= 0) return -1;
    const long long nciB = (Cin + 15) / 16, ncoB = (Cout + 15) / 16;
    return (long long)nwg * (nciB * ncoB * WG_BLK + ncoB * 16);
}

extern "C" int mi_conv_wgrad(const mi_conv_wgrad_params* q, void* stream) {
    if (!q || q->B <= 0 || q->Cin <= 0 || q->Cout <= 0 || q->H <= 0 || q->W <= 0 || q->nwg <= 0 || !q->a || !q->dy || !q->dw || !q->partial) {
        mi_set_error("mi_conv_wgrad: bad arguments");
        return MI_ERR_INVALID;
    }
    WgradArgs p;
    p.a = q->a; p.dy = q->dy; p.partial = q->partial;
    p.B = q->B; p.Cin = q->Cin; p.Cout = q->Cout; p.H = q->H; p.W = q->W;
    p.nciB = (q->Cin + 15) / 16; p.ncoB = (q->Cout + 15) / 16;
    p.tiles_x = (q->W + WG_TW - 1) / WG_TW; p.tiles_y = (q->H + WG_TH - 1) / WG_TH;
    p.a_stats = q->a_stats; p.a_nt = q->a_nt; p.gamma = q->gamma; p.beta = q->beta; p.groups = q->groups; p.eps = q->eps;
    p.ss = q->ss; p.ss_stride = q->ss_stride; p.ss_off = q->ss_off;
    if (p.a_stats && (!p.gamma || !p.beta || p.groups <= 0 || (p.Cin % p.groups) || p.a_nt <= 0)) { mi_set_error("mi_conv_wgrad: fused activation needs gamma / beta / groups dividing Cin / a_nt"); return MI_ERR_INVALID; }
    p.partial_db = q->db ? q->partial + (long long)q->nwg * p.nciB * p.ncoB * WG_BLK : nullptr;
    const long long ntiles = (long long)p.B * p.tiles_x * p.tiles_y;
    if (ntiles > 0x7fffffffLL || (long long)p.nciB * p.ncoB > 65535) { mi_set_error("mi_conv_wgrad: problem too large"); return MI_ERR_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    const int pack = q->Cin <= 8 ? 1 : 0;
    if (pack) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_partial_kernel<true>), dim3(q->nwg, p.nciB * p.ncoB), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_partial_kernel<false>), dim3(q->nwg, p.nciB * p.ncoB), dim3(256), 0, st, p);
    int rc = mi_check_launch("conv_wgrad_partial_kernel");
    if (rc) return rc;
    const int n = q->Cout * q->Cin * 9 + (q->db ? q->Cout : 0);
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((n + 15) / 16), dim3(256), 0, st, (const float*)q->partial, (const float*)p.partial_db, q->dw, q->db,
                       q->Cin, q->Cout, p.nciB, p.ncoB, q->nwg, pack);
    return mi_check_launch("conv_wgrad_reduce_kernel");
}
