// K3: CrossEmbedLayer (MinImagen layers.py:254-305; Unet.py:169-172,396-400): three parallel
// stride-1 convolutions (k = 3, 7, 15) over the raw input image (x, or cat(x, lowres_cond_img)
// for super-resolution U-Nets), concatenated on the channel axis.  One kernel: the 15x15 halo
// tile is staged once per input channel in LDS and all three kernels consume it (the 7x7 and
// 3x3 windows are sub-windows of the rows already in registers); the two inputs are read from
// their own pointers (the concat is never materialised); per-channel partial statistics for the
// first GroupNorm come out of the epilogue.
#include "common.hip.h"

namespace {

// ---- specialised: dim_scales (4,2,2), kernel sizes (3,7,15) -- every U-Net with dim=8 ----
template <int NT, int TW>
__global__ __launch_bounds__(NT) void crossembed_422_kernel(const mi_crossembed_params p, const float* __restrict__ w3g,
                                                            const float* __restrict__ w7g, const float* __restrict__ w15g) {
    // (w3g/w7g/w15g repeat p.w[] as const __restrict__ kernel arguments so the wave-uniform weight reads become s_load)
    constexpr int TXN = TW / 4, TH = NT / TXN, HALO = 7;
    constexpr int IH = TH + 2 * HALO, IW = TW + 2 * HALO, IWP = (IW + 3) & ~3;
    constexpr int COUT = 8;
    constexpr int STAGE = IH * IWP, RED = 2 * COUT * (NT + 1);
    constexpr int WIN4 = (TW + 16) / 4;                 // aligned float4 window [ox0-8, ox0+TW+8)
    constexpr int PER4 = (IH * WIN4 + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float smem[STAGE > RED ? STAGE : RED];

    const int tid = threadIdx.x;
    const int tiles_x = (p.W + TW - 1) / TW;
    const int tile = blockIdx.x;
    const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
    const int b = blockIdx.y;
    const int C0 = p.C0, C1 = p.in1 ? p.C1 : 0, Cin = C0 + C1;
    const int b0 = p.in0_batch_mod > 0 ? b % p.in0_batch_mod : b;
    const int b1 = p.in1_batch_mod > 0 ? b % p.in1_batch_mod : b;
    const int ty = tid / TXN, tx = tid % TXN;
    const bool vec = (p.W & 3) == 0;

    float acc[4][COUT];
#pragma unroll
    for (int px = 0; px < 4; ++px)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[px][co] = 0.0f;

    // split staging: one channel's loads are issued into registers and fly under the previous channel's FMAs
    float4 xq4[PER4];
    int msrc[PER4], mdst[PER4];
#pragma unroll
    for (int u = 0; u < PER4; ++u) {
        const int q = tid + u * NT;
        const int iy = q / WIN4, xq = q % WIN4;
        const int gy = oy0 - HALO + iy, gx0 = ox0 - 8 + 4 * xq;
        const bool in = iy < IH && gy >= 0 && gy < p.H && gx0 >= 0 && gx0 < p.W;
        msrc[u] = in ? gy * p.W + gx0 : -1;
        mdst[u] = iy < IH ? (iy * IWP + 4 * xq) | (xq << 20) : -1;
    }
    auto plane = [&](int c) { return (c < C0) ? p.in0 + (size_t)(b0 * C0 + c) * p.H * p.W : p.in1 + (size_t)(b1 * C1 + (c - C0)) * p.H * p.W; };
    auto stage_load = [&](int c) {
        const float* src = plane(c);
        if (vec) {
#pragma unroll
            for (int u = 0; u < PER4; ++u) xq4[u] = mi_ldg4(src + (msrc[u] >= 0 ? msrc[u] : 0));
        }
    };
    auto stage_write = [&](int c) {
        if (vec) {
#pragma unroll
            for (int u = 0; u < PER4; ++u) {
                if (mdst[u] >= 0) {
                    const int d = mdst[u] & 0xfffff, xq = mdst[u] >> 20;
                    const float xe[4] = {xq4[u].x, xq4[u].y, xq4[u].z, xq4[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ix = 4 * xq - 1 + e;
                        if (ix >= 0 && ix < IW) smem[d - 1 + e] = msrc[u] >= 0 ? xe[e] : 0.0f;
                    }
                }
            }
        } else {
            const float* src = plane(c);
            for (int idx = tid; idx < IH * IW; idx += NT) {
                const int ix = idx % IW, iy = idx / IW;
                const int gy = oy0 - HALO + iy, gx = ox0 - HALO + ix;
                float v = 0.0f;
                if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) v = src[(size_t)gy * p.W + gx];
                smem[iy * IWP + ix] = v;
            }
        }
    };

    stage_load(0);
    for (int c = 0; c < Cin; ++c) {
        __syncthreads();
        stage_write(c);
        __syncthreads();
        if (c + 1 < Cin) stage_load(c + 1);
        const float* w3 = w3g + (size_t)c * 9 * 4;
        const float* w7 = w7g + (size_t)c * 49 * 2;
        const float* w15 = w15g + (size_t)c * 225 * 2;
#pragma unroll
        for (int ky = 0; ky < 15; ++ky) {
            float in[18];
            const float* row = &smem[(ty + ky) * IWP + tx * 4];
#pragma unroll
            for (int j = 0; j < 18; ++j) in[j] = row[j];
#pragma unroll
            for (int kx = 0; kx < 15; ++kx) {
                const float wa = w15[(ky * 15 + kx) * 2 + 0], wb = w15[(ky * 15 + kx) * 2 + 1];
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    acc[px][6] = fmaf(in[px + kx], wa, acc[px][6]);
                    acc[px][7] = fmaf(in[px + kx], wb, acc[px][7]);
                }
            }
            if (ky >= 4 && ky < 11) {
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    const float wa = w7[((ky - 4) * 7 + kx) * 2 + 0], wb = w7[((ky - 4) * 7 + kx) * 2 + 1];
#pragma unroll
                    for (int px = 0; px < 4; ++px) {
                        acc[px][4] = fmaf(in[px + kx + 4], wa, acc[px][4]);
                        acc[px][5] = fmaf(in[px + kx + 4], wb, acc[px][5]);
                    }
                }
            }
            if (ky >= 6 && ky < 9) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                    for (int co = 0; co < 4; ++co) {
                        const float wv = w3[((ky - 6) * 3 + kx) * 4 + co];
#pragma unroll
                        for (int px = 0; px < 4; ++px) acc[px][co] = fmaf(in[px + kx + 6], wv, acc[px][co]);
                    }
                }
            }
        }
    }

    const int oy = oy0 + ty, ox = ox0 + tx * 4;
    const bool row_ok = oy < p.H;
    float ssum[COUT], ssq[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float* bp = co < 4 ? p.bias[0] : (co < 6 ? p.bias[1] : p.bias[2]);
        const float bv = bp ? bp[co < 4 ? co : (co < 6 ? co - 4 : co - 6)] : 0.0f;
        float s = 0.0f, q = 0.0f;
        if (row_ok) {
            float* dst = p.out + ((size_t)(b * COUT + co) * p.H + oy) * p.W + ox;
            const float* add = p.addend ? p.addend + ((size_t)(b * COUT + co) * p.H + oy) * p.W + ox : nullptr;
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const float v = acc[px][co] + bv + ((add && ox + px < p.W) ? add[px] : 0.0f);
                if (ox + px < p.W) { dst[px] = v; s += v; q = fmaf(v, v, q); }
            }
        }
        ssum[co] = s;
        ssq[co] = q;
    }
    if (p.out_stats) {
        constexpr int R = 2 * COUT, SEG = NT / R;
        __syncthreads();
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            smem[(2 * co) * (NT + 1) + tid] = ssum[co];
            smem[(2 * co + 1) * (NT + 1) + tid] = ssq[co];
        }
        __syncthreads();
        const int rrow = tid / SEG, seg = tid % SEG;
        float a = 0.0f;
        for (int i = seg; i < NT; i += SEG) a += smem[rrow * (NT + 1) + i];
#pragma unroll
        for (int o = SEG / 2; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (seg == 0) p.out_stats[((size_t)(b * COUT + (rrow >> 1)) * gridDim.x + tile) * 2 + (rrow & 1)] = a;
    }
}

// ---- generic (any dim / kernel sizes): one work-item per output pixel, one output channel per
// blockIdx.z; taps read through L1/L2.  Correct for every constructor argument, not tuned.
template <int NT, int TW>
__global__ __launch_bounds__(NT) void crossembed_generic_kernel(const mi_crossembed_params p, const int Ctot) {
    constexpr int TH = NT / TW;
    __shared__ float red[2][NT / 64];
    const int tid = threadIdx.x;
    const int tiles_x = (p.W + TW - 1) / TW;
    const int tile = blockIdx.x;
    const int oy = (tile / tiles_x) * TH + tid / TW, ox = (tile % tiles_x) * TW + tid % TW;
    const int b = blockIdx.y, co_g = blockIdx.z;
    int ki = 0, co = co_g;
    while (ki < p.n_kernels - 1 && co >= p.cout[ki]) { co -= p.cout[ki]; ++ki; }
    const int K = p.ksize[ki], pad = (K - 1) / 2, CO = p.cout[ki];
    const int C0 = p.C0, C1 = p.in1 ? p.C1 : 0, Cin = C0 + C1;
    const int b0 = p.in0_batch_mod > 0 ? b % p.in0_batch_mod : b;
    const int b1 = p.in1_batch_mod > 0 ? b % p.in1_batch_mod : b;
    float acc = 0.0f;
    const bool ok = oy < p.H && ox < p.W;
    if (ok) {
        for (int c = 0; c < Cin; ++c) {
            const float* src = (c < C0) ? p.in0 + (size_t)(b0 * C0 + c) * p.H * p.W
                                        : p.in1 + (size_t)(b1 * C1 + (c - C0)) * p.H * p.W;
            const float* wc = p.w[ki] + (size_t)c * K * K * CO + co;
            for (int ky = 0; ky < K; ++ky) {
                const int gy = oy - pad + ky;
                if (gy < 0 || gy >= p.H) continue;
                for (int kx = 0; kx < K; ++kx) {
                    const int gx = ox - pad + kx;
                    if (gx < 0 || gx >= p.W) continue;
                    acc = fmaf(src[(size_t)gy * p.W + gx], wc[(ky * K + kx) * CO], acc);
                }
            }
        }
        if (p.bias[ki]) acc += p.bias[ki][co];
        if (p.addend) acc += p.addend[((size_t)(b * Ctot + co_g) * p.H + oy) * p.W + ox];
        p.out[((size_t)(b * Ctot + co_g) * p.H + oy) * p.W + ox] = acc;
    }
    if (p.out_stats) {
        float s = ok ? acc : 0.0f, q = ok ? acc * acc : 0.0f;
        s = mi_wave_sum(s);
        q = mi_wave_sum(q);
        if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = q; }
        __syncthreads();
        if (tid < 2) {
            float a = 0.0f;
            for (int w = 0; w < NT / 64; ++w) a += red[tid][w];
            p.out_stats[((size_t)(b * Ctot + co_g) * gridDim.x + tile) * 2 + tid] = a;
        }
    }
}

}  // namespace

extern "C" int mi_crossembed_fwd(const mi_crossembed_params* pp, void* stream) {
    const mi_crossembed_params& p = *pp;
    hipStream_t st = (hipStream_t)stream;
    if (p.n_kernels < 1 || p.n_kernels > 3 || p.B <= 0) { mi_set_error("mi_crossembed_fwd: bad n_kernels/B"); return MI_ERR_INVALID; }
    int Ctot = 0;
    for (int i = 0; i < p.n_kernels; ++i) Ctot += p.cout[i];
    const bool fast = p.n_kernels == 3 && p.ksize[0] == 3 && p.ksize[1] == 7 && p.ksize[2] == 15 &&
                      p.cout[0] == 4 && p.cout[1] == 2 && p.cout[2] == 2;
    int th, tw;
    if (mi_conv_tile_shape(p.tile_cfg, &th, &tw) != MI_OK) { mi_set_error("mi_crossembed_fwd: bad tile_cfg"); return MI_ERR_INVALID; }
    const int tiles = ((p.H + th - 1) / th) * ((p.W + tw - 1) / tw);
    if (fast) {
        switch (p.tile_cfg) {
            case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_422_kernel<256, 64>), dim3(tiles, p.B), dim3(256), 0, st, p, p.w[0], p.w[1], p.w[2]); break;
            case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_422_kernel<256, 32>), dim3(tiles, p.B), dim3(256), 0, st, p, p.w[0], p.w[1], p.w[2]); break;
            default: hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_422_kernel<64, 32>), dim3(tiles, p.B), dim3(64), 0, st, p, p.w[0], p.w[1], p.w[2]); break;
        }
    } else {
        // generic tiles: same (th x tw) footprint so out_nt matches mi_conv_tile_shape; th*tw work-items <= 1024
        switch (p.tile_cfg) {
            case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_generic_kernel<1024, 64>), dim3(tiles, p.B, Ctot), dim3(1024), 0, st, p, Ctot); break;
            case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_generic_kernel<1024, 32>), dim3(tiles, p.B, Ctot), dim3(1024), 0, st, p, Ctot); break;
            default: hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_generic_kernel<256, 32>), dim3(tiles, p.B, Ctot), dim3(256), 0, st, p, Ctot); break;
        }
    }
    return mi_check_launch("crossembed");
}
