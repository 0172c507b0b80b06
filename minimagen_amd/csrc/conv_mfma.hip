// The 3x3 stride-1 convolutions of the ResnetBlocks (layers.py:131-145, 415-439) on the matrix cores.
//
// Why: the PMC profile of the VALU direct convolution (profiles/r01_*) shows it VALU-issue-bound with only ~45 % of the
// issued instructions being FMAs; v_mfma_f32_16x16x16_f16 does 8192 flop per ~19 cycles and overlaps with VALU work.
// How: implicit GEMM per wave, D[co 16][px 16] += W[co][k] . act[k][px] with K = 16 input channels of one tap; every fp32
// operand is split x = hi + lo (fp16 each, 22 mantissa bits kept) and the product taken as hi*hi + hi*lo + lo*hi, which
// stays inside the fp32 parity tolerance (error ~2^-21 per product).  Activations are staged ONCE per 16-channel round
// through LDS as channel-quads [quad][row][col][4 halves] (GroupNorm-apply + scale/shift + SiLU + hi/lo split fused into the
// staging), so a lane's B operand is one aligned 8-byte read and neighbouring pixels are neighbouring banks.  The 1x1
// res_conv of the ResnetBlock rides the same loop as extra K rounds with a single (centre) tap.
#include "common.hip.h"

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

template <int TW_, bool GN_>
struct MfmaCfg {
    static constexpr int TW = TW_, TH = 512 / TW_;          // 8x64 or 16x32 output pixels per workgroup (4 waves x 8 pixel-tiles)
    static constexpr bool GN = GN_;
    static constexpr int IH = TH + 2, WIN4 = (TW + 8) / 4, PW = 4 * WIN4;   // staged rows; float4 groups / pixels per staged row
    static constexpr int XT = TW / 16;                      // pixel-tiles per output row
    static constexpr int RW = 8 / XT;                       // output rows per wave
    static constexpr int UNITS = 4 * IH * WIN4;             // (channel quad, row, float4 group)
    static constexpr int PER = (UNITS + 255) / 256;
    static constexpr int PLANE = 4 * IH * PW;               // 8-byte entries per plane (hi or lo)
};

template <class CFG>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(const mi_conv_params p, const _Float16* __restrict__ wf, const _Float16* __restrict__ rwf) {
    constexpr int TW = CFG::TW, TH = CFG::TH, IH = CFG::IH, WIN4 = CFG::WIN4, PW = CFG::PW, XT = CFG::XT, RW = CFG::RW, PER = CFG::PER;
    constexpr bool GN = CFG::GN;
    __shared__ __attribute__((aligned(16))) f16x4 actH[CFG::PLANE], actL[CFG::PLANE];
    __shared__ __attribute__((aligned(16))) f16x4 wlds[9 * 64 * 2];      // this round's A fragments: [tap][lane][hi | lo]
    __shared__ __attribute__((aligned(16))) float4 chP[MI_MAX_CIN + 1];
    __shared__ double chS[MI_MAX_CIN], chQ[MI_MAX_CIN];
    __shared__ float gMean[MI_MAX_GROUPS], gRstd[MI_MAX_GROUPS];
    __shared__ float red[4][32];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int tiles_x = (p.W + TW - 1) / TW;
    const int tile = blockIdx.x;
    const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
    const int b = blockIdx.y, mz = blockIdx.z, co0 = mz * 16;
    const int C0 = p.in0.C, C1 = p.in1.data ? p.in1.C : 0, Cin = C0 + C1;
    const int KC = (Cin + 15) / 16;
    const int Cr0 = (p.res0.data && rwf) ? p.res0.C : 0, Cr1 = (Cr0 && p.res1.data) ? p.res1.C : 0, Cres = Cr0 + Cr1;
    const int RC = (Cres + 15) / 16;
    const int H = p.H, W = p.W;

    // ---------------- per-channel affine of the fused GroupNorm / scale-shift (same scheme as conv.hip)
    if constexpr (GN) {
        if (tid < 64) {
            int TPC = 1;
            while (TPC < 64 && TPC * 2 * Cin <= 64) TPC *= 2;
            const int CPP = 64 / TPC;
            for (int base = 0; base < Cin; base += CPP) {
                const int c = base + tid / TPC, sub = tid % TPC;
                double s = 0.0, q = 0.0;
                if (c < Cin) {
                    const bool second = c >= C0;
                    const mi_act& a = second ? p.in1 : p.in0;
                    const int cc = second ? c - C0 : c;
                    const int ba = a.bmod > 0 ? b % a.bmod : b;
                    const float* st = a.stats + ((size_t)(ba * a.C + cc) * a.nt) * 2;
                    for (int t = sub; t < a.nt; t += TPC) {
                        const float2 v = *reinterpret_cast<const float2*>(st + 2 * t);
                        s += (double)v.x;
                        q += (double)v.y;
                    }
                    s *= (double)a.scale;
                    q *= (double)a.scale * (double)a.scale;
                }
                for (int o = TPC >> 1; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
                if (c < Cin && sub == 0) { chS[c] = s; chQ[c] = q; }
            }
        }
        __syncthreads();
        const int cpg = Cin / p.gn_groups;
        for (int g = tid; g < p.gn_groups; g += 256) {
            double s = 0.0, q = 0.0;
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s += chS[c]; q += chQ[c]; }
            const double n = (double)cpg * (double)H * (double)W;
            const double mean = s / n;
            double var = q / n - mean * mean;
            var = var > 0.0 ? var : 0.0;
            gMean[g] = (float)mean;
            gRstd[g] = (float)(1.0 / sqrt(var + (double)p.gn_eps));
        }
        __syncthreads();
        for (int c = tid; c < Cin; c += 256) {
            const int g = c / cpg;
            float A = gRstd[g] * p.gn_gamma[c];
            float Bc = p.gn_beta[c] - gMean[g] * A;
            if (p.scale_shift) {
                const float* ss = p.scale_shift + (size_t)b * p.ss_stride + p.ss_off;
                const float sc = ss[c] + 1.0f, sh = ss[Cin + c];
                A *= sc;
                Bc = Bc * sc + sh;
            }
            A *= (c >= C0) ? p.in1.scale : p.in0.scale;
            chP[c] = make_float4(A, Bc, A * -1.44269504088896340736f, Bc * -1.44269504088896340736f);
        }
    } else {
        for (int c = tid; c < Cin; c += 256) chP[c] = make_float4((c >= C0) ? p.in1.scale : p.in0.scale, 0.0f, 0.0f, 0.0f);
    }

    // ---------------- staging slots (geometry only): unit = (channel quad, staged row, float4 group)
    int msrc[PER], mdst[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int q = tid + u * 256;
        const int xg = q % WIN4, r = q / WIN4;
        const int iy = r % IH, qd = r / IH;
        const int gy = oy0 - 1 + iy, gx0 = ox0 - 4 + 4 * xg;
        const bool used = qd < 4;
        const bool in = used && gy >= 0 && gy < H && gx0 >= 0 && gx0 < W;
        msrc[u] = in ? gy * W + gx0 : -1;
        mdst[u] = used ? ((qd * IH + iy) * PW + 4 * xg) | (qd << 24) : -1;
    }

    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // identity residual (layers.py:439 with res_conv = Identity): fetched now, consumed in the epilogue, so the loads fly
    // under the whole MFMA loop instead of serialising behind it
    const bool idres = p.res0.data && !rwf;
    const int br0 = p.res0.bmod > 0 ? b % p.res0.bmod : b;
    float resv[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + 4 * lg + r, oy = oy0 + wave * RW + t / XT, ox = ox0 + 16 * (t % XT) + lq;
            resv[t][r] = (idres && co < p.Cout && oy < H && ox < W) ? p.res0.data[((size_t)(br0 * p.res0.C + co) * H + oy) * W + ox] * p.res0.scale : 0.0f;
        }

    // one round = 16 channels of either the convolution input (9 taps, activated) or the residual input (centre tap, raw)
    const int rounds = KC + RC;
    for (int rnd = 0; rnd < rounds; ++rnd) {
        const bool isres = rnd >= KC;
        const int kc = isres ? rnd - KC : rnd;
        const int ntap = isres ? 1 : 9;
        const mi_act& t0 = isres ? p.res0 : p.in0;
        const mi_act& t1 = isres ? p.res1 : p.in1;
        const int Ca = isres ? Cr0 : C0, Cb = isres ? Cr1 : C1, Ct = Ca + Cb;
        const int ba = t0.bmod > 0 ? b % t0.bmod : b, bb = t1.bmod > 0 ? b % t1.bmod : b;
        __syncthreads();          // previous round's LDS fully consumed (and chP visible before the first staging)
        // A fragments of this round -> LDS
        {
            const _Float16* src = isres ? rwf + ((size_t)(mz * RC + kc)) * 64 * 8 : wf + ((size_t)(mz * KC + kc) * 9) * 64 * 8;
            for (int i = tid; i < ntap * 64; i += 256) {
                const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)i * 8);
                *reinterpret_cast<uint4*>(&wlds[2 * i]) = v;
            }
        }
        // activations: load 4 channels x 4 pixels, transform, split, write channel-quads
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (mdst[u] < 0) continue;
            const int qd = mdst[u] >> 24, dst = mdst[u] & 0xffffff;
            float v[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 16 * kc + 4 * qd + j;
                float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
                const bool live = msrc[u] >= 0 && c < Ct;
                if (live) {
                    const bool second = c >= Ca;
                    const float* base = second ? t1.data + (size_t)(bb * Cb + (c - Ca)) * H * W : t0.data + (size_t)(ba * Ca + c) * H * W;
                    x4 = *reinterpret_cast<const float4*>(base + msrc[u]);
                }
                const float xe[4] = {x4.x, x4.y, x4.z, x4.w};
                if (isres) {
                    const float sc = (c >= Ca) ? t1.scale : t0.scale;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j][e] = live ? xe[e] * sc : 0.0f;
                } else {
                    const float4 P = chP[live ? c : MI_MAX_CIN];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (GN) {
                            const float a = fmaf(xe[e], P.x, P.y);
                            const float ex = __builtin_amdgcn_exp2f(fmaf(xe[e], P.z, P.w));
                            v[j][e] = live ? a * __builtin_amdgcn_rcpf(1.0f + ex) : 0.0f;      // zero padding follows the activation
                        } else {
                            v[j][e] = live ? xe[e] * P.x : 0.0f;
                        }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f16x4 hi, lo;
                const float ve[4] = {v[0][e], v[1][e], v[2][e], v[3][e]};
                mi_split_f16(ve, hi, lo);
                actH[dst + e] = hi;
                actL[dst + e] = lo;
            }
        }
        __syncthreads();
        if (tid == 0 && rnd == 0) chP[MI_MAX_CIN] = make_float4(0.f, 0.f, 0.f, 0.f);
        // MFMA loop: this lane's B operand = channels 4lg..4lg+3 (quad lg) of pixel lq of the tile, shifted by the tap
        for (int tap = 0; tap < ntap; ++tap) {
            const int ky = isres ? 1 : tap / 3, kx = isres ? 1 : tap % 3;
            const f16x4 ahi = wlds[2 * (tap * 64 + lane)], alo = wlds[2 * (tap * 64 + lane) + 1];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int row = wave * RW + t / XT, xt = t % XT;
                const int idx = (lg * IH + row + ky) * PW + 4 + 16 * xt + lq + kx - 1;
                const f16x4 bhi = actH[idx], blo = actL[idx];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(alo, bhi, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(ahi, blo, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(ahi, bhi, acc[t], 0, 0, 0);
            }
        }
    }

    // ---------------- epilogue: this lane holds output channels co0 + 4lg + r of pixel lq of each of its 8 tiles
    float csum[4] = {0.f, 0.f, 0.f, 0.f}, csq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = co0 + 4 * lg + r;
        if (co >= p.Cout) continue;
        float bv = p.bias ? p.bias[co] : 0.0f;
        if (Cres && p.res_b) bv += p.res_b[co];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int oy = oy0 + wave * RW + t / XT, ox = ox0 + 16 * (t % XT) + lq;
            if (oy < H && ox < W) {
                const float y = acc[t][r] + bv + resv[t][r];
                p.out[((size_t)(b * p.Cout + co) * H + oy) * W + ox] = y;
                csum[r] += y;
                csq[r] = fmaf(y, y, csq[r]);
            }
        }
    }
    if (p.out_stats) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { csum[r] += __shfl_xor(csum[r], o); csq[r] += __shfl_xor(csq[r], o); }
        }
        if (lq == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { red[wave][2 * (4 * lg + r)] = csum[r]; red[wave][2 * (4 * lg + r) + 1] = csq[r]; }
        }
        __syncthreads();
        if (tid < 32 && co0 + (tid >> 1) < p.Cout)
            p.out_stats[((size_t)(b * p.Cout + co0 + (tid >> 1)) * gridDim.x + tile) * 2 + (tid & 1)] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    }
}

template <int TW, bool GN>
int launch(const mi_conv_params& p, hipStream_t st) {
    using CFG = MfmaCfg<TW, GN>;
    const int tiles = ((p.H + CFG::TH - 1) / CFG::TH) * ((p.W + TW - 1) / TW);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_mfma_kernel<CFG>), dim3(tiles, p.B, (p.Cout + 15) / 16), dim3(256), 0, st, p,
                       (const _Float16*)p.w_f16, (const _Float16*)p.res_w_f16);
    return mi_check_launch("conv3x3_mfma_kernel");
}

}  // namespace

int mi_conv_mfma_launch(const mi_conv_params& p, hipStream_t st) {
    const int Cin = p.in0.C + (p.in1.data ? p.in1.C : 0);
    if (p.ksize != 3 || p.stride != 1 || p.up2) { mi_set_error("mi_conv_fwd: the matrix-core path is k3 s1 only"); return MI_ERR_UNSUPPORTED; }
    if ((p.W & 3) || Cin > MI_MAX_CIN) { mi_set_error("mi_conv_fwd: matrix-core path needs W %% 4 == 0 and Cin <= %d", MI_MAX_CIN); return MI_ERR_UNSUPPORTED; }
    if (p.res0.data && p.res_w && !p.res_w_f16) { mi_set_error("mi_conv_fwd: matrix-core path needs res_w_f16 for a 1x1 residual"); return MI_ERR_INVALID; }
    const int cfg = p.tile_cfg & 0xff;
    if (cfg != 3 && cfg != 4) { mi_set_error("mi_conv_fwd: matrix-core path uses tile_cfg 3 (8x64) or 4 (16x32)"); return MI_ERR_INVALID; }
    if (p.gn_groups > 0) return cfg == 3 ? launch<64, true>(p, st) : launch<32, true>(p, st);
    return cfg == 3 ? launch<64, false>(p, st) : launch<32, false>(p, st);
}
