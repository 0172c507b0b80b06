// The 3x3 stride-1 convolutions of the ResnetBlocks (layers.py:131-145, 415-439) on the matrix cores.
//
// Why: the PMC profile of the VALU direct convolution (profiles/r01_*) shows it VALU-issue-bound with only ~45 % of the
// issued instructions being FMAs; v_mfma_f32_16x16x16_f16 does 8192 flop per ~19 cycles and overlaps with VALU work.
// How: implicit GEMM per wave, D[px 16][co 16] += act[px][k] . W[k][co] with K = 16 input channels of one tap (so a lane ends
// up with four consecutive pixels of one output channel: float4 stores, two shuffles for the statistics); every fp32
// operand is split x = hi + lo (fp16 each, 22 mantissa bits kept) and the product taken as hi*hi + hi*lo + lo*hi, which
// stays inside the fp32 parity tolerance (error ~2^-21 per product).  Activations are staged ONCE per 16-channel round
// through LDS as channel-quads [quad][row][col][4 halves] (GroupNorm-apply + scale/shift + SiLU + hi/lo split fused into the
// staging), so a lane's A operand is one aligned 8-byte read and neighbouring pixels are neighbouring banks.  The 1x1
// res_conv of the ResnetBlock rides the same loop as extra K rounds with a single (centre) tap.
#include "common.hip.h"

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifdef MI_TRACE
// development aid (tools/gpu_trace_conv.sh): per-phase shader-clock stamps of the first workgroups of the last launch
__device__ unsigned long long mi_trace_buf[1024 * 8];
#define MI_STAMP(k) do { if (threadIdx.x == 0) { const int wg = (blockIdx.y * gridDim.x + blockIdx.x); if (wg < 1024) mi_trace_buf[wg * 8 + (k)] = clock64(); } } while (0)
extern "C" int mi_debug_read_trace(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mi_trace_buf), bytes); }
#else
#define MI_STAMP(k) do { } while (0)
#endif

namespace {

template <int TW_, bool GN_, int NW_, bool HALF_>
struct MfmaCfg {
    static constexpr int TW = TW_, TH = 512 / TW_;          // 8x64 or 16x32 output pixels per workgroup = 32 pixel-tiles of 16
    static constexpr bool GN = GN_, HALF = HALF_;          // HALF: single fp16 term per product (reduced-precision configuration)
    static constexpr int NW = NW_, NT = 64 * NW_;           // waves / work-items per workgroup
    static constexpr int TPW = 32 / NW_;                    // pixel-tiles per wave
    static constexpr int IH = TH + 2, WIN4 = (TW + 8) / 4, PW = 4 * WIN4;   // staged rows; float4 groups / pixels per staged row
    static constexpr int XT = TW / 16;                      // pixel-tiles per output row
    static constexpr int UNITS = 4 * IH * WIN4;             // (channel quad, row, float4 group)
    static constexpr int PER = (UNITS + NT - 1) / NT;
    static constexpr int PLANE = 4 * IH * PW;               // 8-byte entries per plane (hi or lo)
};

template <class CFG>
// two workgroups per CU (LDS-bound): 2 (NW = 4) or 4 (NW = 8) waves per SIMD -> at most 256 / 128 registers per work-item
__global__ __launch_bounds__(CFG::NT, CFG::NW / 2) void conv3x3_mfma_kernel(const mi_conv_params p, const _Float16* __restrict__ wf, const _Float16* __restrict__ rwf) {
    constexpr int TW = CFG::TW, TH = CFG::TH, IH = CFG::IH, WIN4 = CFG::WIN4, PW = CFG::PW, XT = CFG::XT, PER = CFG::PER;
    constexpr int NT = CFG::NT, NW = CFG::NW, TPW = CFG::TPW;
    constexpr bool GN = CFG::GN, HALF = CFG::HALF;
    __shared__ __attribute__((aligned(16))) f16x4 actH[CFG::PLANE], actL[HALF ? 1 : CFG::PLANE];
    __shared__ __attribute__((aligned(16))) f16x4 wl[5][64][4];         // this round's weight fragments: [tap pair][lane][hiA, hiB, loA, loB]
    __shared__ __attribute__((aligned(16))) float4 chP[MI_MAX_CIN + 1];
    __shared__ double chS[MI_MAX_CIN], chQ[MI_MAX_CIN];
    __shared__ float gMean[MI_MAX_GROUPS], gRstd[MI_MAX_GROUPS];
    __shared__ float red[NW][32];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int tiles_x = (p.W + TW - 1) / TW;
    const int tile = blockIdx.x;
    const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
    const int b = blockIdx.y, mz = blockIdx.z, co0 = mz * 16;
    const int C0 = p.in0.C, C1 = p.in1.data ? p.in1.C : 0, Cin = C0 + C1;
    const int KC = (Cin + 15) / 16;
    const int Cr0 = (p.res0.data && rwf) ? p.res0.C : 0, Cr1 = (Cr0 && p.res1.data) ? p.res1.C : 0, Cres = Cr0 + Cr1;
    const int RC = (Cres + 15) / 16;
    const int H = p.H, W = p.W, HW = H * W;
    const int rounds = KC + RC;

    MI_STAMP(0);
    // ---------------- GroupNorm statistics of the input: wave 0 reduces the producers' per-tile partial sums (fp64) FIRST, so its
    // (short, latency-bound) loads are already in flight while every wave computes its staging geometry and issues round 0's loads
    if constexpr (GN) {
        mi_gn_channel_totals(p.in0, p.in1, C0, Cin, b, tid, NT, chS, chQ);
    }
    // ---------------- staging slots (geometry only): unit = (channel quad, staged row, float4 group)
    int msrc[PER], mdst[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int q = tid + u * NT;
        const int xg = q % WIN4, r = q / WIN4;
        const int iy = r % IH, qd = r / IH;
        const int gy = oy0 - 1 + iy, gx0 = ox0 - 4 + 4 * xg;
        const bool used = qd < 4;
        const bool in = used && gy >= 0 && gy < H && gx0 >= 0 && gx0 < W;
        msrc[u] = in ? gy * W + gx0 : -1;
        mdst[u] = used ? ((qd * IH + iy) * PW + 4 * xg) | (qd << 24) : -1;
    }

    // one round = 16 channels of either the convolution input (9 taps, activated) or the residual input (centre tap, raw).
    // Raw loads of a round: 4 channels x 4 pixels per slot, kept in registers until the transform -- round 0's are issued
    // before the GroupNorm-statistics prologue and every later round's under the previous round's MFMA loop.
    float4 raw[PER][4];
    auto load_raw = [&](int rnd) {
        const bool isres = rnd >= KC;
        const int kc = isres ? rnd - KC : rnd;
        const mi_act& t0 = isres ? p.res0 : p.in0;
        const mi_act& t1 = isres ? p.res1 : p.in1;
        const int Ca = isres ? Cr0 : C0, Cb = isres ? Cr1 : C1, Ct = Ca + Cb;
        const int ba = mi_row_of(b, t0.bmod), bb = mi_row_of(b, t1.bmod);
        const float* base0 = t0.data + (size_t)ba * Ca * HW;
        const float* base1 = Cb ? t1.data + (size_t)bb * Cb * HW : base0;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int qd = mdst[u] >> 24;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 16 * kc + 4 * qd + j;
                float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (mdst[u] >= 0 && msrc[u] >= 0 && c < Ct) {
                    const float* src = c >= Ca ? base1 + (c - Ca) * HW : base0 + c * HW;
                    x4 = mi_ldg4(src + msrc[u]);
                }
                raw[u][j] = x4;
            }
        }
    };
    load_raw(0);

    // identity residual (layers.py:439 with res_conv = Identity): fetched now, consumed in the epilogue.  This lane owns
    // output channel co0 + lq and the four pixels 4lg .. 4lg+3 of each of its pixel-tiles.
    const int co = co0 + lq;
    const bool idres = p.res0.data && !rwf;
    const int br0 = mi_row_of(b, p.res0.bmod);
    float4 resv[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int gt = wave * TPW + t, oy = oy0 + gt / XT, ox = ox0 + 16 * (gt % XT) + 4 * lg;
        resv[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idres && co < p.Cout && oy < H && ox < W)
            resv[t] = mi_ldg4(p.res0.data + ((size_t)(br0 * p.res0.C + co) * H + oy) * W + ox);
    }

    MI_STAMP(1);
    // ---------------- per-channel affine of the fused GroupNorm / scale-shift (same scheme as conv.hip)
    if (tid == 0) chP[MI_MAX_CIN] = make_float4(0.f, 0.f, 0.f, 0.f);      // what padded / out-of-range slots read: activation(0) = 0
    if constexpr (GN) {
        __syncthreads();
        const int cpg = Cin / p.gn_groups;
        for (int g = tid; g < p.gn_groups; g += NT)
            mi_gn_group_moments(chS, chQ, g * cpg, (g + 1) * cpg, (double)cpg * (double)H * (double)W, p.gn_eps, gMean[g], gRstd[g]);
        __syncthreads();
        for (int c = tid; c < Cin; c += NT) {
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
        for (int c = tid; c < Cin; c += NT) chP[c] = make_float4((c >= C0) ? p.in1.scale : p.in0.scale, 0.0f, 0.0f, 0.0f);
    }

    MI_STAMP(2);
    f32x4 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int rnd = 0; rnd < rounds; ++rnd) {
        const bool isres = rnd >= KC;
        const int kc = isres ? rnd - KC : rnd;
        const int ntap = isres ? 1 : 9;
        const int Ct = isres ? Cres : Cin;
        __syncthreads();          // previous round's LDS fully consumed (and chP visible before the first transform)
        if (rnd == 0) MI_STAMP(3);
        // this round's weight fragments (B operand: lane = output channel lq, input-channel quad lg; 4 hi | 4 lo halves per tap)
        {
            const _Float16* src = isres ? rwf + ((size_t)(mz * RC + kc)) * 64 * 8 : wf + ((size_t)(mz * KC + kc) * 9) * 64 * 8;
            for (int i = tid; i < ntap * 64; i += NT) {
                union { uint4 u; f16x4 h[2]; } v;
                v.u = mi_ldg4u(src + (size_t)i * 8);
                const int tap = isres ? 8 : i >> 6, ln = i & 63;
                wl[tap >> 1][ln][tap & 1] = v.h[0];
                wl[tap >> 1][ln][2 + (tap & 1)] = v.h[1];
            }
        }
        // activations: transform, split, write channel-quads
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (mdst[u] < 0) continue;
            const int qd = mdst[u] >> 24, dst = mdst[u] & 0xffffff;
            float v[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 16 * kc + 4 * qd + j;
                const bool live = msrc[u] >= 0 && c < Ct;
                const float xe[4] = {raw[u][j].x, raw[u][j].y, raw[u][j].z, raw[u][j].w};
                if (isres) {
                    const float sc = (c >= Cr0) ? p.res1.scale : p.res0.scale;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j][e] = xe[e] * sc;                      // not-live slots hold zeros
                } else {
                    const float4 P = chP[live ? c : MI_MAX_CIN];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (GN) {
                            // zero padding follows the activation: a not-live slot has x = 0 and P = 0 -> 0 * 1/(1+1) = 0
                            const float a = fmaf(xe[e], P.x, P.y);
                            const float ex = __builtin_amdgcn_exp2f(fmaf(xe[e], P.z, P.w));
                            v[j][e] = a * __builtin_amdgcn_rcpf(1.0f + ex);
                        } else {
                            v[j][e] = xe[e] * P.x;
                        }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f16x4 hi, lo;
                const float ve[4] = {v[0][e], v[1][e], v[2][e], v[3][e]};
                if constexpr (HALF) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) hi[k] = (_Float16)ve[k];
                    actH[dst + e] = hi;
                } else {
                    mi_split_f16(ve, hi, lo);
                    actH[dst + e] = hi;
                    actL[dst + e] = lo;
                }
            }
        }
        __syncthreads();
        if (rnd == 0) MI_STAMP(4);
        if (rnd + 1 < rounds) load_raw(rnd + 1);
        // MFMA loop, D[px 16][co 16] += act[px][k] . W[k][co]: this lane's A operand = channels 4lg..4lg+3 (quad lg) of pixel
        // lq of the tile, shifted by the tap
        // Taps go through the K = 32 instruction in pairs (A, B): {hiA,hiB}.{whiA,whiB} + {loA,loB}.{whiA,whiB} + {hiA,hiB}.{wloA,wloB};
        // the odd tap (and the single centre tap of a residual round) uses the K = 16 form.
        const int npair = isres ? 0 : 4;
#pragma unroll 1
        for (int pr = 0; pr < npair; ++pr) {        // not unrolled: the scheduler otherwise hoists every tile's LDS reads (250+ registers)
            const int ta = 2 * pr, tb = 2 * pr + 1;
            const f16x8 whi = *reinterpret_cast<const f16x8*>(&wl[pr][lane][0]);
            const f16x8 wlo = *reinterpret_cast<const f16x8*>(&wl[pr][lane][2]);
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int gt = wave * TPW + t;
                const int base = (lg * IH + gt / XT) * PW + 4 + 16 * (gt % XT) + lq - 1;
                const int ia = base + (ta / 3) * PW + ta % 3, ib = base + (tb / 3) * PW + tb % 3;
                const f16x8 ahi = __builtin_shufflevector(actH[ia], actH[ib], 0, 1, 2, 3, 4, 5, 6, 7);
                if constexpr (!HALF) {
                    const f16x8 alo = __builtin_shufflevector(actL[ia], actL[ib], 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, whi, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, wlo, acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, whi, acc[t], 0, 0, 0);
            }
        }
        {
            const int tap = isres ? 0 : 8;
            const int ky = isres ? 1 : 2, kx = isres ? 1 : 2;
            const f16x4 whi = wl[4][lane][0], wlo = wl[4][lane][2];
            (void)tap;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int gt = wave * TPW + t;
                const int idx = (lg * IH + gt / XT + ky) * PW + 4 + 16 * (gt % XT) + lq + kx - 1;
                const f16x4 ahi = actH[idx];
                if constexpr (!HALF) {
                    const f16x4 alo = actL[idx];
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(alo, whi, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(ahi, wlo, acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(ahi, whi, acc[t], 0, 0, 0);
            }
        }
    }

    MI_STAMP(5);
    // ---------------- epilogue: float4 of four consecutive pixels per tile, one output channel per lane
    float csum = 0.f, csq = 0.f;
    if (co < p.Cout) {
        float bv = p.bias ? p.bias[co] : 0.0f;
        if (Cres && p.res_b) bv += p.res_b[co];
        const float rs = idres ? p.res0.scale : 0.0f;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int gt = wave * TPW + t, oy = oy0 + gt / XT, ox = ox0 + 16 * (gt % XT) + 4 * lg;
            if (oy < H && ox < W) {
                float4 y;
                y.x = acc[t][0] + bv + resv[t].x * rs;
                y.y = acc[t][1] + bv + resv[t].y * rs;
                y.z = acc[t][2] + bv + resv[t].z * rs;
                y.w = acc[t][3] + bv + resv[t].w * rs;
                mi_stg4(p.out + ((size_t)(b * p.Cout + co) * H + oy) * W + ox, y);
                csum += (y.x + y.y) + (y.z + y.w);
                csq = fmaf(y.x, y.x, fmaf(y.y, y.y, fmaf(y.z, y.z, fmaf(y.w, y.w, csq))));
            }
        }
    }
    if (p.out_stats) {
        csum += __shfl_xor(csum, 16); csq += __shfl_xor(csq, 16);
        csum += __shfl_xor(csum, 32); csq += __shfl_xor(csq, 32);
        if (lg == 0) { red[wave][2 * lq] = csum; red[wave][2 * lq + 1] = csq; }
        __syncthreads();
        if (tid < 32 && co0 + (tid >> 1) < p.Cout) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) a += red[w][tid];
            p.out_stats[((size_t)(b * p.Cout + co0 + (tid >> 1)) * gridDim.x + tile) * 2 + (tid & 1)] = a;
        }
    }
    MI_STAMP(6);
}

template <int TW, bool GN, int NW, bool HALF>
int launch(const mi_conv_params& p, hipStream_t st) {
    using CFG = MfmaCfg<TW, GN, NW, HALF>;
    const int tiles = ((p.H + CFG::TH - 1) / CFG::TH) * ((p.W + TW - 1) / TW);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_mfma_kernel<CFG>), dim3(tiles, p.B, (p.Cout + 15) / 16), dim3(CFG::NT), 0, st, p,
                       (const _Float16*)p.w_f16, (const _Float16*)p.res_w_f16);
    return mi_check_launch("conv3x3_mfma_kernel");
}

template <int TW, bool GN>
int launch_w(const mi_conv_params& p, hipStream_t st) {
    // MI_CONV_WAVES8: 8 waves x 4 pixel-tiles instead of 4 x 8 (same tile, same LDS; more waves in flight per CU)
    if (p.tile_cfg & MI_CONV_HALF) return launch<TW, GN, 8, true>(p, st);       // reduced-precision configuration: single fp16 term
    return (p.tile_cfg & MI_CONV_WAVES8) ? launch<TW, GN, 8, false>(p, st) : launch<TW, GN, 4, false>(p, st);
}

}  // namespace

int mi_conv_mfma_launch(const mi_conv_params& p, hipStream_t st) {
    const int Cin = p.in0.C + (p.in1.data ? p.in1.C : 0);
    if (p.ksize != 3 || p.stride != 1 || p.up2) { mi_set_error("mi_conv_fwd: the matrix-core path is k3 s1 only"); return MI_ERR_UNSUPPORTED; }
    if ((p.W & 3) || Cin > MI_MAX_CIN) { mi_set_error("mi_conv_fwd: matrix-core path needs W %% 4 == 0 and Cin <= %d", MI_MAX_CIN); return MI_ERR_UNSUPPORTED; }
    if (p.res0.data && p.res_w && !p.res_w_f16) { mi_set_error("mi_conv_fwd: matrix-core path needs res_w_f16 for a 1x1 residual"); return MI_ERR_INVALID; }
    const int cfg = p.tile_cfg & 0xff;
    if (cfg != 3 && cfg != 4) { mi_set_error("mi_conv_fwd: matrix-core path uses tile_cfg 3 (8x64) or 4 (16x32)"); return MI_ERR_INVALID; }
    const size_t biggest = (size_t)p.B * (p.in0.C > p.in1.C ? p.in0.C : p.in1.C) * p.H * p.W;
    if (biggest >= (1ull << 31)) { mi_set_error("mi_conv_fwd: matrix-core path indexes activations with 32-bit offsets"); return MI_ERR_UNSUPPORTED; }
    if (p.gn_groups > 0) return cfg == 3 ? launch_w<64, true>(p, st) : launch_w<32, true>(p, st);
    return cfg == 3 ? launch_w<64, false>(p, st) : launch_w<32, false>(p, st);
}
