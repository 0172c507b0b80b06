// K4 / K6 / K7 / K8 / final conv: the direct-convolution family for narrow NCHW fp32
// activations (8..32 channels), MinImagen layers.py:107-145 (Block), :308-319 (Downsample),
// :346-356 (Parallel, 1x1 folded into the 3x3 centre tap by the host), :502-515 (Upsample),
// :415/:439 (ResnetBlock residual).  HBM-bound by design: one pass over the input (halo tile
// staged once through LDS with GroupNorm-apply + scale/shift + SiLU fused into the staging),
// all output channels of a channel tile held in VGPRs per work-item, weights broadcast from
// SGPRs (wave-uniform addresses), and the NEXT GroupNorm's per-channel partial statistics
// emitted from the epilogue so no tensor is ever re-read for normalisation.
#include "common.hip.h"

#ifdef MI_TRACE
// development aid (tools/trace_conv.py): accumulated shader-clock time per phase of the first workgroups of the last launch
__device__ unsigned long long mi_trace_conv_buf[1024 * 8];
extern "C" int mi_debug_read_trace_conv(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mi_trace_conv_buf), bytes); }
#define MI_TSTART() unsigned long long mi_t_last = clock64(), mi_t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define MI_TPHASE(k) do { const unsigned long long mi_t_now = clock64(); mi_t_acc[k] += mi_t_now - mi_t_last; mi_t_last = mi_t_now; } while (0)
#define MI_TEND() do { if (threadIdx.x == 0) { const int wg = blockIdx.y * gridDim.x + blockIdx.x; if (wg < 1024) for (int k = 0; k < 8; ++k) mi_trace_conv_buf[wg * 8 + k] = mi_t_acc[k]; } } while (0)
#else
#define MI_TSTART() do { } while (0)
#define MI_TPHASE(k) do { } while (0)
#define MI_TEND() do { } while (0)
#endif

namespace {

template <int NT_, int TW_, int COUT_T_, int KS_, int S_, bool UP2_, bool VEC_, bool GN_>
struct ConvCfg {
    static constexpr int NT = NT_, TW = TW_, COUT_T = COUT_T_, KS = KS_, S = S_;
    static constexpr bool UP2 = UP2_, VEC = VEC_, GN = GN_;
    static constexpr int TXN = TW / 4;            // work-items along x (4 output pixels each)
    static constexpr int TH = NT / TXN;           // output rows per tile (1 row per work-item)
    static constexpr int IH = TH * S + KS - S;    // staged input rows / cols (with halo)
    static constexpr int IW = TW * S + KS - S;
    // vector staging path: every staged row is the ALIGNED float4 window of SOURCE pixels that covers the tile's halo:
    //   k3 s1: [ox0-4, ox0+TW+4)   k4 s2: [2ox0-4, 2ox0+2TW+4)   nearest-x2 + k3: source cols [ox0/2-4, ox0/2+TW/2+4)
    static constexpr int SW = UP2_ ? TW_ / 2 + 8 : TW_ * S_ + 8;
    static constexpr int SW_HALF_PAD() { return ((SW / 2 + 3) + 3) & ~3; }
    static constexpr int WIN4 = SW / 4;
    static constexpr int IHS = VEC_ ? (UP2_ ? TH / 2 + 2 : IH) : IH;          // staged rows
    // stride 2 (vector path): a staged row is DE-INTERLEAVED into an even-column and an odd-column plane of PE floats each, so a
    // work-item's ten taps are two aligned float4 + two scalars at a 16-byte lane stride (interleaved, the 32-byte lane stride made
    // the four SIMDs of a CU queue on 8-way LDS bank conflicts: measured 4x the FMA time)
    static constexpr int PE = (SW_HALF_PAD());
    static constexpr int IWP = VEC_ ? (S_ == 2 ? 2 * PE : SW) : ((IW + 3) & ~3);   // LDS row pitch (16-byte aligned rows)
    static constexpr int XOFF = VEC_ ? 3 : 0;     // LDS column of the tile's first halo pixel
    static constexpr int CK = (S == 2) ? 2 : 4;   // input channels staged per round
    static constexpr int NIN = 4 * S + KS - S;    // input floats per row a work-item consumes
    static constexpr int STAGE_FLOATS = CK * IHS * IWP;
    static constexpr int RED_FLOATS = 2 * COUT_T * (NT + 1);
    static constexpr int SMEM_FLOATS = STAGE_FLOATS > RED_FLOATS ? STAGE_FLOATS : RED_FLOATS;
};

template <class CFG>
__global__ __launch_bounds__(CFG::NT) void conv_tile_kernel(const mi_conv_params p, const int CoutPad,
                                                            const float* __restrict__ wts, const float* __restrict__ res_wts) {
    // wts / res_wts repeat p.w / p.res_w as `const __restrict__` kernel arguments: only then does the compiler prove the
    // (wave-uniform) weight reads invariant and issue them as scalar s_load instead of per-lane global loads
    constexpr int NT = CFG::NT, TW = CFG::TW, TH = CFG::TH, TXN = CFG::TXN, COUT_T = CFG::COUT_T;
    constexpr int KS = CFG::KS, S = CFG::S, IH = CFG::IH, IW = CFG::IW, IWP = CFG::IWP, CK = CFG::CK, NIN = CFG::NIN;
    constexpr bool UP2 = CFG::UP2, VEC = CFG::VEC;
    constexpr int XOFF = CFG::XOFF, IHS = CFG::IHS;

    __shared__ __attribute__((aligned(16))) float smem[CFG::SMEM_FLOATS];
    __shared__ __attribute__((aligned(16))) float4 chP[MI_MAX_CIN + 1];   // {A, B, -log2(e)A, -log2(e)B}; last entry = zeros
    __shared__ double chS[MI_MAX_CIN], chQ[MI_MAX_CIN];
    __shared__ float gMean[MI_MAX_GROUPS], gRstd[MI_MAX_GROUPS];

    const int tid = threadIdx.x;
    const int tiles_x = (p.W + TW - 1) / TW;
    const int tile = blockIdx.x;
    const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
    const int b = blockIdx.y;
    const int co0 = blockIdx.z * COUT_T;
    const int C0 = p.in0.C, C1 = p.in1.data ? p.in1.C : 0, Cin = C0 + C1;
    // real input extent, and the extent of the (virtual) image the taps address
    const int Hin = UP2 ? p.H / 2 : p.H * S, Win = UP2 ? p.W / 2 : p.W * S;
    const int Hv = UP2 ? p.H : Hin, Wv = UP2 ? p.W : Win;

    const int ty = tid / TXN, tx = tid % TXN;
    float acc[4][COUT_T];
#pragma unroll
    for (int px = 0; px < 4; ++px)
#pragma unroll
        for (int co = 0; co < COUT_T; ++co) acc[px][co] = 0.0f;

    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;   // pad = 1 for every member of the family
    const int b0 = mi_row_of(b, p.in0.bmod), b1 = mi_row_of(b, p.in1.bmod);
    const int br0 = mi_row_of(b, p.res0.bmod), br1 = mi_row_of(b, p.res1.bmod);
    constexpr bool gn = CFG::GN;         // GroupNorm + SiLU prologue compiled in or out (no per-element branches)
    MI_TSTART();
    // per-channel totals of the producer's per-tile partial sums: wave 0 only, first thing (see common.hip.h)
    if constexpr (CFG::GN) {
        mi_gn_channel_totals(p.in0, p.in1, C0, Cin, b, tid, NT, chS, chQ);       // all waves: one round trip instead of nt / TPC
    }

    // Staging is split (T14): a round's global loads are issued back-to-back into registers with clamped
    // (always legal) addresses and no branches; the activation + LDS write happens after the barrier, and on
    // the vector path the NEXT round's loads fly under the FMA loop.
    //   VEC  (k3 s1, W % 4 == 0): aligned float4 loads of the window [ox0-4, ox0+TW+4)
    //   !VEC (stride 2, nearest-upsample, ragged W): scalar loads, U at a time
    constexpr int WIN4 = CFG::WIN4;
    constexpr int PER4 = (CK * IHS * WIN4 + NT - 1) / NT;
    const int sy0 = UP2 ? oy0 / 2 - 1 : iy0, sx0 = UP2 ? ox0 / 2 - 4 : ox0 * S - 4;      // source coordinates of the staged window
    float4 xq4[VEC ? PER4 : 1];
    // per-work-item staging slots (tile geometry only -> identical for every channel round):
    //   msrc = element offset of the float4 from the first channel plane of a round, or -1 when the slot is outside the image
    //   mdst = LDS index of the float4 (16-byte aligned), or -1 for an unused slot; mck = channel within the round
    int msrc[VEC ? PER4 : 1], mdst[VEC ? PER4 : 1], mck[VEC ? PER4 : 1];
    if constexpr (VEC) {
#pragma unroll
        for (int u = 0; u < PER4; ++u) {
            const int q = tid + u * NT;
            const int rowid = q / WIN4, xq = q % WIN4;
            const int ck = rowid / IHS, iy = rowid % IHS;
            const int gy = sy0 + iy, gx0 = sx0 + 4 * xq;
            const bool in = (ck < CK) && gy >= 0 && gy < Hin && gx0 >= 0 && gx0 < Win;
            msrc[u] = in ? ck * Hin * Win + gy * Win + gx0 : -1;       // relative to the round's first channel plane
            mdst[u] = (ck < CK) ? (ck * IHS + iy) * IWP + (S == 2 ? 2 : 4) * xq : -1;      // stride 2: index into the even plane
            mck[u] = ck;
        }
    }
    auto stage_load = [&](int c0) {
        if constexpr (VEC) {
            // the vector path requires Cin % CK == 0 and C0 % CK == 0: a round never straddles the two concatenated inputs, so
            // its base pointer is wave-uniform (SGPR) and every load is base + a per-slot offset computed once per tile
            const bool second = c0 >= C0;
            const float* base = second ? p.in1.data + (size_t)(b1 * C1 + (c0 - C0)) * Hin * Win : p.in0.data + (size_t)(b0 * C0 + c0) * Hin * Win;
#pragma unroll
            for (int u = 0; u < PER4; ++u) {
                const unsigned off = msrc[u] >= 0 ? (unsigned)msrc[u] : 0u;     // unused / outside slots read element 0 (legal, ignored)
                xq4[u] = mi_ldg4(base + off);
            }
        }
    };
    auto stage_write = [&](int c0) {
        if constexpr (VEC) {
#pragma unroll
            for (int u = 0; u < PER4; ++u) {
                // slots outside the image use the all-zero parameter entry: 0*x+0 -> SiLU(0) = 0 = the zero padding,
                // which (as in the reference) follows the activation
                const float4 P = chP[msrc[u] >= 0 ? c0 + mck[u] : MI_MAX_CIN];
                const float xe[4] = {xq4[u].x, xq4[u].y, xq4[u].z, xq4[u].w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (gn) {
                        const float v = fmaf(xe[e], P.x, P.y);
                        const float ex = __builtin_amdgcn_exp2f(fmaf(xe[e], P.z, P.w));     // exp(-v)
                        o[e] = v * __builtin_amdgcn_rcpf(1.0f + ex);
                    } else {
                        o[e] = xe[e] * P.x;
                    }
                }
                if constexpr (S == 2) {
                    if (mdst[u] >= 0) {      // staged column c = 4xq + e: even e -> even plane at c/2 + 2, odd e -> odd plane at (c-1)/2 + 3
                        *reinterpret_cast<float2*>(&smem[mdst[u] + 2]) = make_float2(o[0], o[2]);
                        smem[mdst[u] + CFG::PE + 3] = o[1];
                        smem[mdst[u] + CFG::PE + 4] = o[3];
                    }
                } else {
                    if (mdst[u] >= 0) *reinterpret_cast<float4*>(&smem[mdst[u]]) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        } else {
            constexpr int TOT = CK * IH * IW, U = 8;
#pragma unroll 1
            for (int it = 0; it < TOT; it += U * NT) {
                float xs[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = it + tid + u * NT;
                    const int ix = idx % IW, r = idx / IW;
                    const int iy = r % IH, ck = r / IH;
                    const int c = c0 + ck;
                    const int gy = iy0 + iy, gx = ix0 + ix;
                    const bool inimg = idx < TOT && c < Cin && gy >= 0 && gy < Hv && gx >= 0 && gx < Wv;
                    const int sy = UP2 ? (gy >> 1) : gy, sx = UP2 ? (gx >> 1) : gx;
                    const bool second = inimg && c >= C0;
                    const float* base = second ? p.in1.data : p.in0.data;
                    const int cc = second ? (b1 * C1 + (c - C0)) : (b0 * C0 + c);
                    const unsigned off = inimg ? (unsigned)((cc * Hin + sy) * Win + sx) : 0u;
                    xs[u] = mi_ldg(base + off);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = it + tid + u * NT;
                    const int ix = idx % IW, r = idx / IW;
                    const int iy = r % IH, ck = r / IH;
                    const int c = c0 + ck;
                    const int gy = iy0 + iy, gx = ix0 + ix;
                    const bool inimg = idx < TOT && c < Cin && gy >= 0 && gy < Hv && gx >= 0 && gx < Wv;
                    float v = 0.0f;
                    if (inimg) { const float4 P = chP[c]; v = gn ? mi_silu(fmaf(xs[u], P.x, P.y)) : xs[u] * P.x; }
                    if (idx < TOT) smem[(ck * IH + iy) * IWP + ix] = v;
                }
            }
        }
    };

    MI_TPHASE(0);       // stats totals (wave 0) + geometry
    stage_load(0);      // first round's loads fly under the statistics prologue
    MI_TPHASE(1);

    // ---------------- prologue: per-channel affine for the fused GroupNorm / scale-shift
    if constexpr (CFG::GN) {
        __syncthreads();
        const int cpg = Cin / p.gn_groups;
        for (int g = tid; g < p.gn_groups; g += NT)
            mi_gn_group_moments(chS, chQ, g * cpg, (g + 1) * cpg, (double)cpg * (double)Hin * (double)Win, p.gn_eps, gMean[g], gRstd[g]);
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
            if (c >= C0) A *= p.in1.scale; else A *= p.in0.scale;
            chP[c] = make_float4(A, Bc, A * -1.44269504088896340736f, Bc * -1.44269504088896340736f);
        }
    } else {
        for (int c = tid; c < Cin; c += NT) chP[c] = make_float4((c >= C0) ? p.in1.scale : p.in0.scale, 0.0f, 0.0f, 0.0f);
    }
    if (tid == 0) chP[MI_MAX_CIN] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);

    for (int c0 = 0; c0 < Cin; c0 += CK) {
        MI_TPHASE(c0 == 0 ? 2 : 5);     // rest of the prologue | previous round's FMA loop
        __syncthreads();   // previous round fully consumed (and chA/chB visible on the first round)
        MI_TPHASE(3);
        stage_write(c0);
        MI_TPHASE(4);
        __syncthreads();
        MI_TPHASE(3);
        if (c0 + CK < Cin) stage_load(c0 + CK);       // in flight during the FMA loop below
        MI_TPHASE(1);
        const int nck = (Cin - c0) < CK ? (Cin - c0) : CK;
        for (int ck = 0; ck < nck; ++ck) {
            const float* wc = wts + (size_t)(c0 + ck) * KS * KS * CoutPad + co0;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                float in[NIN];
                if constexpr (VEC && UP2) {
                    // virtual column x0+j-1 is source column (x0+j-1)>>1: six taps share four source pixels
                    const float* row = &smem[(ck * IHS + ((ty + ky - 1) >> 1) + 1) * IWP + 2 * tx + 3];
                    const float s0 = row[0];
                    const float2 s12 = *reinterpret_cast<const float2*>(row + 1);     // 8-byte aligned: column 2tx+4
                    const float s3 = row[3];
                    in[0] = s0; in[1] = s12.x; in[2] = s12.x; in[3] = s12.y; in[4] = s12.y; in[5] = s3;
                } else if constexpr (VEC && S == 2) {
                    // taps 2x0-1 .. 2x0+8 of the four outputs x0 .. x0+3: odd columns O[4tx+1..4tx+5], even columns E[4tx+2..4tx+6]
                    const float* rowE = &smem[(ck * IHS + ty * 2 + ky) * IWP + 4 * tx + 4];
                    const float* rowO = rowE + CFG::PE;
                    const float4 e4 = *reinterpret_cast<const float4*>(rowE), o4 = *reinterpret_cast<const float4*>(rowO);
                    in[0] = o4.x; in[1] = e4.x; in[2] = o4.y; in[3] = e4.y; in[4] = o4.z; in[5] = e4.z; in[6] = o4.w; in[7] = e4.w;
                    in[8] = rowO[4];
                    in[9] = rowE[4];
                } else if constexpr (VEC) {
                    const float* row = &smem[(ck * IHS + ty + ky) * IWP + tx * 4 + XOFF];
                    in[0] = row[0];
                    const float4 m4 = *reinterpret_cast<const float4*>(row + 1);      // 16-byte aligned: column 4tx+4
                    in[1] = m4.x; in[2] = m4.y; in[3] = m4.z; in[4] = m4.w;
                    in[5] = row[5];
                } else {
                    const float* row = &smem[(ck * IH + ty * S + ky) * IWP + tx * 4 * S];
#pragma unroll
                    for (int j = 0; j < NIN; ++j) in[j] = row[j];
                }
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const float* wk = wc + (ky * KS + kx) * CoutPad;
                    float wv[COUT_T];
#pragma unroll
                    for (int co = 0; co < COUT_T; ++co) wv[co] = wk[co];
#pragma unroll
                    for (int px = 0; px < 4; ++px)
#pragma unroll
                        for (int co = 0; co < COUT_T; ++co) acc[px][co] = fmaf(in[px * S + kx], wv[co], acc[px][co]);
                }
            }
        }
    }

    MI_TPHASE(5);       // last round's FMA loop
    // ---------------- epilogue: bias, residual, store, statistics of the output
    const int oy = oy0 + ty, ox = ox0 + tx * 4;
    const bool row_ok = oy < p.H;
    const int Cres0 = p.res0.data ? p.res0.C : 0, Cres1 = (p.res0.data && p.res1.data) ? p.res1.C : 0;
#pragma unroll
    for (int co = 0; co < COUT_T; ++co) {
        const float bv = (p.bias && (co0 + co) < p.Cout) ? p.bias[co0 + co] : 0.0f;
#pragma unroll
        for (int px = 0; px < 4; ++px) acc[px][co] += bv;
    }
    if (p.res0.data && row_ok) {
        if (p.res_w) {
            for (int cr = 0; cr < Cres0 + Cres1; ++cr) {
                const bool second = cr >= Cres0;
                const float* src = second ? p.res1.data + ((size_t)(br1 * Cres1 + (cr - Cres0)) * p.H + oy) * p.W
                                          : p.res0.data + ((size_t)(br0 * Cres0 + cr) * p.H + oy) * p.W;
                const float sc = second ? p.res1.scale : p.res0.scale;
                float rv[4];
                if (ox + 3 < p.W && (p.W & 3) == 0) {
                    const float4 t4 = mi_ldg4(src + ox);
                    rv[0] = t4.x * sc; rv[1] = t4.y * sc; rv[2] = t4.z * sc; rv[3] = t4.w * sc;
                } else {
#pragma unroll
                    for (int px = 0; px < 4; ++px) rv[px] = (ox + px < p.W) ? mi_ldg(src + ox + px) * sc : 0.0f;
                }
                const float* wr = res_wts + (size_t)cr * CoutPad + co0;
#pragma unroll
                for (int co = 0; co < COUT_T; ++co) {
                    const float wv = wr[co];
#pragma unroll
                    for (int px = 0; px < 4; ++px) acc[px][co] = fmaf(rv[px], wv, acc[px][co]);
                }
            }
#pragma unroll
            for (int co = 0; co < COUT_T; ++co) {
                const float bv = (p.res_b && (co0 + co) < p.Cout) ? p.res_b[co0 + co] : 0.0f;
#pragma unroll
                for (int px = 0; px < 4; ++px) acc[px][co] += bv;
            }
        } else {
#pragma unroll
            for (int co = 0; co < COUT_T; ++co) {
                if (co0 + co < p.Cout) {
                    const float* src = p.res0.data + ((size_t)(br0 * Cres0 + co0 + co) * p.H + oy) * p.W;
                    if (ox + 3 < p.W && (p.W & 3) == 0) {
                        const float4 t4 = mi_ldg4(src + ox);
                        acc[0][co] += t4.x * p.res0.scale; acc[1][co] += t4.y * p.res0.scale;
                        acc[2][co] += t4.z * p.res0.scale; acc[3][co] += t4.w * p.res0.scale;
                    } else {
#pragma unroll
                        for (int px = 0; px < 4; ++px)
                            if (ox + px < p.W) acc[px][co] += mi_ldg(src + ox + px) * p.res0.scale;
                    }
                }
            }
        }
    }
    // per-channel (sum, sum of squares) of this work-item's four pixels, in fp64 from the start (common.hip.h: why the statistics are fp64)
    double ssum[COUT_T], ssq[COUT_T];
#pragma unroll
    for (int co = 0; co < COUT_T; ++co) {
        double s = 0.0, q = 0.0;
        if (row_ok && (co0 + co) < p.Cout) {
            float* dst = p.out + ((size_t)(b * p.Cout + co0 + co) * p.H + oy) * p.W + ox;
            if (ox + 3 < p.W && (p.W & 3) == 0) {
                mi_stg4(dst, make_float4(acc[0][co], acc[1][co], acc[2][co], acc[3][co]));
#pragma unroll
                for (int px = 0; px < 4; ++px) { const double v = (double)acc[px][co]; s += v; q = fma(v, v, q); }
            } else {
#pragma unroll
                for (int px = 0; px < 4; ++px)
                    if (ox + px < p.W) { mi_stg(dst + px, acc[px][co]); const double v = (double)acc[px][co]; s += v; q = fma(v, v, q); }
            }
        }
        ssum[co] = s;
        ssq[co] = q;
    }
    if (p.out_stats) {
        static_assert(NT % 64 == 0 && 2 * COUT_T <= NT, "statistics reduction: whole waves");
        constexpr int NW = NT / 64;
        static_assert(2 * COUT_T * NW * 2 <= CFG::SMEM_FLOATS, "statistics staging fits the tile buffer");
        __syncthreads();   // staging buffer no longer read
        double* const redd = reinterpret_cast<double*>(smem);          // [2 * COUT_T][NW]
#pragma unroll
        for (int co = 0; co < COUT_T; ++co) {
            double s = ssum[co], q = ssq[co];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
            if ((tid & 63) == 0) { redd[(2 * co) * NW + (tid >> 6)] = s; redd[(2 * co + 1) * NW + (tid >> 6)] = q; }
        }
        __syncthreads();
        if (tid < 2 * COUT_T) {
            double a = 0.0;
            for (int w = 0; w < NW; ++w) a += redd[tid * NW + w];
            const int co = co0 + (tid >> 1);
            if (co < p.Cout) {
                const int nt = gridDim.x;
                p.out_stats[((size_t)(b * p.Cout + co) * nt + tile) * 2 + (tid & 1)] = a;
            }
        }
    }
    MI_TPHASE(6);
    MI_TEND();
}

template <int NT, int TW, int COUT_T, int KS, int S, bool UP2, bool VEC, bool GN>
int launch_conv_g(const mi_conv_params& p, hipStream_t st) {
    using CFG = ConvCfg<NT, TW, COUT_T, KS, S, UP2, VEC, GN>;
    const int tiles = ((p.H + CFG::TH - 1) / CFG::TH) * ((p.W + TW - 1) / TW);
    const int cz = (p.Cout + COUT_T - 1) / COUT_T;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_tile_kernel<CFG>), dim3(tiles, p.B, cz), dim3(NT), 0, st, p, cz * COUT_T, p.w, p.res_w);
    return mi_check_launch("conv_tile_kernel");
}

template <int NT, int TW, int COUT_T, int KS, int S, bool UP2, bool VEC>
int launch_conv_v(const mi_conv_params& p, hipStream_t st) {
    if constexpr (KS == 3 && S == 1 && !UP2) {          // only the Block convs (layers.py:131-145) carry a GroupNorm prologue
        if (p.gn_groups > 0) return launch_conv_g<NT, TW, COUT_T, KS, S, UP2, VEC, true>(p, st);
    } else {
        if (p.gn_groups > 0) { mi_set_error("mi_conv_fwd: GroupNorm prologue is only built for the k3 s1 family"); return MI_ERR_UNSUPPORTED; }
    }
    return launch_conv_g<NT, TW, COUT_T, KS, S, UP2, VEC, false>(p, st);
}

template <int NT, int TW, int COUT_T, int KS, int S, bool UP2>
int launch_conv(const mi_conv_params& p, hipStream_t st) {
    // vector staging needs float4-aligned source rows and whole channel rounds; otherwise the scalar staging handles it
    const int Cin = p.in0.C + (p.in1.data ? p.in1.C : 0);
    const int Win = UP2 ? p.W / 2 : p.W * S;
    constexpr int CKv = (S == 2) ? 2 : 4;
    if ((Win & 3) == 0 && (Cin % CKv) == 0 && (p.in0.C % CKv) == 0 &&
        (size_t)p.B * (p.in0.C > p.in1.C ? p.in0.C : p.in1.C) * p.H * p.W * S * S < (1ull << 31))
        return launch_conv_v<NT, TW, COUT_T, KS, S, UP2, true>(p, st);
    return launch_conv_v<NT, TW, COUT_T, KS, S, UP2, false>(p, st);
}

template <int COUT_T, int KS, int S, bool UP2>
int dispatch_tile(const mi_conv_params& p, hipStream_t st) {
    switch (p.tile_cfg & 0xff) {
        case 0: return launch_conv<256, 64, COUT_T, KS, S, UP2>(p, st);
        case 1: return launch_conv<256, 32, COUT_T, KS, S, UP2>(p, st);
        case 2: return launch_conv<64, 32, COUT_T, KS, S, UP2>(p, st);
    }
    mi_set_error("mi_conv_fwd: bad tile_cfg %d", p.tile_cfg);
    return MI_ERR_INVALID;
}

}  // namespace

extern "C" int mi_conv_tile_shape(int tile_cfg, int* th, int* tw) {
    switch (tile_cfg & 0xff) {
        case 6: *th = 8; *tw = 64; return MI_OK;      // row-paired matrix-core path (conv_rp.hip)
        case 7: *th = 8; *tw = 32; return MI_OK;
        case 10: *th = 16; *tw = 16; return MI_OK;    // (wide k3 s1 member only)
        case 11: *th = 8; *tw = 16; return MI_OK;     // wide GEMM kernel (conv_wide.hip)
        // (12: full-width stripes, conv_stripe.hip -- the statistics blocks are mi_conv_stripe_rows(p) rows x the image width, not a fixed tile)
        case 0: *th = 16; *tw = 64; return MI_OK;
        case 1: *th = 32; *tw = 32; return MI_OK;
        case 2: *th = 8; *tw = 32; return MI_OK;
        case 8: *th = 32; *tw = 64; return MI_OK;     // matrix-core CrossEmbed (crossembed.hip)
        case 9: *th = 16; *tw = 32; return MI_OK;
    }
    return MI_ERR_INVALID;
}

extern "C" int mi_conv_cout_tile(int Cout) { return Cout <= 4 ? 4 : (Cout % 16 == 0 ? 16 : 8); }

extern "C" int mi_conv_fwd(const mi_conv_params* pp, void* stream) {
    const mi_conv_params& p = *pp;
    hipStream_t st = (hipStream_t)stream;
    const int Cin = p.in0.C + (p.in1.data ? p.in1.C : 0);
    const bool any16 = p.in0.st || (p.in1.data && p.in1.st) || (p.res0.data && p.res0.st) || (p.res1.data && p.res1.st) || p.out_st;
    if (any16 && !(p.w_rp && (p.tile_cfg & 0x400))) { mi_set_error("mi_conv_fwd: bf16 activation storage is read / written by the single-term row-paired kernels only (w_rp, tile_cfg | 0x400)"); return MI_ERR_UNSUPPORTED; }
    if (p.w_rp) {           // row-paired matrix-core family (narrow and wide regimes): its own limits
        if (p.gn_groups > MI_MAX_GROUPS || (p.gn_groups > 0 && (Cin % p.gn_groups) != 0)) { mi_set_error("mi_conv_fwd: Cin %d / groups %d", Cin, p.gn_groups); return MI_ERR_INVALID; }
        if (p.gn_groups > 0 && (!p.in0.stats || (p.in1.data && !p.in1.stats))) { mi_set_error("mi_conv_fwd: GroupNorm input without channel statistics"); return MI_ERR_INVALID; }
        if (p.res0.data && !p.res_w && p.res0.C != p.Cout) { mi_set_error("mi_conv_fwd: identity residual needs Cres == Cout"); return MI_ERR_INVALID; }
        if (p.B <= 0 || p.H <= 0 || p.W <= 0 || p.Cout <= 0) { mi_set_error("mi_conv_fwd: empty problem"); return MI_ERR_INVALID; }
        if ((p.tile_cfg & 0xff) == 11) return mi_conv_wide_launch(p, st);
        if ((p.tile_cfg & 0xff) == 12) return mi_conv_stripe_launch(p, st);
        return mi_conv_rp_launch(p, st);
    }
    if (Cin > MI_MAX_CIN || p.gn_groups > MI_MAX_GROUPS) { mi_set_error("mi_conv_fwd: Cin %d / groups %d too large for the direct-conv family", Cin, p.gn_groups); return MI_ERR_UNSUPPORTED; }
    if (p.gn_groups > 0 && (Cin % p.gn_groups) != 0) { mi_set_error("mi_conv_fwd: Cin %d not divisible by groups %d", Cin, p.gn_groups); return MI_ERR_INVALID; }
    if (p.gn_groups > 0 && (!p.in0.stats || (p.in1.data && !p.in1.stats))) { mi_set_error("mi_conv_fwd: GroupNorm input without channel statistics"); return MI_ERR_INVALID; }
    if (p.res0.data && !p.res_w && p.res0.C != p.Cout) { mi_set_error("mi_conv_fwd: identity residual needs Cres == Cout"); return MI_ERR_INVALID; }
    if (p.B <= 0 || p.H <= 0 || p.W <= 0 || p.Cout <= 0) { mi_set_error("mi_conv_fwd: empty problem"); return MI_ERR_INVALID; }
    if (p.up2 && ((p.H | p.W) & 1)) { mi_set_error("mi_conv_fwd: up2 needs even output size"); return MI_ERR_INVALID; }
    const int ct = mi_conv_cout_tile(p.Cout);
    // MI_CONV_SPLIT8: 8 output channels as two 4-channel workgroups -- twice the waves for the small (latency-bound) launches
    const bool split8 = (p.tile_cfg & MI_CONV_SPLIT8) && ct == 8 && p.Cout == 8;
    if (p.ksize == 3 && p.stride == 1 && !p.up2) {
        if (ct == 4 || split8) return dispatch_tile<4, 3, 1, false>(p, st);
        if (ct == 8 || (p.tile_cfg & MI_CONV_SPLIT16)) return dispatch_tile<8, 3, 1, false>(p, st);
        return dispatch_tile<16, 3, 1, false>(p, st);
    }
    if (p.ksize == 3 && p.stride == 1 && p.up2) {
        if (ct == 16) return dispatch_tile<16, 3, 1, true>(p, st);
        if (ct == 8 && !split8) return dispatch_tile<8, 3, 1, true>(p, st);
        return dispatch_tile<4, 3, 1, true>(p, st);
    }
    if (p.ksize == 4 && p.stride == 2 && !p.up2) {
        if (ct == 16) return dispatch_tile<8, 4, 2, false>(p, st);     // 16 taps x 16 channels would not fit the SGPR file: 2 channel tiles
        if (ct == 8 && !split8) return dispatch_tile<8, 4, 2, false>(p, st);
        return dispatch_tile<4, 4, 2, false>(p, st);
    }
    mi_set_error("mi_conv_fwd: unsupported conv k%d s%d up%d", p.ksize, p.stride, p.up2);
    return MI_ERR_UNSUPPORTED;
}
