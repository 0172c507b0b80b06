// Implicit-GEMM 3x3 stride-1 convolution for the WIDE presets of MinImagen's U-Net (Unet() default, Base, Super: 128 .. 1024+ channels):
// the convs of Block / ResnetBlock (layers.py:131-145, 415-439) with the fused GroupNorm -> scale/shift -> SiLU prologue, the identity or
// 1x1-conv residual and the next GroupNorm's partial statistics -- the same operator as conv_rp.hip's wide regime (tile_cfg 11 selects this file).
//
// Why a second kernel: conv_rp.hip is built for 8..32 channels (N = output-row parity x 8 output channels, the activation transform inside the
// conv kernel).  At 128+ channels every workgroup of a layer (one per 32 output channels) repeated the GroupNorm / SiLU / fp16-split transform
// of the same input window, a wave re-read every B fragment from LDS for two pixel groups only, and both phases alternated between barriers:
// 145-195 TFLOP/s algorithmic (profiles/r05_wide_conv_phase_trace.txt).  Here
//   * conv_prep_kernel transforms the conv input ONCE per layer into the matrix operand itself: fp16 hi / lo planes, pixel-major 16-byte
//     chunks of 8 channels, zero halo included ([B][octet][hi | lo][H' + 2][W' + 2] chunks, H' / W' rounded up to the 8 x 16 tile);
//   * conv_wide_kernel is a plain GEMM loop: M = 128 pixels (8 x 16) x N = 128 output channels per workgroup, K = 32 input channels x one tap per
//     step; the activation window of a 32-channel group and the weights of a step arrive by LDS-DMA into double buffers (no registers, no
//     transform, one barrier per step); each wave owns 64 pixels x 64 channels = 4 x 4 accumulator fragments, so an LDS fragment read feeds
//     twelve matrix instructions.
// Arithmetic as in conv_rp.hip: x = hi + lo (fp16), products hi*hi + lo*hi + hi*lo on v_mfma_f32_16x16x32_f16, fp32 accumulation, operands
// brought into the fp16 range by exact powers of two (weights at pack time, activations per image by mi_gn_coef_fwd) and undone in the epilogue.
#include "common.hip.h"
#include <type_traits>
#include <cstdlib>

namespace {

typedef _Float16 cw_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 cw_f16x2 __attribute__((ext_vector_type(2)));
typedef float cw_f32x2 __attribute__((ext_vector_type(2)));

constexpr int CW_TH = 8, CW_TW = 16, CW_IH = CW_TH + 2, CW_UW = CW_TW + 2;
// one 32-channel window in LDS: [octet 4][hi | lo][row][column] 16-byte chunks, the octets CW_OCT = 368 chunks apart (360 used).  The stride is a
// multiple of 16 chunks on purpose: a ds_read_b128 is served in lane groups like {0-3, 12-15, 20-27} -- lanes of two neighbouring octets (lane / 16) in
// one group --, so with the octets 360 chunks apart (= 8 mod 16) every A read was a 2-way bank conflict (33-40 % of the LDS cycles, PMC)
constexpr int CW_OCT = 368, CW_OCT_USED = 2 * CW_IH * CW_UW;
constexpr int CW_WIN_INSTR = 4 * CW_OCT / 64;        // copied by 23 wave-wide LDS-DMA instructions (6 per wave, 5 for the last)
// NF: 16-channel N fragments per wave (4: 128 output channels per workgroup; 2: 64 -- for launches that would otherwise leave CUs with one workgroup)

__device__ __forceinline__ void cw_split8(const float (&y)[8], uint4& hi, uint4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const cw_f32x2 v = {y[2 * i], y[2 * i + 1]};
        const cw_f16x2 h2 = __builtin_convertvector(v, cw_f16x2);
        h[i] = __builtin_bit_cast(unsigned, h2);
        l[i] = mi_split_lo2(h[i], y[2 * i], y[2 * i + 1]);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

__host__ __device__ __forceinline__ int cw_hp(int H) { return CW_TH * ((H + CW_TH - 1) / CW_TH) + 2; }
__host__ __device__ __forceinline__ int cw_wp(int W) { return CW_TW * ((W + CW_TW - 1) / CW_TW) + 2; }

// ---- operand preparation: one work-item per padded pixel and channel octet (conv input octets first, then the 1x1-residual input's)
__global__ __launch_bounds__(256) void conv_prep_kernel(const mi_conv_params p, uint4* __restrict__ prep) {
    const int H = p.H, W = p.W, Hp = cw_hp(H), Wp = cw_wp(W), plane = Hp * Wp;
    const int b = blockIdx.z, o = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    if (q >= plane) return;
    const int yy = q / Wp, xx = q - yy * Wp, y = yy - 1, x = xx - 1;      // output-resolution pixel; with up2 (nearest x2 in front of the conv,
    const int Hs = p.up2 ? H / 2 : H, Ws = p.up2 ? W / 2 : W;             // layers.py:512-515) the planes are written up-sampled: source (y >> 1, x >> 1)
    const int C0 = p.in0.C, C1 = p.in1.data ? p.in1.C : 0, Cin = C0 + C1;
    const int Cr0 = (p.res0.data && p.res_w_rp) ? p.res0.C : 0, Cr1 = (Cr0 && p.res1.data) ? p.res1.C : 0;
    const int KO = Cin >> 3, NO = KO + ((Cr0 + Cr1) >> 3);
    uint4 hv = make_uint4(0u, 0u, 0u, 0u), lv = hv;               // the zero padding follows the activation (as in the reference)
    if (y >= 0 && y < H && x >= 0 && x < W) {
        const bool isres = o >= KO;
        const int c0 = 8 * (isres ? o - KO : o);
        const int Ca = isres ? Cr0 : C0, Cb = isres ? Cr1 : C1;
        const bool second = c0 >= Ca;
        const mi_act& t = isres ? (second ? p.res1 : p.res0) : (second ? p.in1 : p.in0);
        const int bb = mi_row_of(b, t.bmod);
        const size_t HW = (size_t)Hs * Ws;
        const float* src = t.data + ((size_t)bb * (second ? Cb : Ca) + (second ? c0 - Ca : c0)) * HW + (p.up2 ? (size_t)(y >> 1) * Ws + (x >> 1) : (size_t)y * Ws + x);
        float v[8];
        if (isres) {
            const float rsc = ldexpf(t.scale, p.gn_exps[2 * b + 1]);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = mi_ldg(src + j * HW) * rsc;
        } else {
            const float4* cf = reinterpret_cast<const float4*>(p.gn_coef) + (size_t)b * Cin + c0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = mi_ldg(src + j * HW);
                const float4 P = cf[j];
                if (p.gn_groups > 0) {
                    const float a = fmaf(xv, P.x, P.y);
                    const float ex = __builtin_amdgcn_exp2f(fmaf(xv, P.z, P.w));       // exp(-a)
                    v[j] = a * __builtin_amdgcn_rcpf(1.0f + ex);
                } else {
                    v[j] = xv * P.x;
                }
            }
        }
        cw_split8(v, hv, lv);
    }
    uint4* dst = prep + ((size_t)b * NO + o) * 2 * plane + q;
    dst[0] = hv;
    dst[plane] = lv;
}

// ---- the GEMM loop
template <int NF>
__global__ __launch_bounds__(256, 2) void conv_wide_kernel(const mi_conv_params p, const uint4* __restrict__ wf, const uint4* __restrict__ rwf,
                                                           const uint4* __restrict__ prep) {
    constexpr int CW_NT = 2 * NF;                          // N tiles per workgroup
    constexpr int CW_BCH = CW_NT * 2 * 64;                 // chunks of one step's B fragments: [N tile][hi | lo][lane] (16 / 8 KB)
    __shared__ __attribute__((aligned(16))) uint4 aw[2][CW_WIN_INSTR * 64];
    __shared__ __attribute__((aligned(16))) uint4 bw[2][CW_BCH];
    // NF = 4: 2 x 23 552 + 2 x 16 384 = 79 872 bytes: two workgroups per CU (the epilogue's partial statistics reuse bw)
    static_assert(CW_OCT % 16 == 0 && CW_OCT >= CW_OCT_USED && (4 * CW_OCT) % 64 == 0, "window layout");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;               // the wave's 4 rows / NF N tiles of the workgroup's 8 x 2 NF
    const int H = p.H, W = p.W, Hp = cw_hp(H), Wp = cw_wp(W), plane = Hp * Wp;
    const int tiles_x = (W + CW_TW - 1) / CW_TW, tiles = tiles_x * ((H + CW_TH - 1) / CW_TH);
    // XCD-aware placement as in conv_rp.hip: workgroup L runs on XCD L % 8; whole images per XCD
    int b, tile;
    if ((p.B & 7) == 0) {
        const int L = blockIdx.x, k = L >> 3;
        b = (L & 7) + 8 * (k / tiles);
        tile = k % tiles;
    } else {
        b = blockIdx.x / tiles;
        tile = blockIdx.x % tiles;
    }
    const int ty0 = (tile / tiles_x) * CW_TH, tx0 = (tile % tiles_x) * CW_TW;
    const int C0 = p.in0.C, C1 = p.in1.data ? p.in1.C : 0, Cin = C0 + C1;
    const int Cr0 = (p.res0.data && rwf) ? p.res0.C : 0, Cr1 = (Cr0 && p.res1.data) ? p.res1.C : 0, Cres = Cr0 + Cr1;
    const int G = Cin >> 5, GR = Cres >> 5, GT = G + GR, NO = (Cin + Cres) >> 3;
    const int njt = p.Cout >> 4, nt0 = blockIdx.y * CW_NT;

    // window copy: lane -> (octet, plane, row, column) of the linear LDS index; the same for every channel group
    unsigned aoff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        int q = (wave * 6 + i) * 64 + lane;                 // (instruction 23 does not exist: wave 3 copies five)
        q = q < 4 * CW_OCT ? q : 4 * CW_OCT - 1;
        const int oc = q / CW_OCT;
        int r1 = q - oc * CW_OCT;
        r1 = r1 < CW_OCT_USED ? r1 : CW_OCT_USED - 1;       // (the 8 padding lanes of an octet re-fetch its last chunk)
        const int pl = r1 / (CW_IH * CW_UW), r2 = r1 - pl * (CW_IH * CW_UW);
        const int iy = r2 / CW_UW, c = r2 - iy * CW_UW;
        aoff[i] = (unsigned)(((oc * 2 + pl) * Hp + ty0 + iy) * Wp + tx0 + c);
    }
    const uint4* const pimg = prep + (size_t)b * NO * 2 * plane;
    auto issue_a = [&](int g, int buf) {
        const uint4* src = pimg + (size_t)g * 8 * plane;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i == 5 && wave == 3) break;
            const uint4* gp = src + aoff[i];
            uint4* l = &aw[buf][(wave * 6 + i) * 64];
#if defined(HIPEMU)
            l[lane] = *gp;
#else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
#endif
        }
    };
    // B fragments of step (g, tap): [group][tap][N tile of the layer][hi | lo][lane]; the workgroup's 8 N tiles are one contiguous 16 KB block
    auto issue_b = [&](int g, int tap, int buf) {
        const uint4* src = g < G ? wf + (((size_t)g * 9 + tap) * njt + nt0) * 128 : rwf + ((size_t)(g - G) * njt + nt0) * 128;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const uint4* gp = src + (wave * NF + i) * 64 + lane;
            uint4* l = &bw[buf][(wave * NF + i) * 64];
#if defined(HIPEMU)
            l[lane] = *gp;
#else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
#endif
        }
    };

    f32x4 acc[4][NF];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue_a(0, 0);
    issue_b(0, 0, 0);
    int kb = 0;                                            // B buffer of the step about to run
    // one step: wait for its operands, let everybody leave the previous step, start the next copies, multiply.  The first step of a group
    // requests [B of the next step][window of the next group], in this order: the following step has to wait for the B copy only and
    // lets the six window instructions fly on (vmcnt counts in order); every other step waits for everything it has in flight.
    auto step = [&](const int g, auto tap_tag, const bool first, const int ng, const int ntap) {
        constexpr int tap = decltype(tap_tag)::value;
#if !defined(HIPEMU)
        if (tap == 1 && g + 1 < GT) {                       // (tap 1 only occurs in conv groups)
            if (wave == 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#endif
        __syncthreads();
        if (ng < GT) issue_b(ng, ntap, kb ^ 1);
        if (first && g + 1 < GT) issue_a(g + 1, (g + 1) & 1);
        const uint4* const ab = aw[g & 1];
        const uint4* const bb = bw[kb];
        constexpr int dy = tap / 3, dx = tap % 3;
        cw_f16x8 bh[NF], bl[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            bh[nf] = __builtin_bit_cast(cw_f16x8, bb[((wn * NF + nf) * 2) * 64 + lane]);
            bl[nf] = __builtin_bit_cast(cw_f16x8, bb[((wn * NF + nf) * 2 + 1) * 64 + lane]);
        }
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int idx = lg * CW_OCT + (wm * 4 + mf + dy) * CW_UW + dx + lq;
            const cw_f16x8 ah = __builtin_bit_cast(cw_f16x8, ab[idx]), al = __builtin_bit_cast(cw_f16x8, ab[idx + CW_IH * CW_UW]);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[nf], acc[mf][nf], 0, 0, 0);
                acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[nf], acc[mf][nf], 0, 0, 0);
                acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[nf], acc[mf][nf], 0, 0, 0);
            }
        }
        kb ^= 1;
    };
    for (int g = 0; g < G; ++g) {                          // conv groups: nine taps of one 32-channel window
        step(g, std::integral_constant<int, 0>{}, true, g, 1);
        step(g, std::integral_constant<int, 1>{}, false, g, 2);
        step(g, std::integral_constant<int, 2>{}, false, g, 3);
        step(g, std::integral_constant<int, 3>{}, false, g, 4);
        step(g, std::integral_constant<int, 4>{}, false, g, 5);
        step(g, std::integral_constant<int, 5>{}, false, g, 6);
        step(g, std::integral_constant<int, 6>{}, false, g, 7);
        step(g, std::integral_constant<int, 7>{}, false, g, 8);
        step(g, std::integral_constant<int, 8>{}, false, g + 1, g + 1 < G ? 0 : 4);
    }
    for (int g = G; g < GT; ++g)                           // 1x1-residual groups: one step each, the centre tap of the residual input's window
        step(g, std::integral_constant<int, 4>{}, true, g + 1, 4);

    // ---------------- epilogue: lane (lq, lg) holds pixels 4 lg .. 4 lg + 3 of output channel lq of every fragment
    const int E = p.gn_exps[2 * b] + p.w_rp_exp;
    const float unscale = ldexpf(1.0f, -E);
    const bool idres = p.res0.data && !rwf;
    const float rs = idres ? p.res0.scale : 0.0f;
    double (*red)[CW_NT * 16 * 2] = reinterpret_cast<double (*)[CW_NT * 16 * 2]>(&bw[0][0]);
    if (p.out_stats) __syncthreads();                      // (the last step's B fragments have been read)
    const size_t HW = (size_t)H * W;
    const float* rimg = idres ? p.res0.data + (size_t)mi_row_of(b, p.res0.bmod) * p.res0.C * HW : nullptr;
    float* oimg = p.out + (size_t)b * p.Cout * HW;
    const int ox = tx0 + 4 * lg;
    int wcnt = 0;
    {
        const int wx = W - tx0, nx = wx > CW_TW ? CW_TW : wx;
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) wcnt += (ty0 + wm * 4 + mf < H) ? nx : 0;
    }
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const int col = (wn * NF + nf) * 16 + lq, co = nt0 * 16 + col;
        float bv = p.bias ? p.bias[co] : 0.0f;
        if (Cres && p.res_b) bv += p.res_b[co];
        float4 yv[4];
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int oy = ty0 + wm * 4 + mf;
            const bool ok = oy < H && ox < W;
            float4 y;
            y.x = fmaf(acc[mf][nf][0], unscale, bv); y.y = fmaf(acc[mf][nf][1], unscale, bv);
            y.z = fmaf(acc[mf][nf][2], unscale, bv); y.w = fmaf(acc[mf][nf][3], unscale, bv);
            if (idres) {
                const float4 r = ok ? mi_ldg4(rimg + ((size_t)co * H + oy) * W + ox) : make_float4(0.f, 0.f, 0.f, 0.f);
                y.x = fmaf(r.x, rs, y.x); y.y = fmaf(r.y, rs, y.y); y.z = fmaf(r.z, rs, y.z); y.w = fmaf(r.w, rs, y.w);
            }
            yv[mf] = y;
            if (ok) mi_stg4(oimg + ((size_t)co * H + oy) * W + ox, y);
        }
        if (p.out_stats) {
            // statistics about a per-channel shift (the channel's first value in this wave): common.hip.h mi_stat_acc
            mi_stat_acc a;
            a.c = __shfl(yv[0].x, lq);
            a.s = 0.0f; a.q = 0.0f; a.n = wcnt;
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) {
                const bool ok = ty0 + wm * 4 + mf < H && ox < W;
                const float d0 = yv[mf].x - a.c, d1 = yv[mf].y - a.c, d2 = yv[mf].z - a.c, d3 = yv[mf].w - a.c;
                a.s += ok ? (d0 + d1) + (d2 + d3) : 0.0f;
                a.q += ok ? fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, d3 * d3))) : 0.0f;
            }
            a.s += __shfl_xor(a.s, 16); a.q += __shfl_xor(a.q, 16);
            a.s += __shfl_xor(a.s, 32); a.q += __shfl_xor(a.q, 32);
            if (lg == 0) mi_stat_finish(a, red[wm][2 * col], red[wm][2 * col + 1]);
        }
    }
    if (p.out_stats) {
        __syncthreads();
        if (tid < CW_NT * 16 * 2)
            p.out_stats[((size_t)(b * p.Cout + nt0 * 16 + (tid >> 1)) * tiles + tile) * 2 + (tid & 1)] = red[0][tid] + red[1][tid];
    }
}

}  // namespace

extern "C" long long mi_conv_prep_bytes(int B, int Cin, int Cres, int H, int W) {
    return (long long)B * ((Cin + Cres) / 8) * 2 * cw_hp(H) * cw_wp(W) * 16;
}

static int cw_check(const mi_conv_params& p, const char* who) {
    const int C0 = p.in0.C, C1 = p.in1.data ? p.in1.C : 0, Cin = C0 + C1;
    const int Cr0 = (p.res0.data && p.res_w_rp) ? p.res0.C : 0, Cr1 = (Cr0 && p.res1.data) ? p.res1.C : 0, Cres = Cr0 + Cr1;
    if (p.ksize != 3 || p.stride != 1) { mi_set_error("%s: the wide GEMM kernel is k3 s1 (optionally behind a nearest x2 up-sampling)", who); return MI_ERR_UNSUPPORTED; }
    if (p.up2 && (((p.H | p.W) & 1) || p.res0.data)) { mi_set_error("%s: up2 needs an even output size and takes no residual", who); return MI_ERR_INVALID; }
    if ((C0 & 7) || (C1 & 7) || (Cr0 & 7) || (Cr1 & 7) || (Cin & 31) || (Cres & 31) || Cin <= 0 || (p.Cout & 63) || (p.W & 3) || p.B <= 0 || p.H <= 0) {
        mi_set_error("%s: the wide GEMM kernel needs input / residual channels in multiples of 32 (each concat part of 8), output channels of 64, W %% 4 == 0", who);
        return MI_ERR_UNSUPPORTED;
    }
    if (!p.gn_coef || !p.gn_exps || !p.act_prep || !p.w_rp) { mi_set_error("%s: gn_coef / gn_exps (mi_gn_coef_fwd), act_prep and w_rp are required", who); return MI_ERR_INVALID; }
    if (p.act_prep_bytes < mi_conv_prep_bytes(p.B, Cin, Cres, p.H, p.W)) { mi_set_error("%s: act_prep buffer too small", who); return MI_ERR_INVALID; }
    if (p.in0.st || (p.in1.data && p.in1.st) || p.out_st || (p.tile_cfg & MI_CONV_HALF)) { mi_set_error("%s: fp32 storage, three-term products only", who); return MI_ERR_UNSUPPORTED; }
    if ((size_t)((Cin + Cres) / 8) * 2 * cw_hp(p.H) * cw_wp(p.W) >= (1ull << 31)) { mi_set_error("%s: one image's operand planes are indexed with 32 bits", who); return MI_ERR_UNSUPPORTED; }
    return MI_OK;
}

extern "C" int mi_conv_prep_fwd(const mi_conv_params* pp, void* stream) {
    const mi_conv_params& p = *pp;
    if (int rc = cw_check(p, "mi_conv_prep_fwd")) return rc;
    const int Cin = p.in0.C + (p.in1.data ? p.in1.C : 0);
    const int Cres = (p.res0.data && p.res_w_rp) ? p.res0.C + (p.res1.data ? p.res1.C : 0) : 0;
    const int plane = cw_hp(p.H) * cw_wp(p.W);
    hipLaunchKernelGGL(conv_prep_kernel, dim3((plane + 255) / 256, (Cin + Cres) / 8, p.B), dim3(256), 0, (hipStream_t)stream, p, (uint4*)p.act_prep);
    return mi_check_launch("conv_prep_kernel");
}

int mi_conv_wide_launch(const mi_conv_params& p, hipStream_t st) {
    if (int rc = cw_check(p, "mi_conv_fwd")) return rc;
    if (p.res0.data && !p.res_w_rp && p.res0.C != p.Cout) { mi_set_error("mi_conv_fwd: identity residual needs Cres == Cout"); return MI_ERR_INVALID; }
    if (p.res0.data && !p.res_w_rp && p.res0.st) { mi_set_error("mi_conv_fwd: fp32 residual only"); return MI_ERR_UNSUPPORTED; }
    const int tiles = ((p.H + CW_TH - 1) / CW_TH) * ((p.W + CW_TW - 1) / CW_TW);
    // 128 output channels per workgroup where that still gives every CU two workgroups (or the layer has no 64-channel remainder to spare), else 64:
    // the results do not depend on the choice (the K order of an output is the same)
    const bool n128 = (p.Cout & 127) == 0 && ((long long)tiles * p.B * (p.Cout / 128) >= 512 || getenv("MI_CONV_WIDE_N128"));
    if (n128) hipLaunchKernelGGL(conv_wide_kernel<4>, dim3(tiles * p.B, p.Cout / 128), dim3(256), 0, st, p, (const uint4*)p.w_rp,
                                 (const uint4*)(p.res0.data ? p.res_w_rp : nullptr), (const uint4*)p.act_prep);
    else hipLaunchKernelGGL(conv_wide_kernel<2>, dim3(tiles * p.B, p.Cout / 64), dim3(256), 0, st, p, (const uint4*)p.w_rp,
                            (const uint4*)(p.res0.data ? p.res_w_rp : nullptr), (const uint4*)p.act_prep);
    return mi_check_launch("conv_wide_kernel");
}
