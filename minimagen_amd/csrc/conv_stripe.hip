// Full-width-stripe, wave-specialised form of the row-paired matrix-core 3x3 convolution (conv_rp.hip) for the narrow k3 s1 layers of
// MinImagen's U-Nets on images 32 .. 256 pixels wide: Block = GroupNorm -> [scale/shift] -> SiLU -> Conv3x3 (layers.py:131-145), the
// identity or 1x1-conv residual of ResnetBlock (layers.py:415-439), the skip concatenation (Unet.py:445), the next GroupNorm's partial
// statistics.  Same arithmetic as conv_rp_kernel -- the same per-channel affine, the same fp16 hi / lo operand split, the same MFMA
// triples in the same order, the same epilogue -- so outputs are bit-identical to it; what differs is how the data moves (round 6):
//
//   conv_rp_kernel: 8 x 64 tiles, every wave loads, transforms, multiplies and stores in turn.  Measured (profiles/r05_conv_dma_ablation.txt):
//   the phases of a tile ADD UP, the 10 x 66 windows re-read 29 % of the input, ~13 us of every launch is lock-step pipeline fill / drain.
//
//   here: a workgroup owns RS consecutive output rows of one image at FULL WIDTH (no horizontal halo; vertical halo (RS + 2) / RS) and walks
//   them two rows per step -- the row-paired MFMA form turns 4 input rows into 2 output rows -- over a RING of 6 transformed input rows in
//   LDS.  Waves are specialised:
//     * loader / transform waves: a lane owns 4 consecutive pixels of one new input row and the 8 channels of one octet = 8 dwordx4
//       loads per step, each wave-instruction a whole row run of a channel plane (1 KB at 256 wide); TWO steps in flight in registers,
//       issued unconditionally so that the wait counts stay exact; GroupNorm affine + SiLU + fp16 split in registers; 4 pixel chunks
//       (16 B hi + 16 B lo) written to the ring;
//     * MFMA / epilogue waves: A fragments from the ring (16-byte reads, lane group <-> row permutation 0, 2, 1, 3: conflict-free with the
//       pitch W + 8), B fragments from LDS (registers for the 8 -> 8 layers), identity-residual loads one step ahead, bias + residual +
//       float4 stores in the MFMA layout, statistics accumulated in registers and flushed once per block of W / 8 rows.  Before the first
//       multiply these waves have nothing to do, so THEY run the prologue: the producers' partial statistics -> channel totals -> (one wave)
//       group moments + per-channel affine + operand exponents, the B fragments, and the load + transform of the stripe's first four input
//       rows (steps -1 and 0) -- the loader waves start with steps 1 and 2 in flight.
//   One workgroup barrier per step: a step costs max(loader, consumer) instead of their sum, and the first output rows leave ~one memory
//   round trip after the launch instead of after the whole tile's load / transform / multiply sequence.
//   Measured on the skeleton of this structure before it was built (tools/ubench/stripe_pipe.hip, profiles/r06_stripe_pipe_ubench.txt):
//   8 -> 8 @256^2 + identity residual 79 us against 96 us (a plain copy of the same three streams: 78 us), @128^2 18.7 against 24.2 us.
//
// Statistics partition: one partial (sum, sum of squares) per channel and BLOCK of W / 8 rows (32 rows at 256 wide .. 4 at 32) -- a function
// of the image size only, whatever the number of blocks a workgroup takes (tile_cfg bits 12..15; speed only; default one), so a sharded
// batch reproduces the unsharded rows bit for bit.
// Phase traces of this kernel (tools/trace_stripe.py, profiles/r06_stripe_phase_trace.txt) are why the prologue looks the way it does: with
// one round of workgroups starting together, everything before the first multiply is exposed latency -- 22 % of a workgroup's life at 256^2,
// 46 % at 64^2 in the first version (statistics round trip under the initial burst of row loads, moments and affine through two barriers,
// the first two steps transformed by the loader waves alone).
#include "rp_common.hip.h"

#ifdef ST_TRACE
// development aid (tools/trace_stripe.py; -DST_TRACE builds only): shader-clock stamps of the prologue / pipeline phases of the first 1024
// workgroups of the last launch.  Slots 0..9: consumer wave 0 (start, totals done, barrier 1, affine done, B fragments staged, barrier 3, barrier 4,
// first step multiplied, loop done, statistics published); 10..15: loader wave 0 (start, first rows requested, barrier 3, -, barrier 4, done)
__device__ unsigned long long mi_trace_st_buf[1024 * 32];
extern "C" int mi_debug_read_trace_st(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mi_trace_st_buf), bytes); }
#define ST_STAMP(k) do { if (lane == 0 && blockIdx.x < 1024) mi_trace_st_buf[blockIdx.x * 32 + (k)] = clock64(); } while (0)
#else
#define ST_STAMP(k) do { } while (0)
#endif

#ifndef ST_ABL
#define ST_ABL 0            // timing-only ablation builds (tools/sweep_stripe.py): 1 plain workgroup -> image map, 2 no output statistics, 4 no statistics prologue
#endif
#ifndef ST_BREG
#define ST_BREG 1           // one conv octet, one N tile (8 -> 8 channels): the six B fragments stay in registers instead of being re-read from LDS every step
#endif
#ifndef ST_HALF_OCTETS
#define ST_HALF_OCTETS 1
#endif
#ifndef ST_NB_HS
#define ST_NB_HS 2
#endif
#ifndef ST_LDG2
#define ST_LDG2 1
#endif
#ifndef ST_LOOKAHEAD
#define ST_LOOKAHEAD 2
#endif
#ifndef ST_RO4
#define ST_RO4 1            // 0: A/B build without the 16 -> 16 + 1x1-residual-over-32 member (those launches on the tile kernel)
#endif
#ifndef ST_LOADER_DELAY
#define ST_LOADER_DELAY 56        // x 64 clocks (s_sleep) before the loader waves' first requests, <= 128 wide (0 = none)
#endif

namespace {

// KO_ / NJ_: channel octets of the conv input / N tiles (8 output channels each).  OM_ (output mode): 0 = every channel of the N tiles is an
// output channel, no identity residual; 1 = ... with the identity residual; 2 = fewer output channels than the N tiles hold (the final 8 -> 3 conv:
// masked stores), no residual; 3 = nearest x2 up-sampling in front of the conv (Upsample, layers.py:512-515: no GroupNorm, no residual): the ring
// holds SOURCE rows of half the width, one new row per step, and the A-fragment indices go through (v >> 1) as in conv_rp's MODE 1.  Compile-time, because a residual load or an output store inside a run-time conditional makes the compiler's wait
// counts conservative: the wait for step it's residual then also waits for the stores of step it - 1 and for the residual loads of step it + 1
// that were issued a moment ago (seen in the ISA of the first version: vmcnt(3) .. (0) in every step).
// (Launches with a 1x1 residual conv stay on the tile kernel: the residual octets' ring rows would leave one workgroup per CU.)
// RO_: channel octets of a 1x1 residual conv's input (ResnetBlock.res_conv over the block's input, layers.py:415,439): extra loader waves bring the two
// rows of the step's own output positions (no halo: the centre tap only) into a 4-row ring of their own; one more MFMA triple per octet and group.
template <int W_, int KO_, int NJ_, bool GN_, int OM_, int RO_ = 0>
struct StCfg {
    static constexpr int W = W_, KO = KO_, NJ = NJ_, OM = OM_, RO = RO_;
    static constexpr bool GN = GN_;
    // loader units per step: 2 rows x pixel quads x octets (x 2 channel halves: HS).  A work-item owns 4 consecutive pixels of one row and the 8
    // (HS: 4) channels of one (half) octet: that many dwordx4 loads per step, TWO steps in flight, issued unconditionally and unrolled by two so that
    // the wait before a transform is exact.  HS below 256 wide: a step's GroupNorm / SiLU / split of 32 values per work-item was the per-step critical
    // path there (phase trace: 2.7 us per step whatever the width); 16 values on twice the work-items halve it.  256-wide steps are paced by memory.
    // OM 4 (DS): the 4x4 stride-2 Downsample conv (layers.py:308-319) as in conv_rp's MODE 2 -- N = 16 output channels of ONE output row, K = 4 input
    // rows x 8 channels per horizontal tap (four taps); W_ is the OUTPUT width, the ring holds source rows of twice the width as two column-parity
    // planes (tap kx of output pixel x reads source column 2 x - 1 + kx: consecutive pixels are consecutive chunks of one plane), two new source
    // rows and one output row per step
    static constexpr bool UP = OM_ == 3, DS = OM_ == 4;
    static constexpr int NTAP = DS ? 4 : 3, NCO = DS ? 16 : 8, OPS = DS ? 1 : 2;           // horizontal taps, output channels per N tile, output rows per step
    static constexpr int WS = DS ? 2 * W_ : (UP ? W_ / 2 : W_), RPS = UP ? 1 : 2, NPR = UP ? 3 : 4;       // source width, new source rows per step, rows of the prologue
    static constexpr bool HS = W_ < 256 && ST_HALF_OCTETS;
    static constexpr int NCH = HS ? 4 : 8;
    // steps of rows in flight per loader work-item: the step time of the pipeline cannot be shorter than (load latency) / NB.  Half-octet units hold 4
    // dwordx4 per step, so four steps cost the registers two steps of whole octets do
    static constexpr int NB = (HS && W_ >= 64) ? ST_NB_HS : 2;
    static constexpr int QPR = WS / 4, UNITS = RPS * QPR * KO * (HS ? 2 : 1), NLWC = (UNITS + 63) / 64;
    // loader GROUPS: with one or two loader waves next to four MFMA waves, two of the CU's four SIMDs carry a loader wave AND an MFMA wave, the other
    // two an MFMA wave only -- and the transform (two transcendentals per value) costs about what the step's MFMAs do, so the step is paced by the
    // loaded SIMDs.  Two groups of loader waves take alternate steps (each with NB of ITS steps in flight): a loader wave on every SIMD, each
    // transforming every other step.  GroupNorm members only (a plain scaling is cheap), where the step count is a multiple of NB * LDG.
    // stages of LDS operands requested ahead of the MFMAs that use them (0 = the plain source-order loop: 256 wide, where the step is paced by memory
    // and the registers are taken by four pixel groups per wave)
    // (one stage for the member whose two accumulators, identity-residual prefetch and two stages of six fragments do not fit 128 registers)
    static constexpr int LA = W_ <= 128 ? ((NJ_ == 2 && KO_ == 2 && OM_ == 1 && ST_LOOKAHEAD > 1) ? 1 : ST_LOOKAHEAD) : 0;
    static constexpr int LDG = (ST_LDG2 && GN_ && RO_ == 0 && W_ >= 64 && W_ <= 128 && NLWC <= 2) ? 2 : 1;
    static constexpr int NLC = NLWC * LDG;                                                  // conv loader waves
    static constexpr int UNITS_R = 2 * QPR * RO, NLWR = (UNITS_R + 63) / 64, NLW = NLC + NLWR;      // residual rows: whole octets (a plain scaling: cheap)
    static constexpr int NG = W_ / 16, NCW = NG >= 4 ? 4 : NG, GPW = NG / NCW;             // 16-pixel groups of a row pair, MFMA waves
    static constexpr int NT = (NLW + NCW) * 64;
    static constexpr int PWH = WS / 2 + 2;                                                  // DS: chunks of one column-parity plane of a ring row (a zero chunk at either end)
    static constexpr int PW = DS ? 2 * PWH : WS + 8, RING = 6, RINGR = 4, PLANE = (KO * RING + RO * RINGR) * PW;
    static constexpr int SR0 = W_ / 8;                                                      // rows per statistics block
    static constexpr int WCH = NTAP * NJ * 128, WTOT = KO * WCH + RO * NJ * 128;               // 16-byte chunks of B fragments
    static constexpr bool BREG = ST_BREG && KO == 1 && NJ == 1 && RO == 0 && !DS && !(LDG == 2 && OM_ == 1);      // (with the identity residual prefetch they do not fit the 128 registers of two 8-wave workgroups per CU)
    static_assert(RO == 0 || (OM_ == 0 && GN_), "a 1x1 residual conv comes with a Block and replaces the identity residual");
    // waves per SIMD the register allocation must leave room for: the 128-wide residual-conv member has eight waves per workgroup and LDS for two
    // workgroups per CU -- 136 registers would leave one
    static constexpr int WPE = ((RO > 0 && W_ == 128) || LDG == 2) ? 4 : 1;
    // prologue units of the MFMA waves: the stripe's first 4 input rows, as half octets where that still fits one pass
    static constexpr bool PHS = HS && 2 * NPR * QPR * KO <= NCW * 64;
    static constexpr int PCH = PHS ? 4 : 8, PU = NPR * QPR * KO * (PHS ? 2 : 1);
    static_assert(PU <= NCW * 64, "the MFMA waves transform the first four rows in one pass");
    static_assert(!DS || (LA > 0 && !GN_ && RO_ == 0 && W_ <= 128), "the stride-2 member is a plain conv, at most 128 output columns");
};

// mi_gn_totals_issue (common.hip.h) with every load UNCONDITIONAL (clamped, always legal addresses; a missing statistics pointer reads the
// activation instead and is ignored): a load inside a run-time conditional makes the compiler's wait counts conservative, and the wait for these
// few latency-critical values then also waits for the bulk row loads requested right behind them (phase trace: 5 of a 64^2 launch's 21 us).
// Returns whether the loaded partials cover the problem (else the caller falls back to mi_gn_channel_totals).
__device__ __forceinline__ bool st_totals_issue(const mi_act& in0, const mi_act& in1, int C0, int Cin, int b, int lane, int nlanes, bool have, mi_stats_regs& r) {
    int TPC = 1;
    while (TPC < 64 && TPC * 2 * Cin <= nlanes) TPC *= 2;
    const int nt_max = (Cin > C0 && in1.nt > in0.nt) ? in1.nt : in0.nt;
    const bool covers = have && TPC * Cin <= nlanes && nt_max <= MI_STATS_K * TPC;
    const int c = lane / TPC, sub = lane % TPC;
    const bool live = covers && c < Cin;
    const bool second = live && c >= C0;
    const int cc = live ? (second ? c - C0 : c) : 0;
    const int nt = second ? in1.nt : in0.nt, CC = second ? in1.C : in0.C;
    const double* sp = second ? in1.stats : in0.stats;
    const int ba = mi_row_of(b, second ? in1.bmod : in0.bmod);
    const mi_gptr<const double> st = live ? mi_global(sp) + ((size_t)(ba * CC + cc) * nt) * 2 : mi_global(reinterpret_cast<const double*>(in0.data));
#pragma unroll
    for (int k = 0; k < MI_STATS_K; ++k) {
        const int t = sub + k * TPC;
        const bool ok = live && t < nt;
        const double x = st[2 * (ok ? t : 0)], y = st[2 * (ok ? t : 0) + 1];
        r.v[k] = make_double2(ok ? x : 0.0, ok ? y : 0.0);
    }
    r.c = live ? c : -1;
    r.tpc = TPC;
    r.scale = second ? in1.scale : in0.scale;
    return covers;
}

template <class CFG>
__global__ __launch_bounds__(CFG::NT, CFG::WPE) void conv_stripe_kernel(const mi_conv_params p, const uint4* __restrict__ wrp, const uint4* __restrict__ rwrp, const int nblk) {
    constexpr int W = CFG::W, KO = CFG::KO, NJ = CFG::NJ, QPR = CFG::QPR, UNITS = CFG::UNITS, NLT = CFG::NLW, NLWC = CFG::NLWC, NLC = CFG::NLC, LDG = CFG::LDG, RO = CFG::RO, RINGR = CFG::RINGR;
    constexpr int NCW = CFG::NCW, GPW = CFG::GPW, PW = CFG::PW, RING = CFG::RING, SR0 = CFG::SR0, WCH = CFG::WCH, WTOT = CFG::WTOT, PU = CFG::PU;
    constexpr bool GN = CFG::GN, BREG = CFG::BREG, UP = CFG::UP, DS = CFG::DS;
    constexpr int NTAP = CFG::NTAP, NCO = CFG::NCO, OPS = CFG::OPS, PWH = CFG::PWH;
    constexpr int WS = CFG::WS, RPS = CFG::RPS, NPR = CFG::NPR;
    __shared__ __attribute__((aligned(16))) uint4 actH[CFG::PLANE];
    __shared__ __attribute__((aligned(16))) uint4 actL[CFG::PLANE];
    __shared__ __attribute__((aligned(16))) uint4 wl[BREG ? 1 : WTOT];
    __shared__ __attribute__((aligned(16))) float4 chP[RP_MAXC];
    __shared__ double chS[RP_MAXC], chQ[RP_MAXC], chS2[RO ? RP_MAXC : 1], chQ2[RO ? RP_MAXC : 1];
    __shared__ double red[NCW][2 * NCO * NJ];
    __shared__ int sExp[4];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int H = p.H, HW = H * W;
    const int Hs = UP ? H / 2 : (DS ? 2 * H : H), HWs = UP ? HW / 4 : (DS ? 4 * HW : HW);           // source image
    const int nt = H / SR0;                            // statistics blocks per image
    const int wgs = nt / nblk;                         // workgroups per image
    // XCD-aware placement as in conv_rp.hip (speed only): workgroup L runs on XCD L % 8; whole images per XCD
    int b, sidx;
    if ((p.B & 7) == 0 && !(ST_ABL & 1)) {
        const int L = blockIdx.x;
        int k = L >> 3;
        if (p.tile_cfg & MI_CONV_REVERSE) k = (int)(gridDim.x >> 3) - 1 - k;
        b = (L & 7) + 8 * (k / wgs);
        sidx = k % wgs;
    } else {
        b = blockIdx.x / wgs;
        sidx = blockIdx.x % wgs;
    }
    const int blk0 = sidx * nblk, y0 = blk0 * SR0, RS = nblk * SR0, NSTEP = RS / OPS;
    const int ys0 = UP ? y0 / 2 : (DS ? 2 * y0 : y0);        // first source row of the stripe's own rows
    const int C0 = p.in0.C, C1 = p.in1.data ? p.in1.C : 0, Cin = C0 + C1;
    const bool have_stats = !(ST_ABL & 4) && (GN || p.in0.stats != nullptr);
    const int Cr0 = RO ? p.res0.C : 0, Cr1 = (RO && p.res1.data) ? p.res1.C : 0, Cres = Cr0 + Cr1;
    const bool res_stats = RO > 0 && p.res0.stats != nullptr && (Cr1 == 0 || p.res1.stats != nullptr);
    // first element of the 8 channel planes of octet `oct` of image b (in0, then in1: the skip concatenation).  The two tensors' image bases are
    // formed ONCE from the (scalar) kernel arguments; a work-item then selects between two computed pointers.  (Selecting the struct FIELDS per
    // work-item -- `second ? p.in1.data : p.in0.data` -- compiles to a vector load from the kernel-argument segment with a per-lane address and a
    // s_waitcnt vmcnt(0) right behind it: two dependent memory round trips in front of every wave's first bulk load, seen in the ISA.)
    const float* const img0 = p.in0.data + (size_t)mi_row_of(b, p.in0.bmod) * C0 * HWs;
    const float* const img1 = C1 ? p.in1.data + (size_t)mi_row_of(b, p.in1.bmod) * C1 * HWs : img0;
    auto octet_base = [&](int oct) {
        const int c0 = 8 * oct;
        const float* a0 = img0 + (size_t)c0 * HWs;
        const float* a1 = img1 + (size_t)(c0 - C0) * HWs;
        return mi_global(c0 >= C0 ? a1 : a0);
    };
    // GroupNorm affine + scale/shift + SiLU (or the plain operand scaling) + fp16 split of 4 pixels x NC channels (an octet, or its half `half`) ->
    // the 16-byte (8-byte) pieces of 4 pixel chunks of ring row `slot`
    auto transform_quad = [&](auto nc_tag, const f32x4* raw, const float4* P, bool inimg, bool live, int oct, int half, int slot, int q) {
        constexpr int NC = decltype(nc_tag)::value;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            float y[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const float x = raw[j][px];
                if constexpr (GN) {
                    const float a = fmaf(x, P[j].x, P[j].y);
                    const float ex = __builtin_amdgcn_exp2f(fmaf(x, P[j].z, P[j].w));       // exp(-a)
                    y[j] = a * __builtin_amdgcn_rcpf(1.0f + ex);
                } else {
                    y[j] = x * P[j].x;
                }
            }
            unsigned h[NC / 2], l[NC / 2];
#pragma unroll
            for (int i = 0; i < NC / 2; ++i) {
                const rp_f32x2 v = {y[2 * i], y[2 * i + 1]};
                h[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, rp_f16x2));
                l[i] = mi_split_lo2(h[i], y[2 * i], y[2 * i + 1]);
            }
            if (!inimg) {                                                                   // zero padding follows the activation (as in the reference)
#pragma unroll
                for (int i = 0; i < NC / 2; ++i) { h[i] = 0u; l[i] = 0u; }
            }
            if (live) {
                const int idx = (oct * RING + slot) * PW + (DS ? (px & 1) * PWH + 2 * q + (px >> 1) + 1 : 1 + 4 * q + px);      // DS: source column 4 q + px -> its parity plane
                if constexpr (NC == 8) {
                    actH[idx] = make_uint4(h[0], h[1], h[2], h[3]);
                    actL[idx] = make_uint4(l[0], l[1], l[2], l[3]);
                } else {
                    reinterpret_cast<uint2*>(&actH[idx])[half] = make_uint2(h[0], h[1]);
                    reinterpret_cast<uint2*>(&actL[idx])[half] = make_uint2(l[0], l[1]);
                }
            }
        }
    };

    if (RO > 0 && wave >= NLC && wave < NLT) {
        // =================================================================== residual-row loader waves (1x1 residual conv input, centre tap only)
        const int u = (wave - NLC) * 64 + lane;
        const bool live = u < CFG::UNITS_R;
        const int uu = live ? u : 0;
        const int oct = uu / (2 * QPR), lrow = (uu / QPR) & 1, q = uu % QPR;
        const float* const rimg0 = p.res0.data + (size_t)mi_row_of(b, p.res0.bmod) * Cr0 * HW;
        const float* const rimg1 = Cr1 ? p.res1.data + (size_t)mi_row_of(b, p.res1.bmod) * Cr1 * HW : rimg0;
        const bool second = 8 * oct >= Cr0;
        const mi_gptr<const float> base = mi_global(second ? rimg1 + (size_t)(8 * oct - Cr0) * HW : rimg0 + (size_t)(8 * oct) * HW);
        f32x4 raw[2][8];
        auto issue = [&](int s, auto buf_tag) {            // step s: the rows of its two output rows (steps past the stripe re-read the last one)
            constexpr int buf = decltype(buf_tag)::value;
            const int y = y0 + 2 * (s < NSTEP ? s : NSTEP - 1) + lrow;
            const unsigned off = (unsigned)(y * W + 4 * q);
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[buf][j] = *reinterpret_cast<mi_gptr<const f32x4>>(base + (size_t)j * HW + off);
        };
        constexpr std::integral_constant<int, 0> B0{};
        constexpr std::integral_constant<int, 1> B1{};
        // these waves also add up the residual input's statistics (its magnitude sets the 1x1 conv's operand exponent) -- FIRST, and their rows
        // after that: the rows are not needed before the first multiply, two memory round trips later, and the partials' 32 registers are free
        // again when the 64 of the two steps in flight are taken (together they spilled under the 128-register budget of two workgroups per CU)
        {
            mi_stats_regs sr2;
            const int rt = (wave - NLC) * 64 + lane;
            const bool fast2 = st_totals_issue(p.res0, p.res1, Cr0, Cres, b, rt, CFG::NLWR * 64, res_stats, sr2);
            if (fast2) mi_gn_totals_finish(sr2, rt, chS2, chQ2);
            else if (res_stats) mi_gn_channel_totals(p.res0, p.res1, Cr0, Cres, b, rt, CFG::NLWR * 64, chS2, chQ2);
        }
        issue(0, B0);
        issue(1, B1);
        if (have_stats) __syncthreads();                   // (1)
        __syncthreads();                                   // (3)
        const float rsc = ldexpf(second ? p.res1.scale : p.res0.scale, sExp[1]);
        // (one base address per plane and compile-time offsets from it: buffer b always holds a step of parity b, so the ring row is a constant per call
        //  site and the 16 LDS addresses of a step fold into the instructions' offset fields -- the first version kept them in registers and spilled)
        uint4* const dstH = &actH[(KO * RING + oct * RINGR + lrow) * PW + 1 + 4 * q];
        uint4* const dstL = &actL[(KO * RING + oct * RINGR + lrow) * PW + 1 + 4 * q];
        auto transform = [&](auto buf_tag) {               // buffer 0: even steps (ring rows 0, 1 of the octet), buffer 1: odd steps (rows 2, 3)
            constexpr int buf = decltype(buf_tag)::value;
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = raw[buf][j][px] * rsc;
                uint4 hv, lv;
                rp_split8(y, hv, lv);
                if (live) { dstH[2 * buf * PW + px] = hv; dstL[2 * buf * PW + px] = lv; }
            }
        };
        transform(B0);                                     // step 0
        issue(2, B0);
        __syncthreads();                                   // (4)
        for (int it = 0; it < NSTEP; it += 2) {
            transform(B1);                                 // step it + 1
            issue(it + 3, B1);
            __syncthreads();
            transform(B0);                                 // step it + 2
            issue(it + 4, B0);
            __syncthreads();
        }
    } else if (wave < NLT) {
        // =================================================================== loader / transform waves
        constexpr int NCH = CFG::NCH, HSM = CFG::HS ? 2 : 1;
        const int grp = LDG > 1 ? wave / NLWC : 0;         // loader group (wave-uniform): takes the steps s with (s - 1) % LDG == grp
        const int u = (wave - grp * NLWC) * 64 + lane;
        const bool live = u < UNITS;
        const int uu = live ? u : 0;
        const int oct = uu / (RPS * QPR * HSM), half = HSM == 2 ? (uu / (RPS * QPR)) & 1 : 0, lrow = RPS == 2 ? (uu / QPR) & 1 : 0, q = uu % QPR;   // octet, channel half, row of the step, pixel quad
        const mi_gptr<const float> base = octet_base(oct) + (size_t)(NCH * half) * HWs;
        constexpr int NB = CFG::NB;
        f32x4 raw[NB][NCH];
        bool inimg[NB];
        auto issue = [&](int s, auto buf_tag) {            // step s brings input rows y0 + 2 s + 1 + lrow (UP: source row y0 / 2 + s + 1); steps -1 and 0 are the MFMA waves' prologue
            constexpr int buf = decltype(buf_tag)::value;
            int y = UP ? ys0 + s + 1 : ys0 + 2 * s + 1 + lrow;
            const int ylast = UP ? (y0 + RS) / 2 : (DS ? 2 * (y0 + RS) : y0 + RS);
            y = y > ylast ? ylast : y;                    // steps past the stripe (issued UNCONDITIONALLY: a conditional issue makes the compiler's wait counts conservative) re-read its last halo row
            const bool ok = y < Hs;
            inimg[buf] = ok;
            const unsigned off = ok ? (unsigned)(y * WS + 4 * q) : 0u;
#pragma unroll
            for (int j = 0; j < NCH; ++j) raw[buf][j] = *reinterpret_cast<mi_gptr<const f32x4>>(base + (size_t)j * HWs + off);
        };
        float4 P[NCH];
        auto transform = [&](int s, auto buf_tag) {
            constexpr int buf = decltype(buf_tag)::value;
            transform_quad(std::integral_constant<int, NCH>{}, raw[buf], P, inimg[buf], live, oct, half, (UP ? s + 2 : 2 * s + 2 + lrow) % RING, q);      // ring row = source row - (first source row of the stripe - 1)
        };
        // one copy of the loop per group (a group's steps are issued unconditionally WITHIN its copy); every copy passes the same barriers
        auto run = [&](auto grp_tag) {
            constexpr int g = decltype(grp_tag)::value;
            if (wave == 0) ST_STAMP(10);
#if ST_LOADER_DELAY && !defined(HIPEMU)
            // Every workgroup of a small launch starts at once, and what it requests in its first microsecond is half of everything the launch reads:
            // the statistics and the first four rows -- the only requests the first multiply waits for -- queue among the loaders' two steps of rows,
            // which nobody needs for another ~8 us.  The loaders start ~1.5 us late (timing only): 64^2 level of the SR step 360 -> 347 us, SR step
            // -0.8 .. -1.5 % (profiles/r06_stripe_late_ab.txt; 0.4 / 0.8 us: less, 2.6 us: nothing).
            if constexpr (W <= 128) __builtin_amdgcn_s_sleep(ST_LOADER_DELAY);
#endif
            rp_for_rounds(std::make_integer_sequence<int, NB>{}, [&](auto j) { issue(1 + g + LDG * decltype(j)::value, j); });       // the group's first NB steps: its k-th step sits in buffer k % NB
            if (wave == 0) ST_STAMP(11);
            if (have_stats) __syncthreads();               // (1) channel totals in LDS
            __syncthreads();                               // (3) chP / sExp / B fragments / pads visible
            if (wave == 0) ST_STAMP(12);
#pragma unroll
            for (int j = 0; j < NCH; ++j) P[j] = chP[8 * oct + NCH * half + j];
            __syncthreads();                               // (4) rows of steps -1 and 0 in the ring
            if (wave == 0) ST_STAMP(14);
            for (int it = 0; it < NSTEP; it += NB * LDG) { // slot t: the MFMA waves multiply step t; the group with (t % LDG) == g transforms step t + 1 and requests its step t + 1 + NB LDG
                rp_for_rounds(std::make_integer_sequence<int, NB * LDG>{}, [&](auto jh) {
                    constexpr int j = decltype(jh)::value / LDG, h = decltype(jh)::value % LDG;
                    if constexpr (h == g) {
                        const int t = it + j * LDG + h;
                        if (wave == 0 && t == 2) ST_STAMP(20);
                        transform(t + 1, std::integral_constant<int, j>{});       // (the step past the last lands in ring rows nobody reads)
                        if (wave == 0 && t == 2) ST_STAMP(21);
                        issue(t + 1 + NB * LDG, std::integral_constant<int, j>{});
                        if (wave == 0 && t == 2) ST_STAMP(22);
                    }
                    __syncthreads();
                    if (wave == 0 && it + j * LDG + h == 2) ST_STAMP(23);
                });
            }
        };
        if (LDG == 1 || grp == 0) run(std::integral_constant<int, 0>{});
        else run(std::integral_constant<int, LDG - 1>{});
        if (wave == 0) ST_STAMP(15);
    } else {
        // =================================================================== MFMA / epilogue waves
        const int cw = wave - NLT, ct = tid - NLT * 64;
        constexpr int NCT = NCW * 64;
        if (cw == 0) ST_STAMP(0);
        // ---- prologue, everything requested in ONE memory round trip, every load UNCONDITIONAL (null pointers read a dummy: see st_totals_issue),
        // the short latency-critical ones first:
        // (a) the producers' partial statistics
        mi_stats_regs sr;
        const bool fast = st_totals_issue(p.in0, p.in1, C0, Cin, b, ct, NCT, have_stats, sr);
        // (b) the layer's parameters
        // (the loaded values are only TOUCHED after the first barrier: a select or an add right here would put a wait for them -- and for every
        //  older load -- in front of the bulk loads below)
        float pg = 0.f, pb = 0.f, ld_s1 = 0.f, ld_s2 = 0.f, ld_b[NJ], ld_rb[RO ? NJ : 1];
        if constexpr (GN) {
            const int c = lane < Cin ? lane : 0;
            pg = p.gn_gamma[c];
            pb = p.gn_beta[c];
            const float* ss = p.scale_shift ? p.scale_shift + (size_t)b * p.ss_stride + p.ss_off : p.gn_gamma;
            ld_s1 = ss[p.scale_shift ? c : 0];
            ld_s2 = ss[p.scale_shift ? Cin + c : 0];
        }
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            const int co = NCO * jt + (DS ? lq : (lq & 7));
            const bool hb = p.bias != nullptr && co < p.Cout;
            ld_b[jt] = (hb ? p.bias : p.in0.data)[hb ? co : 0];
            if constexpr (RO > 0) { const bool hr = p.res_b != nullptr && co < p.Cout; ld_rb[jt] = (hr ? p.res_b : p.in0.data)[hr ? co : 0]; }
        }
        // (c) the B fragments: straight into registers for the 8 -> 8 layers (global layout [tap][lane][hi | lo]), else staged for their LDS planes
        constexpr int WPER = BREG ? 1 : (WTOT + NCT - 1) / NCT;
        uint4 wreg[WPER];
        rp_f16x8 breg[BREG ? 6 : 1];
        if constexpr (BREG) {
#pragma unroll
            for (int k = 0; k < 6; ++k) breg[k] = __builtin_bit_cast(rp_f16x8, mi_ldg4u(wrp + 128 * (k >> 1) + 2 * lane + (k & 1)));
        } else {
#pragma unroll
            for (int i = 0; i < WPER; ++i) {
                const int k = ct + i * NCT;
                const int kc = k < WTOT ? k : 0;
                wreg[i] = (RO > 0 && kc >= KO * WCH) ? mi_ldg4u(rwrp + (kc - KO * WCH)) : mi_ldg4u(wrp + kc);
            }
        }
        // (d) the stripe's first four input rows (steps -1 and 0): one pixel quad x (half) octet per work-item
        constexpr int PCH = CFG::PCH, PHM = CFG::PHS ? 2 : 1;
        const bool plive = ct < PU;
        const int pu = plive ? ct : 0;
        const int poct = pu / (NPR * QPR * PHM), phalf = PHM == 2 ? (pu / (NPR * QPR)) & 1 : 0, prow = (pu / QPR) % NPR, pq = pu % QPR;
        f32x4 praw[PCH];
        const int py = ys0 - 1 + prow;
        const bool pin = py >= 0 && py < Hs;
        {
            const mi_gptr<const float> pbase = octet_base(poct) + (size_t)(PCH * phalf) * HWs;
            const unsigned off = pin ? (unsigned)(py * WS + 4 * pq) : 0u;
#pragma unroll
            for (int j = 0; j < PCH; ++j) praw[j] = *reinterpret_cast<mi_gptr<const f32x4>>(pbase + (size_t)j * HWs + off);
        }
        // the horizontal zero padding of the conv = a zero chunk left and right of every ring row
        for (int k = ct; k < KO * RING * 2; k += NCT) {
            const int r = k >> 1, c = DS ? ((k & 1) ? WS / 2 + 1 : PWH) : ((k & 1) ? WS + 1 : 0);      // (DS: source column -1 is chunk 0 of the odd plane, column WS the last chunk of the even one)
            actH[r * PW + c] = make_uint4(0u, 0u, 0u, 0u);
            actL[r * PW + c] = make_uint4(0u, 0u, 0u, 0u);
        }
        // ---- everything of the affine that does not need the statistics is done BEFORE they are waited for (the trace showed 1.9 us between the
        // totals and the affine in the first version): the fp64 reciprocal of the group size, the magnitude bound m = 4 |gamma'| + |beta'| of the
        // normalised activation and its power-of-two exponent (one wave; six cross-lane steps), and the B fragments' way into LDS
        double inv_n = 1.0;
        if constexpr (GN) inv_n = 1.0 / ((double)(Cin / p.gn_groups) * (double)HW);
        float psc = 1.0f, psh = 0.0f, An = 0.f, Bn = 0.f;
        int ka = 0;
        if constexpr (GN) {
            if (cw == 0) {
                psc = p.scale_shift ? ld_s1 + 1.0f : 1.0f;
                psh = p.scale_shift ? ld_s2 : 0.0f;
                float m = 0.f;
                if (lane < Cin) {
                    An = pg; Bn = pb;
                    if (p.scale_shift) { An *= psc; Bn = Bn * psc + psh; }
                    m = 4.0f * fabsf(An) + fabsf(Bn);                    // |SiLU(a)| <= |a|; 4 sigma of the normalised input
                }
                m = mi_wave_max(m);
                ka = (m > 0.f) ? rp_clamp_exp(4 - rp_exponent(m)) : 0;   // scaled magnitudes land in [8, 16)
            }
        }
        if constexpr (!BREG) {
#pragma unroll
            for (int i = 0; i < WPER; ++i) {
                const int k = ct + i * NCT;
                if (k < WTOT) wl[rp_wl_index(k)] = wreg[i];
            }
        }
        if (fast) mi_gn_totals_finish(sr, ct, chS, chQ);
        else if (have_stats) mi_gn_channel_totals(p.in0, p.in1, C0, Cin, b, ct, NCT, chS, chQ);          // (rare: more partials than the registers hold)
        if (cw == 0) ST_STAMP(1);
        if (have_stats) __syncthreads();                   // (1)
        if (cw == 0) ST_STAMP(2);
        // ---- ONE wave: group moments (each channel's lane adds up its group, in mi_gn_group_moments' order) -> per-channel affine of the fused
        // GroupNorm / scale-shift with the power-of-two operand scaling (conv_rp.hip's arithmetic)
        int kr = 0, Eall = 0;
        if (cw == 0) {
            const int c = lane;
            if constexpr (RO > 0) {      // one accumulator for the conv and the 1x1 residual conv: both products carry 2^E; lowering a scale is always safe (conv_rp.hip)
                float mr = 4.0f;
                if (res_stats) {
                    double qq = (c < Cres) ? chQ2[c] : 0.0;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) qq += __shfl_xor(qq, o);
                    mr = 4.0f * sqrtf((float)(qq / ((double)Cres * (double)HW)));
                }
                const int kr_max = (mr > 0.f) ? rp_clamp_exp(4 - rp_exponent(mr)) : 0;
                int E = ka + p.w_rp_exp;
                const int Er = kr_max + p.res_w_rp_exp;
                if (Er < E) E = Er;
                ka = E - p.w_rp_exp;
                kr = E - p.res_w_rp_exp;
                Eall = E;
            }
            if constexpr (GN) {
                if (c < Cin) {
                    const int cpg = Cin / p.gn_groups, g = c / cpg;
                    float mean, rstd;
                    if (ST_ABL & 4) { mean = 0.3f; rstd = 0.6f; }
                    else {
                        double gs = 0.0, gq = 0.0;
                        for (int cc = g * cpg; cc < (g + 1) * cpg; ++cc) { gs += chS[cc]; gq += chQ[cc]; }
                        const double mean_d = gs * inv_n;
                        double var = gq * inv_n - mean_d * mean_d;
                        var = var > 0.0 ? var : 0.0;
                        mean = (float)mean_d;
                        rstd = 1.0f / sqrtf((float)(var + (double)p.gn_eps));
                    }
                    float A = rstd * pg;
                    float Bc = pb - mean * A;
                    if (p.scale_shift) {
                        A *= psc;
                        Bc = Bc * psc + psh;
                    }
                    A *= (c >= C0) ? p.in1.scale : p.in0.scale;
                    chP[c] = make_float4(ldexpf(A, ka), ldexpf(Bc, ka), A * -1.44269504088896340736f, Bc * -1.44269504088896340736f);
                }
            } else {
                float m = 0.f;
                if (have_stats) {
                    double qq = (c < Cin) ? chQ[c] : 0.0;               // rms of the raw input (the tensors' scales are already applied)
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) qq += __shfl_xor(qq, o);
                    m = 4.0f * sqrtf((float)(qq / ((double)Cin * (double)HWs)));
                }
                ka = (m > 0.f) ? rp_clamp_exp(4 - rp_exponent(m)) : 0;
                if (c < Cin) chP[c] = make_float4(ldexpf((c >= C0) ? p.in1.scale : p.in0.scale, ka), 0.f, 0.f, 0.f);
            }
            if (lane == 0) { sExp[0] = ka; sExp[1] = kr; sExp[2] = RO > 0 ? Eall : ka + p.w_rp_exp; }
        }
        if (cw == 0) ST_STAMP(3);
        if (cw == 0) ST_STAMP(4);
        __syncthreads();                                   // (3)
        if (cw == 0) ST_STAMP(5);
        {
            float4 PP[PCH];
#pragma unroll
            for (int j = 0; j < PCH; ++j) PP[j] = chP[8 * poct + PCH * phalf + j];
            transform_quad(std::integral_constant<int, PCH>{}, praw, PP, pin, plive, poct, phalf, prow, pq);       // ring rows 0 .. 3 = input rows y0 - 1 .. y0 + 2
        }

        float bvv[NJ];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            const int cob = NCO * jt + (DS ? lq : (lq & 7));
            bvv[jt] = (p.bias != nullptr && cob < p.Cout) ? ld_b[jt] : 0.0f;
            if constexpr (RO > 0) bvv[jt] += (p.res_b != nullptr && cob < p.Cout) ? ld_rb[jt] : 0.0f;
        }
        const int perm = ((lg & 1) << 1) | (lg >> 1);     // lane group -> input row (0, 2, 1, 3)
        const int dy = DS ? 0 : lq >> 3;
        constexpr bool idres = CFG::OM == 1, MASKC = CFG::OM == 2 || DS;
        const mi_gptr<float> obuf = mi_global(p.out + (size_t)b * p.Cout * HW);
        const mi_gptr<const float> rbuf = mi_global(idres ? p.res0.data + (size_t)mi_row_of(b, p.res0.bmod) * p.res0.C * HW : p.out);
        const float unscale = ldexpf(1.0f, -sExp[2]);
        const float rs = idres ? p.res0.scale : 0.0f;
        f32x4 rv[2][idres ? GPW : 1][idres ? NJ : 1];
        auto issue_res = [&](int it, auto buf_tag) {       // identity residual of step it: unconditional loads (past the stripe: its last step again)
            constexpr int buf = decltype(buf_tag)::value;
            if constexpr (idres) {
                const int itc = it < NSTEP ? it : NSTEP - 1;
#pragma unroll
                for (int g = 0; g < GPW; ++g)
#pragma unroll
                    for (int jt = 0; jt < NJ; ++jt) {
                        const int co = 8 * jt + (lq & 7);
                        const int oy = y0 + 2 * itc + dy, ox = 16 * (cw * GPW + g) + 4 * lg;
                        rv[buf][g][jt] = *reinterpret_cast<mi_gptr<const f32x4>>(rbuf + (unsigned)(co * HW + oy * W + ox));
                    }
            }
        };
        float csum[NJ], csq[NJ], cshift[NJ];
        auto compute = [&](int it, auto buf_tag, bool first_of_block) {
            constexpr int buf = decltype(buf_tag)::value;
            f32x4 acc[GPW][NJ];
#pragma unroll
            for (int g = 0; g < GPW; ++g)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) acc[g][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (cw == 0 && it == 2) ST_STAMP(16);
            const int slot = (UP ? it + 1 + ((perm - 1) >> 1) : (DS ? 2 * it + lg : 2 * it + perm)) % RING;     // DS: lane group = vertical tap (source rows 2 y - 1 .. 2 y + 2)    // UP: up-sampled rows 2 it - 1 .. 2 it + 2 = source rows it - 1, it, it, it + 1
            // D[px 16][(dy, co)] += act[px][(r, ci)] . B[(r, ci)][(dy, co)]: octets in conv_rp's round order, one instruction triple per horizontal tap
            if constexpr (CFG::LA > 0) {
                // software pipeline over the step's STAGES (octet x horizontal tap, then the residual octets): the operands of stage st + LA are
                // requested from LDS before stage st is multiplied.  Written in source order the compiler requests a stage's six fragments right in
                // front of their first use -- a step of the 16 -> 16 @64^2 member then carried 21 exposed LDS waits, 1.9 of its 2.1 us (phase
                // trace, profiles/r06_stripe_phase_trace.txt) with 36 MFMAs = 0.25 us of matrix-core time.  Every accumulator still takes its
                // terms in the order (lo.hi, hi.lo, hi.hi) per stage: the bits do not change.
                constexpr int LA = CFG::LA, NS = NTAP * KO + RO, NBUF = LA + 1;
                rp_f16x8 sah[NBUF][GPW], sal[NBUF][GPW], sbh[NBUF][NJ], sbl[NBUF][NJ];
                const int rslot = RO > 0 ? 2 * (it & 1) + (perm >> 1) : 0;
                auto load = [&](auto st_tag) {
                    constexpr int st = decltype(st_tag)::value, bi = st % NBUF;
                    if constexpr (st < NTAP * KO) {
                        constexpr int o = st / NTAP, t = st % NTAP;
                        if constexpr (!BREG) {
#pragma unroll
                            for (int jt = 0; jt < NJ; ++jt) {
                                sbh[bi][jt] = __builtin_bit_cast(rp_f16x8, wl[o * WCH + (t * NJ + jt) * 128 + lane]);
                                sbl[bi][jt] = __builtin_bit_cast(rp_f16x8, wl[o * WCH + (t * NJ + jt) * 128 + 64 + lane]);
                            }
                        }
#pragma unroll
                        for (int g = 0; g < GPW; ++g) {
                            const int col = 16 * (cw * GPW + g) + lq + t;
                            const int idx = (o * RING + slot) * PW + (DS ? ((t + 1) & 1) * PWH + 16 * (cw * GPW + g) + lq + ((t + 1) >> 1) : (UP ? ((col - 1) >> 1) + 1 : col));
                            sah[bi][g] = __builtin_bit_cast(rp_f16x8, actH[idx]);
                            sal[bi][g] = __builtin_bit_cast(rp_f16x8, actL[idx]);
                        }
                    } else {                               // the 1x1 residual conv = the centre tap over the residual octets' own two rows (lane groups 0 / 3 multiply zero weights)
                        constexpr int o = st - NTAP * KO;
#pragma unroll
                        for (int jt = 0; jt < NJ; ++jt) {
                            sbh[bi][jt] = __builtin_bit_cast(rp_f16x8, wl[KO * WCH + (o * NJ + jt) * 128 + lane]);
                            sbl[bi][jt] = __builtin_bit_cast(rp_f16x8, wl[KO * WCH + (o * NJ + jt) * 128 + 64 + lane]);
                        }
#pragma unroll
                        for (int g = 0; g < GPW; ++g) {
                            const int idx = (KO * RING + o * RINGR + rslot) * PW + 16 * (cw * GPW + g) + lq + 1;
                            sah[bi][g] = __builtin_bit_cast(rp_f16x8, actH[idx]);
                            sal[bi][g] = __builtin_bit_cast(rp_f16x8, actL[idx]);
                        }
                    }
                };
                auto mult = [&](auto st_tag) {
                    constexpr int st = decltype(st_tag)::value, bi = st % NBUF;
                    rp_f16x8 bh[NJ], bl[NJ];
#pragma unroll
                    for (int jt = 0; jt < NJ; ++jt) {
                        if constexpr (BREG) { bh[jt] = breg[2 * (st % 3)]; bl[jt] = breg[2 * (st % 3) + 1]; }
                        else { bh[jt] = sbh[bi][jt]; bl[jt] = sbl[bi][jt]; }
                    }
                    // term by term over all accumulators: consecutive MFMAs go to different accumulators wherever the wave has more than one
#pragma unroll
                    for (int g = 0; g < GPW; ++g)
#pragma unroll
                        for (int jt = 0; jt < NJ; ++jt) acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sal[bi][g], bh[jt], acc[g][jt], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < GPW; ++g)
#pragma unroll
                        for (int jt = 0; jt < NJ; ++jt) acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sah[bi][g], bl[jt], acc[g][jt], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < GPW; ++g)
#pragma unroll
                        for (int jt = 0; jt < NJ; ++jt) acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sah[bi][g], bh[jt], acc[g][jt], 0, 0, 0);
                };
                rp_for_rounds(std::make_integer_sequence<int, (LA < NS ? LA : NS)>{}, load);
                rp_for_rounds(std::make_integer_sequence<int, NS>{}, [&](auto st_tag) {
                    constexpr int st = decltype(st_tag)::value;
                    if constexpr (st + LA < NS) load(std::integral_constant<int, st + LA>{});
                    __builtin_amdgcn_sched_barrier(0);     // (the machine scheduler otherwise sinks every request back to its first use)
                    mult(st_tag);
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else {
    #pragma unroll
                for (int o = 0; o < KO; ++o) {
    #pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        rp_f16x8 bh[NJ], bl[NJ];
    #pragma unroll
                        for (int jt = 0; jt < NJ; ++jt) {
                            if constexpr (BREG) { bh[jt] = breg[2 * s]; bl[jt] = breg[2 * s + 1]; }
                            else {
                                bh[jt] = __builtin_bit_cast(rp_f16x8, wl[o * WCH + (s * NJ + jt) * 128 + lane]);
                                bl[jt] = __builtin_bit_cast(rp_f16x8, wl[o * WCH + (s * NJ + jt) * 128 + 64 + lane]);
                            }
                        }
    #pragma unroll
                        for (int g = 0; g < GPW; ++g) {
                            const int col = 16 * (cw * GPW + g) + lq + s;
                            const int idx = (o * RING + slot) * PW + (UP ? ((col - 1) >> 1) + 1 : col);
                            const rp_f16x8 ah = __builtin_bit_cast(rp_f16x8, actH[idx]);
                            const rp_f16x8 al = __builtin_bit_cast(rp_f16x8, actL[idx]);
    #pragma unroll
                            for (int jt = 0; jt < NJ; ++jt) {
                                acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[jt], acc[g][jt], 0, 0, 0);
                                acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[jt], acc[g][jt], 0, 0, 0);
                            }
    #pragma unroll
                            for (int jt = 0; jt < NJ; ++jt) acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[jt], acc[g][jt], 0, 0, 0);
                        }
                    }
                }
                if constexpr (RO > 0) {      // the 1x1 residual conv = the centre tap over the residual octets' own two rows (lane groups 0 / 3 multiply zero weights)
                    const int rslot = 2 * (it & 1) + (perm >> 1);
    #pragma unroll
                    for (int o = 0; o < RO; ++o) {
                        rp_f16x8 bh[NJ], bl[NJ];
    #pragma unroll
                        for (int jt = 0; jt < NJ; ++jt) {
                            bh[jt] = __builtin_bit_cast(rp_f16x8, wl[KO * WCH + (o * NJ + jt) * 128 + lane]);
                            bl[jt] = __builtin_bit_cast(rp_f16x8, wl[KO * WCH + (o * NJ + jt) * 128 + 64 + lane]);
                        }
    #pragma unroll
                        for (int g = 0; g < GPW; ++g) {
                            const int idx = (KO * RING + o * RINGR + rslot) * PW + 16 * (cw * GPW + g) + lq + 1;
                            const rp_f16x8 ah = __builtin_bit_cast(rp_f16x8, actH[idx]);
                            const rp_f16x8 al = __builtin_bit_cast(rp_f16x8, actL[idx]);
    #pragma unroll
                            for (int jt = 0; jt < NJ; ++jt) {
                                acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[jt], acc[g][jt], 0, 0, 0);
                                acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[jt], acc[g][jt], 0, 0, 0);
                            }
    #pragma unroll
                            for (int jt = 0; jt < NJ; ++jt) acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[jt], acc[g][jt], 0, 0, 0);
                        }
                    }
                }
            }
            // epilogue: lane (lq, lg) holds pixels 4 lg .. 4 lg + 3 of channel lq & 7, row parity lq >> 3, of every group
#ifdef ST_TRACE
            if (cw == 0 && it == 2) { if (acc[0][0][0] == 123.456f) ST_STAMP(31); ST_STAMP(17); }       // (the comparison makes the stamp wait for the accumulators)
#endif
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                const int co = NCO * jt + (DS ? lq : (lq & 7));
                const float bv = bvv[jt];
                f32x4 yv[GPW];
#pragma unroll
                for (int g = 0; g < GPW; ++g) {
                    f32x4 y;
                    y[0] = fmaf(acc[g][jt][0], unscale, bv); y[1] = fmaf(acc[g][jt][1], unscale, bv);
                    y[2] = fmaf(acc[g][jt][2], unscale, bv); y[3] = fmaf(acc[g][jt][3], unscale, bv);
                    if constexpr (idres) { const f32x4 r = rv[buf][g][jt]; y[0] = fmaf(r[0], rs, y[0]); y[1] = fmaf(r[1], rs, y[1]); y[2] = fmaf(r[2], rs, y[2]); y[3] = fmaf(r[3], rs, y[3]); }
                    yv[g] = y;
                }
                if (first_of_block) {                      // statistics about a per-channel shift: the block's first value of the channel in this wave
                    cshift[jt] = __shfl(yv[0][0], DS ? lq : (lq & 7));
                    csum[jt] = 0.f; csq[jt] = 0.f;
                }
                const float cs_ = cshift[jt];
                const bool ok = !MASKC || co < p.Cout;
#pragma unroll
                for (int g = 0; g < GPW; ++g) {
                    const int oy = y0 + OPS * it + dy, ox = 16 * (cw * GPW + g) + 4 * lg;
                    const f32x4 y = yv[g];
                    if constexpr (MASKC) { if (ok) *reinterpret_cast<mi_gptr<f32x4>>(obuf + (unsigned)(co * HW + oy * W + ox)) = y; }
                    else *reinterpret_cast<mi_gptr<f32x4>>(obuf + (unsigned)(co * HW + oy * W + ox)) = y;
                    const float d0 = y[0] - cs_, d1 = y[1] - cs_, d2 = y[2] - cs_, d3 = y[3] - cs_;
                    csum[jt] += ok ? (d0 + d1) + (d2 + d3) : 0.0f;
                    csq[jt] += ok ? fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, d3 * d3))) : 0.0f;
                }
            }
            if (cw == 0 && it == 2) ST_STAMP(18);
        };
        auto flush = [&]() {                               // this wave's partial statistics of the block that just ended -> LDS (fp64)
            if (!p.out_stats || (ST_ABL & 2)) return;
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                float s_ = csum[jt], q_ = csq[jt];
                if constexpr (!DS) { s_ += __shfl_xor(s_, 8); q_ += __shfl_xor(q_, 8); }       // (DS: a lane's 16 columns are 16 channels of one row)
                s_ += __shfl_xor(s_, 16); q_ += __shfl_xor(q_, 16);
                s_ += __shfl_xor(s_, 32); q_ += __shfl_xor(q_, 32);
                if (lane < NCO) {
                    mi_stat_acc a; a.c = cshift[jt]; a.s = s_; a.q = q_; a.n = GPW * 16 * SR0;
                    mi_stat_finish(a, red[cw][2 * (NCO * jt + lane)], red[cw][2 * (NCO * jt + lane) + 1]);
                }
            }
        };
        auto publish = [&](int blk) {                      // after the step barrier: the block's partials of all waves, added in a fixed order
            if (!p.out_stats || cw != 0 || (ST_ABL & 2)) return;
            if (lane < 2 * NCO * NJ && (lane >> 1) < p.Cout) {
                double v = red[0][lane];
                if constexpr (NCW == 2) v = red[0][lane] + red[1][lane];
                if constexpr (NCW == 4) v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
                p.out_stats[((size_t)(b * p.Cout + (lane >> 1)) * nt + blk) * 2 + (lane & 1)] = v;
            }
        };
        issue_res(0, std::integral_constant<int, 0>{});
        __syncthreads();                                   // (4)
        if (cw == 0) ST_STAMP(6);
        constexpr int SPB = SR0 / OPS;                     // steps per statistics block
#pragma unroll 1
        for (int kb = 0; kb < nblk; ++kb) {
            const int it0 = kb * SPB;
#pragma unroll 1
            for (int i = 0; i < SPB; i += 2) {
                issue_res(it0 + i + 1, std::integral_constant<int, 1>{});
                compute(it0 + i, std::integral_constant<int, 0>{}, i == 0);
                if (cw == 0 && kb == 0 && i == 0) ST_STAMP(7);
                __syncthreads();
                if (cw == 0 && it0 + i == 2) ST_STAMP(19);
                issue_res(it0 + i + 2, std::integral_constant<int, 0>{});
                compute(it0 + i + 1, std::integral_constant<int, 1>{}, false);
                if (i + 2 == SPB) flush();
                __syncthreads();
            }
            if (cw == 0 && kb + 1 == nblk) ST_STAMP(8);
            publish(blk0 + kb);
        }
        if (cw == 0) ST_STAMP(9);
    }
}

template <int W, int KO, int NJ, bool GN, int OM, int RO = 0>
int launch_stripe(const mi_conv_params& p, hipStream_t st) {
    using CFG = StCfg<W, KO, NJ, GN, OM, RO>;
    const int nt = p.H / CFG::SR0;
    // statistics blocks per workgroup (speed only): tile_cfg bits 12..15.  Default: one block (W / 8 rows) at 128 / 256 wide, two at <= 64 wide while
    // that leaves a workgroup per CU -- measured isolated (tools/sweep_stripe.py) and in the captured SR step (64^2 level 384 -> 372 us with two,
    // 388 with one; base stage 0.446 -> 0.428 ms per step): profiles/r06_summary.md
    int nblk = (p.tile_cfg >> 12) & 0xf;
    if (nblk == 0) nblk = ((W <= 64 || OM == 2) && nt % 2 == 0 && (long long)p.B * (nt / 2) >= 256) ? 2 : 1;      // (OM 2: the final 8 -> 3 conv, 42 -> 38 us isolated at 256^2)
    if (nblk > nt || nt % nblk) nblk = 1;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_stripe_kernel<CFG>), dim3(p.B * (nt / nblk)), dim3(CFG::NT), 0, st, p, (const uint4*)p.w_rp, (const uint4*)p.res_w_rp, nblk);
    return mi_check_launch("conv_stripe_kernel");
}

template <int W>
int launch_stripe_w(const mi_conv_params& p, hipStream_t st, int ko, int nj, bool gn) {
    const int om = p.ksize == 4 ? 4 : (p.up2 ? 3 : (p.Cout < 8 * nj ? 2 : ((p.res0.data && !p.res_w) ? 1 : 0)));
    // the layer shapes of the BASELINE U-Nets: 8 / 16 / 24 / 32 input channels, 8 or 16 (or 3) output channels
#define ST_BLOCK(KO, NJ) if (gn && ko == KO && nj == NJ) return om == 1 ? launch_stripe<W, KO, NJ, true, 1>(p, st) : launch_stripe<W, KO, NJ, true, 0>(p, st)
#define ST_PLAIN(KO, NJ, OMV) if (!gn && ko == KO && nj == NJ && om == OMV) return launch_stripe<W, KO, NJ, false, OMV>(p, st)
    if constexpr (W == 64 || W == 128) {      // 8 -> 8 behind a Block with the 1x1 residual conv over 16 channels (ups.1 of both U-Nets)
        if (gn && ko == 1 && nj == 1 && om == 0 && p.res0.data && p.res_w) return launch_stripe<W, 1, 1, true, 0, 2>(p, st);
    }
    if constexpr (W == 64) {                  // 16 -> 16 with the 1x1 residual conv over 32 channels (ups.0 of the SR U-Net): one workgroup per CU (105 KB of LDS)
        if (gn && ko == 2 && nj == 2 && om == 0 && p.res0.data && p.res_w) return launch_stripe<W, 2, 2, true, 0, 4>(p, st);
    }
    if (om <= 1) {
        ST_BLOCK(1, 1);
        if constexpr (W <= 128) { ST_BLOCK(2, 1); }
        if constexpr (W <= 64) { ST_BLOCK(2, 2); ST_BLOCK(4, 2); ST_BLOCK(3, 2); }
    }
    ST_PLAIN(1, 1, 0);                                       // 8 -> 8 without GroupNorm
    ST_PLAIN(1, 1, 2);                                       // the final 8 -> 3 conv
    if constexpr (W <= 64) { ST_PLAIN(1, 2, 0); }            // the folded Parallel(3x3, 1x1) conv 8 -> 16 of the base U-Net
    if constexpr (W >= 64 && W <= 128) { ST_PLAIN(1, 1, 3); ST_PLAIN(2, 1, 3); }      // nearest x2 + conv
    if constexpr (W >= 64 && W <= 128) { ST_PLAIN(1, 1, 4); }                         // 4x4 stride 2: 8 -> 8 / 16 channels (one N tile of 16)
#undef ST_BLOCK
#undef ST_PLAIN
    return MI_ERR_UNSUPPORTED;
}

// rows per statistics block if this launch runs on the stripe kernel, else 0
int stripe_block_rows(const mi_conv_params& p, int* ko_, int* nj_) {
    const bool ds = p.ksize == 4 && p.stride == 2;
    if (!((p.ksize == 3 && p.stride == 1) || ds) || !p.w_rp || p.gn_coef || (p.tile_cfg & MI_CONV_HALF) || p.out_st) return 0;
    if (p.in0.st || (p.in1.data && p.in1.st) || (p.res0.data && p.res0.st)) return 0;
    if (!(p.W == 32 || p.W == 64 || p.W == 128 || p.W == 256)) return 0;
    const int sr0 = p.W / 8;
    if (p.H % sr0 || p.H < sr0 || (size_t)p.H * p.W * 64 >= (1ull << 31)) return 0;
    const int C0 = p.in0.C, C1 = p.in1.data ? p.in1.C : 0;
    if ((C0 & 7) || (C1 & 7) || C0 + C1 > RP_MAXC || p.Cout > 16) return 0;
    const bool rconv = p.res0.data && p.res_w;
    if (rconv) {        // 1x1 residual conv: instantiated for 8 -> 8 over 16 residual channels at 64 / 128 wide and 16 -> 16 over 32 at 64 wide
        const int cres = p.res0.C + (p.res1.data ? p.res1.C : 0);
        const bool m8 = cres == 16 && C0 + C1 == 8 && p.Cout == 8 && (p.W == 64 || p.W == 128), m16 = ST_RO4 && cres == 32 && C0 + C1 == 16 && p.Cout == 16 && p.W == 64;
        if (!p.res_w_rp || (p.res0.C & 7) || (p.res1.data && (p.res1.C & 7)) || (p.res1.data && p.res1.st) || !(m8 || m16) || !p.gn_groups || p.up2) return 0;
    } else if (p.res0.data && p.res0.C != p.Cout) return 0;
    if (p.gn_groups > MI_MAX_GROUPS || (p.gn_groups > 0 && ((C0 + C1) % p.gn_groups))) return 0;
    if (p.gn_groups > 0 && (!p.in0.stats || (p.in1.data && !p.in1.stats))) return 0;
    const int ko = (C0 + C1) >> 3, nj = ds ? (p.Cout + 15) >> 4 : (p.Cout + 7) >> 3;
    const bool gn = p.gn_groups > 0;
    if (ds) {           // 4x4 stride 2 (Downsample): 8 input channels, one N tile of 16 output channels, no GroupNorm / residual / up-sampling, output 64 or 128 wide
        if (gn || p.res0.data || p.up2 || C1 || ko != 1 || nj != 1 || !(p.W == 64 || p.W == 128)) return 0;
        if (ko_) { *ko_ = ko; *nj_ = nj; }
        return sr0;
    }
    const int om = p.up2 ? 3 : (p.Cout < 8 * nj ? 2 : ((p.res0.data && !rconv) ? 1 : 0));
    if (om == 3) {      // nearest x2 + conv (Upsample): 8 or 16 -> 8 channels, no GroupNorm, no residual, output at least 64 wide
        // (256-wide outputs stay on the tile kernel: write-bound, 52 against 48 us in the captured SR step; 128 wide: 20.8 against 27.8 us)
        if (gn || p.res0.data || nj != 1 || p.Cout != 8 || ko > 2 || p.W < 64 || p.W > 128 || (p.H & 1)) return 0;
        if (ko_) { *ko_ = ko; *nj_ = nj; }
        return sr0;
    }
    if ((om == 1 && !gn) || (om == 2 && (gn || nj != 1))) return 0;      // instantiated: identity residual behind a Block, masked channels in the final conv
    // the instantiated (ko, nj, gn) combinations of launch_stripe_w
    const bool common = (ko == 1 && nj == 1) || (p.W <= 128 && ko == 2 && nj == 1 && gn);
    const bool small = p.W <= 64 && ((ko == 2 && nj == 2 && gn) || (ko == 4 && nj == 2 && gn) || (ko == 1 && nj == 2 && !gn) || (ko == 3 && nj == 2 && gn));
    if (!common && !small) return 0;
    if (ko_) { *ko_ = ko; *nj_ = nj; }
    return sr0;
}

}  // namespace

/* rows per statistics block (out_nt = H / rows) when mi_conv_fwd would run this launch on the full-width-stripe kernel (tile_cfg 12), else 0 */
extern "C" int mi_conv_stripe_rows(const mi_conv_params* pp) { return stripe_block_rows(*pp, nullptr, nullptr); }

int mi_conv_stripe_launch(const mi_conv_params& p, hipStream_t st) {
    int ko = 0, nj = 0;
    if (!stripe_block_rows(p, &ko, &nj)) { mi_set_error("mi_conv_fwd: tile_cfg 12 (full-width stripes) does not take this launch (see mi_conv_stripe_rows)"); return MI_ERR_UNSUPPORTED; }
    const bool gn = p.gn_groups > 0;
    int rc = MI_ERR_UNSUPPORTED;
    switch (p.W) {
        case 32: rc = launch_stripe_w<32>(p, st, ko, nj, gn); break;
        case 64: rc = launch_stripe_w<64>(p, st, ko, nj, gn); break;
        case 128: rc = launch_stripe_w<128>(p, st, ko, nj, gn); break;
        case 256: rc = launch_stripe_w<256>(p, st, ko, nj, gn); break;
    }
    if (rc == MI_ERR_UNSUPPORTED) mi_set_error("mi_conv_fwd: stripe kernel not instantiated for this shape");
    return rc;
}
