// Row-paired matrix-core convolution for the NARROW layers of MinImagen's U-Nets (8..32 channels): the 3x3 stride-1 convs
// of Block / ResnetBlock (layers.py:131-145, 415-439) incl. the fused GroupNorm -> scale/shift -> SiLU prologue, the identity or
// 1x1-conv residual and the next GroupNorm's partial statistics; with MODE the nearest-x2 Upsample conv (layers.py:512-515) and
// the k4 s2 Downsample conv (layers.py:319).
//
// Why: with 8 channels the plain implicit GEMM leaves the matrix cores half empty (K = 8 of 16/32, N = 8 of 16), so round 1 ran
// these layers as fp32 VALU direct convolutions and measured them VALU-issue-bound at ~2.5 TB/s of algorithmic traffic.
// How: one v_mfma_f32_16x16x32_f16 computes D[16 pixels along x][N = (dy, co)] for TWO output rows at once:
//   N = 16 = output-row parity dy (2) x 8 output channels,
//   K = 32 = 4 input rows r (the rows the two output rows touch) x 8 input channels, for ONE horizontal tap kx,
//   B[(r, ci)][(dy, co)] = W[co][ci][ky = r - dy][kx]   (zero where r - dy is not a tap: 6 of the 8 (r, dy) pairs are live),
// so N is full and K is 75 % full whatever the channel count; wider layers loop over channel octets (rounds) and N tiles.
// A lane's A operand is ONE aligned 16-byte LDS read: activations are staged pixel-major, channel-interleaved, [row][x][8 halves]
// (hi plane, lo plane), with lane group lg <-> input row via the permutation (0, 2, 1, 3) so that the two rows a 16-lane
// ds_read_b128 group touches are 2 * PW chunks apart = 0 mod 16 chunks: conflict-free with an unpadded pitch.
// fp32 parity: every operand is split x = hi + lo (fp16 each) and multiplied as hi*hi + lo*hi + hi*lo (fp32 accumulate).  Both
// operands are brought into the fp16 normal range by exact power-of-two scalings -- the weights once at pack time
// (2^w_rp_exp), the activations per launch from the GroupNorm affine / the producer's statistics -- undone in the epilogue,
// so the split keeps ~22 bits whatever the magnitude of the checkpoint's weights.
#include "rp_common.hip.h"

#ifdef MI_TRACE
// development aid (tools/bench_conv.py): accumulated shader-clock time per phase of the first workgroups of the last launch, and
// (slot 7) their start / end on the 100 MHz wall clock
// slots 8 .. 15: finer stamps inside the MFMA-loop and epilogue phases (RP_TFINE; -DRP_TRACE_FINE): 8 residual + prefetch loads issued,
// 9 the MFMA loop proper, 10 epilogue arithmetic (incl. the wait for the residual loads), 11 stores issued, 12 statistics shuffles,
// 13 the barrier + partial-statistics store
__device__ unsigned long long mi_trace_rp_buf[1024 * 16];
extern "C" int mi_debug_read_trace_rp(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mi_trace_rp_buf), bytes); }
extern "C" int mi_debug_trace_rp_slots() { return 16; }
#define RP_TSTART() const unsigned long long rp_t_wall0 = wall_clock64(); unsigned long long rp_t_last = clock64(), rp_t_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define RP_TPHASE(k) do { const unsigned long long rp_t_now = clock64(); rp_t_acc[k] += rp_t_now - rp_t_last; rp_t_last = rp_t_now; } while (0)
#ifdef RP_TRACE_FINE
#define RP_TFINE(k) RP_TPHASE(k)
#else
#define RP_TFINE(k) do { } while (0)
#endif
#define RP_TEND() do { rp_t_acc[7] = (rp_t_wall0 << 32) | (wall_clock64() & 0xffffffffull); if (threadIdx.x == 0 && blockIdx.x < 1024) for (int k = 0; k < 16; ++k) mi_trace_rp_buf[blockIdx.x * 16 + k] = rp_t_acc[k]; } while (0)
#else
#define RP_TSTART() do { } while (0)
#define RP_TPHASE(k) do { } while (0)
#define RP_TFINE(k) do { } while (0)
#define RP_TEND() do { } while (0)
#endif

#ifndef RP_SCHED_BARRIER
#define RP_SCHED_BARRIER 1
#endif
#ifndef RP_BUF
#define RP_BUF 0          // 1: buffer-addressed loads / stores (32-bit offsets, no address VALU; measured: more registers, slower)
#endif

namespace {

// MODE 0: 3x3 stride 1.  MODE 1: nearest x2 up-sampling followed by 3x3 stride 1 (the low-resolution tile is staged; the taps
// address it through (v >> 1)).  MODE 2: 4x4 stride 2 (N = 16 output channels of ONE output row, K = 4 input rows x 8 channels
// per horizontal tap: full K).
// KO_ / RO_: channel octets of the conv input / of the 1x1-residual input when known at compile time (the layer shapes of the
// BASELINE U-Nets), -1 = read from the parameters: with the round structure static the whole tile loop is straight-line code, which
// is what lets the compiler keep the next tile's loads in flight across the MFMA loop and the epilogue (with run-time rounds it
// falls back to s_waitcnt vmcnt(0) at every join).
// WIDE_: the wide-channel regime (Unet() default, Base, Super: 128 .. 2048+ channels).  The output channels are tiled over blockIdx.y
// (8 NJ, or 16 NJ for MODE 2, per workgroup), the rounds are run-time, and the per-channel GroupNorm affine / the operand exponents come
// precomputed from gn_coef_kernel (global memory) instead of the in-kernel statistics prologue, which is sized for <= 64 channels.
template <int TH_, int TW_, int NJ_, bool GN_, bool HALF_, int MODE_, int KO_ = -1, int RO_ = -1, bool WIDE_ = false>
struct RpCfg {
    static constexpr int TH = TH_, TW = TW_, NJ = NJ_, MODE = MODE_, KO_T = KO_, RO_T = RO_;
    static constexpr bool WIDE = WIDE_;
    static constexpr bool GN = GN_, HALF = HALF_;
    // staged source window (rows x units) and LDS pitch in 16-byte chunks
    static constexpr int IH = MODE_ == 0 ? TH_ + 2 : (MODE_ == 1 ? TH_ / 2 + 2 : 2 * TH_ + 2);
    static constexpr int UW = MODE_ == 0 ? TW_ + 2 : (MODE_ == 1 ? TW_ / 2 + 2 : 2 * TW_ + 2);
    static constexpr int PW = MODE_ == 0 ? TW_ + 8 : (MODE_ == 1 ? TW_ / 2 + 8 : TW_ + 4);      // MODE 2: pitch of ONE column-parity plane
    static constexpr int PLANE = MODE_ == 2 ? 2 * IH * PW : IH * PW;
    static constexpr int NU = IH * UW, PER = (NU + 255) / 256;
    static constexpr int GX = TW_ / 16, GY = MODE_ == 2 ? TH_ : TH_ / 2, NG = GX * GY, GPW = NG / 4;
    static constexpr int NSTEP = MODE_ == 2 ? 4 : 3;
    // waves per SIMD the register allocation leaves room for: 4 where that needs no spills (a spill is a counted memory operation:
    // its s_waitcnt also waits for the prefetch), else 3
    // ... two for the 16-output-channel members on 8 x 64 tiles: at three they keep 168 registers and spill 21 (every spill reload's
    // s_waitcnt also waits for the prefetch); their launches at 64^2 are one round of <= 2 workgroups per CU anyway.  Measured, same box,
    // back to back: the 17 conv launches of the SR U-Net's 64^2 level 423 -> 383 us, SR step 1.461 -> 1.425 ms (profiles/r05_summary.md)
    // (the wide regime's four-N-tile kernels also spill at three, 39 registers, but measured no better at two: 32.9 vs 32.5 ms per step of Unet())
    // (16 x 64 tiles at 256^2, with or without their 17 spills: 285 / 327 us against 270-277 us for the four launches on 8 x 64 tiles)
    static constexpr int WPS_BASE = (NJ_ == 1 && TH_ * TW_ <= 512 && (KO_ + RO_ == 1 || TH_ * TW_ <= 256)) ? 4 : ((NJ_ >= 2 && TH_ * TW_ == 512) ? 2 : 3);
    // (round 6, tools/check_code_objects.py: the members with a 1x1 residual conv -- more rounds of raw registers in flight -- and the single-term
    //  k4 s2 members spilled 12 .. 84 bytes at these targets: one wave per SIMD less for them)
    static constexpr int WPS_REG = ((RO_ > 0 || (MODE_ == 2 && HALF_)) && WPS_BASE > 2) ? WPS_BASE - 1 : WPS_BASE;
    // ... and never more than the LDS of a CU holds workgroups (one wave per SIMD each): asking for an occupancy the LDS forbids only costs
    // registers (spills) -- 12 "failed to meet occupancy target" build warnings until round 6
    static constexpr int WCH_ = NSTEP * NJ_ * 128, WTOT_ = KO_ >= 0 ? KO_ * WCH_ + RO_ * NJ_ * 128 : WCH_;
    static constexpr int LDS_EST = PLANE * 16 * (HALF_ ? 1 : 2) + (HALF_ ? 16 : 0) + WTOT_ * 16 + 64 * 16 + 4 * 64 * 8 + 2 * 32 * 4 + 4 * NJ_ * (MODE_ == 2 ? 32 : 16) * 8 + 16;
    static constexpr int WPS_LDS = 163840 / LDS_EST < 1 ? 1 : 163840 / LDS_EST;
    static constexpr int WPS = WPS_REG < WPS_LDS ? WPS_REG : WPS_LDS;
};

template <class CFG>
__global__ __launch_bounds__(256, CFG::WPS) void conv_rp_kernel(const mi_conv_params p, const uint4* __restrict__ wrp, const uint4* __restrict__ rwrp,
                                                                const int ntile, const float4* __restrict__ coef, const int* __restrict__ exps) {
    constexpr bool WIDE = CFG::WIDE;
    constexpr int TH = CFG::TH, TW = CFG::TW, NJ = CFG::NJ, IH = CFG::IH, UW = CFG::UW, PW = CFG::PW, MODE = CFG::MODE;
    constexpr int NU = CFG::NU, PER = CFG::PER, GX = CFG::GX, GPW = CFG::GPW, NSTEP = CFG::NSTEP;
    constexpr bool GN = CFG::GN, HALF = CFG::HALF;
    __shared__ __attribute__((aligned(16))) uint4 actH[CFG::PLANE];
    __shared__ __attribute__((aligned(16))) uint4 actL[HALF ? 1 : CFG::PLANE];
    __shared__ __attribute__((aligned(16))) float4 chP[RP_MAXC];
    __shared__ double chS[RP_MAXC], chQ[RP_MAXC], chS2[RP_MAXC], chQ2[RP_MAXC];
    __shared__ float gMean[MI_MAX_GROUPS], gRstd[MI_MAX_GROUPS];
    __shared__ double red[4][NJ * (MODE == 2 ? 32 : 16)];
    __shared__ int sExp[4];
    // B fragments [step][jt][lane][hi, lo]: every round's set when the round structure is static (staged once per strip), else one round's
    constexpr int WCH = NSTEP * NJ * 128;                                        // 16-byte chunks of one conv round
    constexpr int WTOT = CFG::KO_T >= 0 ? CFG::KO_T * WCH + CFG::RO_T * NJ * 128 : WCH;
    __shared__ __attribute__((aligned(16))) uint4 wl[WTOT];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int H = p.H, W = p.W;
    const int tiles_x = (W + TW - 1) / TW, tiles = tiles_x * ((H + TH - 1) / TH);
    const int strips = (tiles + ntile - 1) / ntile;
    // One workgroup = one strip of `ntile` consecutive tiles of ONE image: the statistics / affine prologue is paid once per strip and
    // the next tile's loads fly under the current tile's MFMA loop and epilogue.
    // XCD-aware placement (speed only): workgroup L runs on XCD L % 8; give each XCD whole images so that halo re-reads and the
    // producer -> consumer hand-over of an image stay in one L2
    int b, strip;
    if ((p.B & 7) == 0) {
        const int L = blockIdx.x;
        int k = L >> 3;
        if (p.tile_cfg & MI_CONV_REVERSE) k = (int)(gridDim.x >> 3) - 1 - k;   // last image group first: what the previous launch wrote last is read first
        b = (L & 7) + 8 * (k / strips);
        strip = k % strips;
    } else {
        b = blockIdx.x / strips;
        strip = blockIdx.x % strips;
    }
    const int tile_lo = strip * ntile, tile_hi = (tile_lo + ntile < tiles) ? tile_lo + ntile : tiles;
    const int C0 = p.in0.C, C1 = p.in1.data ? p.in1.C : 0, Cin = C0 + C1;
    const int Cr0 = (p.res0.data && rwrp) ? p.res0.C : 0, Cr1 = (Cr0 && p.res1.data) ? p.res1.C : 0, Cres = Cr0 + Cr1;
    constexpr int CPT = MODE == 2 ? 16 : 8;                               // output channels per N tile
    const int jt0 = WIDE ? (int)blockIdx.y * NJ : 0;                      // first N tile of this workgroup
    const int njt = (p.Cout + CPT - 1) / CPT;                             // N tiles of the whole layer (the fragments' jt stride)
    constexpr bool STATIC_ROUNDS = CFG::KO_T >= 0;
    const int KO = STATIC_ROUNDS ? CFG::KO_T : (Cin >> 3), RO = STATIC_ROUNDS ? CFG::RO_T : (Cres >> 3), rounds = KO + RO;
    constexpr int RT = STATIC_ROUNDS ? CFG::KO_T + CFG::RO_T : 1;         // static round count (1 slot in the generic kernel)
    constexpr int NSLOT = RT;                                             // raw-load register slots (one per round; <= PFD + 1 are live at a time)
    constexpr int PFD = RT >= 2 ? 2 : 1;                                  // prefetch distance in rounds
    // source image extent
    const int Hs = MODE == 0 ? H : (MODE == 1 ? H / 2 : 2 * H), Ws = MODE == 0 ? W : (MODE == 1 ? W / 2 : 2 * W);
    const int HWs = Hs * Ws;

    RP_TSTART();
    // ---------------- staging geometry: one unit = one source pixel (all 8 channels of the round); everything but the tile origin is
    // the same for every tile of the strip
    int ldst[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int q = tid + u * 256;
        const int iy = q / UW, c = q - iy * UW;
        const bool valid = q < NU;
        if (MODE == 2) ldst[u] = valid ? ((c & 1) * IH + iy) * PW + (c >> 1) : -1;       // de-interleaved column parities
        else ldst[u] = valid ? iy * PW + c : -1;
    }
    float raw[NSLOT][PER][8];
    bool rst[NSLOT];                  // the slot's loads are 16-bit values (bf16 storage, single-term kernels only)
    unsigned inmask[NSLOT];           // bit u: unit u of the tile whose loads are in the slot lies inside the image
    auto load_raw = [&](int tile, int rnd, auto slot_tag) {
        constexpr int slot = decltype(slot_tag)::value;
        const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
        const int sy0 = MODE == 0 ? oy0 - 1 : (MODE == 1 ? oy0 / 2 - 1 : 2 * oy0 - 1);
        const int sx0 = MODE == 0 ? ox0 - 1 : (MODE == 1 ? ox0 / 2 - 1 : 2 * ox0 - 1);
        const bool isres = rnd >= KO;
        const int c0 = 8 * (isres ? rnd - KO : rnd);
        const mi_act& t0 = isres ? p.res0 : p.in0;
        const mi_act& t1 = isres ? p.res1 : p.in1;
        const int Ca = isres ? Cr0 : C0, Cb = isres ? Cr1 : C1;
        const bool second = c0 >= Ca;
        const int bb = mi_row_of(b, second ? t1.bmod : t0.bmod);
        const size_t e0 = second ? ((size_t)bb * Cb + (c0 - Ca)) * HWs : ((size_t)bb * Ca + c0) * HWs;      // first element of the round's planes
        const float* basep = (second ? t1.data : t0.data) + e0;
        // single-term kernels: the tensor may be stored as bf16 (mi_act.st); the raw registers then hold the 16 bits, expanded in the transform
        bool s16 = false;
        if constexpr (HALF) s16 = (second ? t1.st : t0.st) != 0;
        rst[slot] = s16;
        const mi_gptr<const unsigned short> base16 = mi_global(reinterpret_cast<const unsigned short*>(second ? t1.data : t0.data) + e0);
#if RP_BUF
        const mi_buf base = mi_make_buf(basep);
#else
        const mi_gptr<const float> base = mi_global(basep);
#endif
        unsigned off[PER];
        unsigned im = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int q = tid + u * 256;
            const int iy = q / UW, c = q - iy * UW;
            const int gy = sy0 + iy, gx = sx0 + c;
            const bool in = ldst[u] >= 0 && gy >= 0 && gy < Hs && gx >= 0 && gx < Ws;
            im |= in ? (1u << u) : 0u;
            off[u] = in ? (unsigned)(gy * Ws + gx) * 4u : 0u;  // byte offset; outside / unused slots read element 0 (legal, masked below)
        }
        // channel planes through the scalar offset of a buffer load: one address register per unit, no VALU work per load
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int u = 0; u < PER; ++u) {
#if RP_BUF
                raw[slot][u][j] = mi_buf_load_f32(base, off[u], (unsigned)(j * HWs) * 4u);
#else
                if (HALF && s16)
                    raw[slot][u][j] = __uint_as_float((unsigned)*reinterpret_cast<mi_gptr<const unsigned short>>(reinterpret_cast<mi_gptr<const char>>(base16 + (size_t)j * HWs) + (off[u] >> 1)));
                else
                    raw[slot][u][j] = *reinterpret_cast<mi_gptr<const float>>(reinterpret_cast<mi_gptr<const char>>(base + (size_t)j * HWs) + off[u]);
#endif
            }
        inmask[slot] = im;
    };
    // ---------------- the small loads first (statistics of the inputs: GroupNorm moments / magnitude for the fp16 scaling; the
    // per-channel affine parameters), the first tile's bulk loads right behind them: ONE memory round trip for the whole prologue, and
    // the statistics are reduced while the bulk loads are still in flight (vmcnt is in order)
    const bool have_stats = !WIDE && (GN || p.in0.stats != nullptr);
    const bool res_stats = !WIDE && RO > 0 && p.res0.stats != nullptr && (Cr1 == 0 || p.res1.stats != nullptr);
    constexpr int WPER = (WTOT + 255) / 256;           // 16-byte weight chunks per work-item
    uint4 wreg[WPER];
    if constexpr (STATIC_ROUNDS) {                     // every round's weight fragments: two linear copies (conv rounds, residual rounds)
        constexpr int WC = CFG::KO_T * WCH;
#pragma unroll
        for (int i = 0; i < WPER; ++i) {
            const int k = tid + i * 256;
            const int kc = k < WTOT ? k : 0;
            wreg[i] = (CFG::RO_T > 0 && kc >= WC) ? mi_ldg4u(rwrp + (kc - WC)) : mi_ldg4u(wrp + kc);
        }
    }
    mi_stats_regs sr, sr2;
    const bool fast = have_stats && mi_gn_totals_issue(p.in0, p.in1, C0, Cin, b, tid, 256, sr);
    const bool fast2 = res_stats && mi_gn_totals_issue(p.res0, p.res1, Cr0, Cres, b, tid, 256, sr2);
    float pg = 0.f, pb = 0.f, psc = 1.f, psh = 0.f;        // lane c < Cin: gamma, beta, scale + 1, shift of channel c
    if constexpr (GN && !WIDE) {
        const int c = lane < Cin ? lane : 0;
        pg = p.gn_gamma[c];
        pb = p.gn_beta[c];
        if (p.scale_shift) {
            const float* ss = p.scale_shift + (size_t)b * p.ss_stride + p.ss_off;
            psc = ss[c] + 1.0f;
            psh = ss[Cin + c];
        }
    }
    load_raw(tile_lo, 0, std::integral_constant<int, 0>{});
    if constexpr (PFD >= 2) load_raw(tile_lo, 1, std::integral_constant<int, 1>{});
    RP_TPHASE(0);       // geometry + the prologue's and the first tile's loads issued
    if constexpr (STATIC_ROUNDS) {
#pragma unroll
        for (int i = 0; i < WPER; ++i) {
            const int k = tid + i * 256;
            if (k < WTOT) wl[rp_wl_index(k)] = wreg[i];
        }
    }
    if (fast) mi_gn_totals_finish(sr, tid, chS, chQ);
    else if (have_stats) mi_gn_channel_totals(p.in0, p.in1, C0, Cin, b, tid, 256, chS, chQ);
    if (fast2) mi_gn_totals_finish(sr2, tid, chS2, chQ2);
    else if (res_stats) mi_gn_channel_totals(p.res0, p.res1, Cr0, Cres, b, tid, 256, chS2, chQ2);

    // ---------------- per-channel affine of the fused GroupNorm / scale-shift, and the power-of-two operand scalings
    if (have_stats || res_stats) __syncthreads();
    if constexpr (WIDE) {
        // final exponents from gn_coef_kernel (already reconciled between the conv and the 1x1-residual operands): exps[2b] = ka, [2b + 1] = kr
        if (tid == 0) { sExp[0] = exps[2 * b]; sExp[1] = exps[2 * b + 1]; sExp[2] = exps[2 * b] + p.w_rp_exp; }
    }
    if constexpr (GN && !WIDE) {
        const int cpg = Cin / p.gn_groups;
        for (int g = tid; g < p.gn_groups; g += 256)
            mi_gn_group_moments(chS, chQ, g * cpg, (g + 1) * cpg, (double)cpg * (double)HWs, p.gn_eps, gMean[g], gRstd[g]);
        __syncthreads();
    }
    if (!WIDE && wave == 0) {
        const int c = lane;
        float A = 0.f, Bc = 0.f, m = 0.f;
        if constexpr (GN) {
            if (c < Cin) {
                const int cpg = Cin / p.gn_groups, g = c / cpg;
                float An = pg, Bn = pb;         // the affine in the normalised domain: magnitude of the activation
                A = gRstd[g] * pg;
                Bc = pb - gMean[g] * A;
                if (p.scale_shift) {
                    A *= psc;
                    Bc = Bc * psc + psh;
                    An *= psc;
                    Bn = Bn * psc + psh;
                }
                A *= (c >= C0) ? p.in1.scale : p.in0.scale;
                m = 4.0f * fabsf(An) + fabsf(Bn);                    // |SiLU(a)| <= |a|; 4 sigma of the normalised input
            }
            m = mi_wave_max(m);
        } else if (have_stats) {
            double q = (c < Cin) ? chQ[c] : 0.0;                    // rms of the raw input (the tensors' scales are already applied)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
            m = 4.0f * sqrtf((float)(q / ((double)Cin * (double)HWs)));
        }
        int ka = (m > 0.f) ? rp_clamp_exp(4 - rp_exponent(m)) : 0;            // scaled magnitudes land in [8, 16)
        int E = ka + p.w_rp_exp, kr = 0;
        if (RO > 0) {
            float mr = 4.0f;
            if (res_stats) {
                double q = (c < Cres) ? chQ2[c] : 0.0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
                mr = 4.0f * sqrtf((float)(q / ((double)Cres * (double)HWs)));
            }
            const int kr_max = (mr > 0.f) ? rp_clamp_exp(4 - rp_exponent(mr)) : 0;
            const int Er = kr_max + p.res_w_rp_exp;
            if (Er < E) E = Er;                                      // one accumulator: both products carry 2^E; lowering a scale is always safe
            ka = E - p.w_rp_exp;
            kr = E - p.res_w_rp_exp;
        }
        if (c < Cin) {
            if constexpr (GN) {
                chP[c] = make_float4(ldexpf(A, ka), ldexpf(Bc, ka), A * -1.44269504088896340736f, Bc * -1.44269504088896340736f);
            } else {
                chP[c] = make_float4(ldexpf((c >= C0) ? p.in1.scale : p.in0.scale, ka), 0.f, 0.f, 0.f);
            }
        }
        if (lane == 0) { sExp[0] = ka; sExp[1] = kr; sExp[2] = E; }
    }
    RP_TPHASE(1);       // statistics round trip + affine / exponent prologue

    const int perm = ((lg & 1) << 1) | (lg >> 1);         // lane group -> input row (0, 2, 1, 3)
    const bool idres = p.res0.data && !rwrp;
#if RP_BUF
    const mi_buf obuf = mi_make_buf(p.out + (size_t)b * p.Cout * H * W);
    const mi_buf rbuf = mi_make_buf(idres ? p.res0.data + (size_t)mi_row_of(b, p.res0.bmod) * p.res0.C * H * W : p.out);
#else
    // (bf16 storage in the single-term kernels: the image's first element sits at half the byte offset)
    const size_t oel = (size_t)b * p.Cout * H * W, rel = idres ? (size_t)mi_row_of(b, p.res0.bmod) * p.res0.C * H * W : 0;
    const mi_gptr<float> obuf = mi_global(reinterpret_cast<float*>(reinterpret_cast<char*>(p.out) + oel * ((HALF && p.out_st) ? 2 : 4)));
    const mi_gptr<const float> rbuf = mi_global(reinterpret_cast<const float*>(
        reinterpret_cast<const char*>(idres ? p.res0.data : p.out) + rel * ((HALF && idres && p.res0.st) ? 2 : 4)));
#endif
    constexpr bool idres_any = !(CFG::RO_T > 0);          // a 1x1 residual conv excludes the identity residual

    // generic kernel: one round's B fragments at a time through LDS
    auto load_b = [&](int rnd) {
        const bool isres = rnd >= KO;
        const int k8 = isres ? rnd - KO : rnd;
        const int nstep = isres ? 1 : NSTEP;
        const uint4* src = isres ? rwrp + (size_t)k8 * njt * 128 : wrp + (size_t)k8 * NSTEP * njt * 128;     // [step][jt of the layer][lane][2]
        const int n = nstep * NJ * 128;
#pragma unroll
        for (int i = 0; i < WPER; ++i) {
            const int k = tid + i * 256;
            const int kc = k < n ? k : 0;
            const int s_ = kc / (NJ * 128), rem = kc % (NJ * 128);            // this workgroup's NJ tiles out of the layer's njt
            const int jt = jt0 + rem / 128;
            wreg[i] = (jt < njt) ? mi_ldg4u(src + ((size_t)s_ * njt + jt) * 128 + rem % 128) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto store_b = [&](int rnd) {
        const int n = rnd >= KO ? NJ * 128 : WCH;
#pragma unroll
        for (int i = 0; i < WPER; ++i) {
            const int k = tid + i * 256;
            if (k < n) wl[rp_wl_index(k)] = wreg[i];
        }
    };
    // bias per N tile (this lane's output channel), once per strip: a load inside the tile loop would have to wait for the prefetch
    float bvv[NJ];
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
        const int co = MODE == 2 ? 16 * (jt0 + jt) + lq : 8 * (jt0 + jt) + (lq & 7);
        bvv[jt] = (p.bias && co < p.Cout) ? p.bias[co] : 0.0f;
        if (Cres && p.res_b && co < p.Cout) bvv[jt] += p.res_b[co];
    }

    // the last tile of the strip is peeled (HAS_NEXT = false): a CONDITIONAL prefetch would merge two wait-count states at the join
    // and force the conservative one (wait for everything) on every tile
    auto do_tile = [&](const int tile, auto has_next_tag) {
        constexpr bool HAS_NEXT = decltype(has_next_tag)::value;
        const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
        f32x4 acc[GPW][NJ];
#pragma unroll
        for (int g = 0; g < GPW; ++g)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) acc[g][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float4 rv[idres_any ? GPW : 1][idres_any ? NJ : 1];
        auto one_round = [&](auto rnd_tag) {
            using RTag = decltype(rnd_tag);
            constexpr bool RC = !std::is_same<RTag, int>::value;           // compile-time round index
            constexpr int slot = [] { if constexpr (RC) return (int)RTag::value; else return 0; }();
            const int rnd = rnd_tag;
            const bool isres = rnd >= KO;
            const int k8 = isres ? rnd - KO : rnd;
            if constexpr (!STATIC_ROUNDS) load_b(rnd);
            RP_TPHASE(6);
            __syncthreads();          // previous round's / tile's LDS fully consumed; chP / sExp visible before the first transform
            RP_TPHASE(2);
            // ---- activations: GroupNorm-apply + scale/shift + SiLU (or the raw scaled input), fp16 split, one 16-byte chunk per pixel
            {
                float rsc = 0.f;
                if (isres) rsc = ldexpf((8 * k8 >= Cr0) ? p.res1.scale : p.res0.scale, sExp[1]);
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    if (ldst[u] < 0) continue;
                    float y[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float x = raw[slot][u][j];
                        if constexpr (HALF) { if (rst[slot]) x = __uint_as_float(__float_as_uint(x) << 16); }
                        if (isres) {
                            y[j] = x * rsc;
                        } else {
                            const float4 P = WIDE ? coef[(size_t)b * Cin + 8 * k8 + j] : chP[8 * k8 + j];
                            if constexpr (GN) {
                                const float a = fmaf(x, P.x, P.y);
                                const float ex = __builtin_amdgcn_exp2f(fmaf(x, P.z, P.w));       // exp(-a)
                                y[j] = a * __builtin_amdgcn_rcpf(1.0f + ex);
                            } else {
                                y[j] = x * P.x;
                            }
                        }
                    }
                    uint4 hv, lv = make_uint4(0u, 0u, 0u, 0u);
                    if constexpr (HALF) hv = rp_hi8(y); else rp_split8(y, hv, lv);
                    if (!((inmask[slot] >> u) & 1u)) { hv = make_uint4(0u, 0u, 0u, 0u); lv = hv; }      // zero padding follows the activation (as in the reference)
                    actH[ldst[u]] = hv;
                    if constexpr (!HALF) actL[ldst[u]] = lv;
                }
            }
            if constexpr (!STATIC_ROUNDS) store_b(rnd);
            RP_TPHASE(3);       // wait for the raw loads + transform + LDS write
            __syncthreads();
            RP_TPHASE(2);
            // the next loads (this tile's next round, or the next tile's first) fly under the MFMA loop and the epilogue
            // identity residual of THIS tile first, in the last round (consumed in the epilogue with the later loads still in flight: vmcnt is in order)
            if (rnd + 1 == rounds && idres) {
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    const int co = MODE == 2 ? 16 * (jt0 + jt) + lq : 8 * (jt0 + jt) + (lq & 7);
                    const int dy = MODE == 2 ? 0 : lq >> 3;
#pragma unroll
                    for (int g = 0; g < GPW; ++g) {
                        const int G = wave * GPW + g, gyy = G / GX, gxx = G % GX;
                        const int oy = oy0 + (MODE == 2 ? gyy : 2 * gyy + dy), ox = ox0 + 16 * gxx + 4 * lg;
                        // unconditional load from a clamped (always legal) address: no exec-masked block, so the waits stay counted
                        const bool ok = co < p.Cout && oy < H && ox < W;
                        const unsigned o = ok ? (unsigned)((co * H + oy) * W + ox) * 4u : 0u;
#if RP_BUF
                        const f32x4 r4 = mi_buf_load_f32x4(rbuf, o, 0u);
#else
                        f32x4 r4;
                        if (HALF && p.res0.st) {
                            typedef unsigned rp_u32x2 __attribute__((ext_vector_type(2)));
                            const rp_u32x2 r2 = *reinterpret_cast<mi_gptr<const rp_u32x2>>(reinterpret_cast<mi_gptr<const char>>(rbuf) + (o >> 1));
                            const float4 e = mi_bf16x4_to_f32(make_uint2(r2[0], r2[1]));
                            r4 = (f32x4){e.x, e.y, e.z, e.w};
                        } else {
                            r4 = *reinterpret_cast<mi_gptr<const f32x4>>(reinterpret_cast<mi_gptr<const char>>(rbuf) + o);
                        }
#endif
                        rv[g][jt] = make_float4(r4[0], r4[1], r4[2], r4[3]);
                    }
                }
            }
            // the loads PFD rounds ahead (this tile's, or the next tile's) fly under the MFMA loops and the epilogue
            if constexpr (RC) {
                constexpr int tgt = (int)RTag::value + PFD;
                if constexpr (tgt < RT) load_raw(tile, tgt, std::integral_constant<int, tgt>{});
                else if constexpr (HAS_NEXT) load_raw(tile + 1, tgt - RT, std::integral_constant<int, tgt - RT>{});
            } else {
                if (rnd + 1 < rounds) load_raw(tile, rnd + 1, std::integral_constant<int, 0>{});
                else if constexpr (HAS_NEXT) load_raw(tile + 1, 0, std::integral_constant<int, 0>{});
            }
#if RP_SCHED_BARRIER
            __builtin_amdgcn_sched_barrier(0);      // keep the loads AHEAD of the MFMA loop (the scheduler otherwise spreads them over it)
#endif
            RP_TFINE(8);
            // ---- D[px 16][(dy, co)] += act[px][(r, ci)] . B[(r, ci)][(dy, co)], one instruction triple per horizontal tap
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                if (isres && s > 0) continue;
                const int sx = isres ? 1 : s;                // the 1x1 residual conv is the centre tap
                rp_f16x8 bh[NJ], bl[NJ];
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    const int wo = STATIC_ROUNDS ? (isres ? CFG::KO_T * WCH + k8 * NJ * 128 : k8 * WCH) : 0;
                    bh[jt] = __builtin_bit_cast(rp_f16x8, wl[wo + (s * NJ + jt) * 128 + lane]);
                    if constexpr (!HALF) bl[jt] = __builtin_bit_cast(rp_f16x8, wl[wo + (s * NJ + jt) * 128 + 64 + lane]);
                }
#pragma unroll
                for (int g = 0; g < GPW; ++g) {
                    const int G = wave * GPW + g, gyy = G / GX, gxx = G % GX;
                    int idx;
                    if (MODE == 0) idx = (2 * gyy + perm) * PW + 16 * gxx + lq + sx;
                    else if (MODE == 1) idx = (((2 * gyy + perm - 1) >> 1) + 1) * PW + ((16 * gxx + lq + sx - 1) >> 1) + 1;
                    else idx = ((sx & 1) * IH + 2 * gyy + lg) * PW + 16 * gxx + lq + (sx >> 1);   // MODE 2: lane group = vertical tap; column 2x + kx -> parity plane
                    const rp_f16x8 ah = __builtin_bit_cast(rp_f16x8, actH[idx]);
                    if constexpr (!HALF) {
                        const rp_f16x8 al = __builtin_bit_cast(rp_f16x8, actL[idx]);
#pragma unroll
                        for (int jt = 0; jt < NJ; ++jt) {
                            acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[jt], acc[g][jt], 0, 0, 0);
                            acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[jt], acc[g][jt], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int jt = 0; jt < NJ; ++jt)
                        acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[jt], acc[g][jt], 0, 0, 0);
                }
            }
            RP_TFINE(9);
            RP_TPHASE(4);       // MFMA loop (+ the next load issue)
        };
        if constexpr (STATIC_ROUNDS) {
            rp_for_rounds(std::make_integer_sequence<int, RT>{}, one_round);
        } else {
            for (int rnd = 0; rnd < rounds; ++rnd) one_round(rnd);
        }

        // ---------------- epilogue: lane (lq, lg) holds pixels 4lg .. 4lg+3 of column j = lq of every group
        const float unscale = ldexpf(1.0f, -sExp[2]);
        // statistics of the output about a per-channel shift (the first value of the channel's first lane): common.hip.h mi_stat_acc
        float csum[NJ], csq[NJ], cshift[NJ];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            csum[jt] = 0.f; csq[jt] = 0.f;
            const int co = MODE == 2 ? 16 * (jt0 + jt) + lq : 8 * (jt0 + jt) + (lq & 7);
            const int dy = MODE == 2 ? 0 : lq >> 3;
            const float bv = bvv[jt];
            const float rs = idres ? p.res0.scale : 0.0f;
            float4 yv[GPW];
#pragma unroll
            for (int g = 0; g < GPW; ++g) {          // all arithmetic first, unconditionally (one counted wait for the residual loads) ...
                float4 y;
                y.x = fmaf(acc[g][jt][0], unscale, bv); y.y = fmaf(acc[g][jt][1], unscale, bv);
                y.z = fmaf(acc[g][jt][2], unscale, bv); y.w = fmaf(acc[g][jt][3], unscale, bv);
                if constexpr (idres_any) {
                    if (idres) { const float4 r = rv[g][jt]; y.x = fmaf(r.x, rs, y.x); y.y = fmaf(r.y, rs, y.y); y.z = fmaf(r.z, rs, y.z); y.w = fmaf(r.w, rs, y.w); }
                }
                yv[g] = y;
            }
            RP_TFINE(10);
#ifdef MI_STATS_PLAIN      // timing A/B only: statistics about zero (the fp32 sums of round 4)
            const float cs_ = 0.0f;
#else
            const float cs_ = __shfl(yv[0].x, MODE == 2 ? lq : (lq & 7));
#endif
            cshift[jt] = cs_;
#pragma unroll
            for (int g = 0; g < GPW; ++g) {          // ... then the (edge-masked) stores, which need no wait
                const int G = wave * GPW + g, gyy = G / GX, gxx = G % GX;
                const int oy = oy0 + (MODE == 2 ? gyy : 2 * gyy + dy), ox = ox0 + 16 * gxx + 4 * lg;
                const bool ok = co < p.Cout && oy < H && ox < W;
                const float4 y = yv[g];
#if RP_BUF
                if (ok) mi_buf_store_f32x4(obuf, (unsigned)((co * H + oy) * W + ox) * 4u, 0u, (f32x4){y.x, y.y, y.z, y.w});
#else
                if (HALF && p.out_st) {
                    typedef unsigned rp_u32x2 __attribute__((ext_vector_type(2)));
                    const uint2 q = mi_f32x4_to_bf16(y);
                    if (ok) *reinterpret_cast<mi_gptr<rp_u32x2>>(reinterpret_cast<mi_gptr<char>>(obuf) + (unsigned)((co * H + oy) * W + ox) * 2u) = (rp_u32x2){q.x, q.y};
                } else {
                    if (ok) *reinterpret_cast<mi_gptr<f32x4>>(reinterpret_cast<mi_gptr<char>>(obuf) + (unsigned)((co * H + oy) * W + ox) * 4u) = (f32x4){y.x, y.y, y.z, y.w};
                }
#endif
                const float d0 = y.x - cs_, d1 = y.y - cs_, d2 = y.z - cs_, d3 = y.w - cs_;
                csum[jt] += ok ? (d0 + d1) + (d2 + d3) : 0.0f;
                csq[jt] += ok ? fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, d3 * d3))) : 0.0f;
            }
        }
        RP_TFINE(11);
        if (p.out_stats) {
            // pixels of this tile the wave's groups cover inside the image (wave-uniform; the same for every channel)
            int wcnt = 0;
#pragma unroll
            for (int g = 0; g < GPW; ++g) {
                const int G = wave * GPW + g, gyy = G / GX, gxx = G % GX;
                const int wx = W - (ox0 + 16 * gxx), nx = wx < 0 ? 0 : (wx > 16 ? 16 : wx);
                const int ny = MODE == 2 ? (oy0 + gyy < H ? 1 : 0) : ((oy0 + 2 * gyy < H ? 1 : 0) + (oy0 + 2 * gyy + 1 < H ? 1 : 0));
                wcnt += nx * ny;
            }
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                if (MODE != 2) { csum[jt] += __shfl_xor(csum[jt], 8); csq[jt] += __shfl_xor(csq[jt], 8); }
                csum[jt] += __shfl_xor(csum[jt], 16); csq[jt] += __shfl_xor(csq[jt], 16);
                csum[jt] += __shfl_xor(csum[jt], 32); csq[jt] += __shfl_xor(csq[jt], 32);
                if (lane < CPT) {
                    mi_stat_acc a; a.c = cshift[jt]; a.s = csum[jt]; a.q = csq[jt]; a.n = wcnt;
                    mi_stat_finish(a, red[wave][2 * (CPT * jt + lane)], red[wave][2 * (CPT * jt + lane) + 1]);
                }
            }
            RP_TFINE(12);
            __syncthreads();
            if (tid < 2 * CPT * NJ && CPT * jt0 + (tid >> 1) < p.Cout)
                p.out_stats[((size_t)(b * p.Cout + CPT * jt0 + (tid >> 1)) * tiles + tile) * 2 + (tid & 1)] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        }
        RP_TFINE(13);
        RP_TPHASE(5);       // epilogue
    };
    for (int tile = tile_lo; tile + 1 < tile_hi; ++tile) do_tile(tile, std::true_type{});
    do_tile(tile_hi - 1, std::false_type{});
    RP_TEND();
}


// Wide-channel regime: per image, the per-channel affine of the fused GroupNorm -> scale/shift (or the plain input scale) and the
// power-of-two operand exponents -- what the narrow kernel's prologue computes in LDS for <= 64 channels -- into global memory.
// One workgroup per image; Cin <= MI_COEF_MAXC.
#define MI_COEF_MAXC 4096
__global__ __launch_bounds__(256) void gn_coef_kernel(const mi_conv_params p) {
    __shared__ double chS[MI_COEF_MAXC], chQ[MI_COEF_MAXC];
    __shared__ float gMean[MI_MAX_GROUPS], gRstd[MI_MAX_GROUPS];
    __shared__ float redm[4];
    __shared__ double redq[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int C0 = p.in0.C, C1 = p.in1.data ? p.in1.C : 0, Cin = C0 + C1;
    const bool gn = p.gn_groups > 0;
    const int mode = p.up2 ? 1 : (p.stride == 2 ? 2 : 0);
    const double HWs = mode == 0 ? (double)p.H * p.W : (mode == 1 ? (double)(p.H / 2) * (p.W / 2) : 4.0 * p.H * p.W);
    const bool have_stats = gn || p.in0.stats != nullptr;
    if (have_stats) mi_gn_channel_totals(p.in0, p.in1, C0, Cin, b, tid, 256, chS, chQ);
    __syncthreads();
    float m = 0.f;
    if (gn) {
        const int cpg = Cin / p.gn_groups;
        for (int g = tid; g < p.gn_groups; g += 256)
            mi_gn_group_moments(chS, chQ, g * cpg, (g + 1) * cpg, (double)cpg * HWs, p.gn_eps, gMean[g], gRstd[g]);
        __syncthreads();
        for (int c = tid; c < Cin; c += 256) {
            float An = p.gn_gamma[c], Bn = p.gn_beta[c];
            if (p.scale_shift) {
                const float* ss = p.scale_shift + (size_t)b * p.ss_stride + p.ss_off;
                const float sc = ss[c] + 1.0f, sh = ss[Cin + c];
                An *= sc;
                Bn = Bn * sc + sh;
            }
            m = fmaxf(m, 4.0f * fabsf(An) + fabsf(Bn));
        }
        m = mi_wave_max(m);
    } else if (have_stats) {
        double q = 0.0;
        for (int c = tid; c < Cin; c += 256) q += chQ[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        if (lane == 0) redq[wave] = q;
        __syncthreads();
        m = 4.0f * sqrtf((float)(((redq[0] + redq[1]) + (redq[2] + redq[3])) / ((double)Cin * HWs)));
    }
    if (lane == 0) redm[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    int ka = (m > 0.f) ? rp_clamp_exp(4 - rp_exponent(m)) : 0;
    // 1x1 residual conv input: the largest exponent that keeps it inside the fp16 range (from its statistics; 4 sigma -> [8, 16)); both
    // products go into ONE accumulator, so they must carry the same total exponent: lower the larger one (always safe)
    int kr = 0;
    if (p.res0.data && p.res_w_rp) {
        const int Cr0 = p.res0.C, Cr1 = p.res1.data ? p.res1.C : 0, Cres = Cr0 + Cr1;
        float mr = 4.0f;
        if (p.res0.stats && (Cr1 == 0 || p.res1.stats)) {
            __syncthreads();
            mi_gn_channel_totals(p.res0, p.res1, Cr0, Cres, b, tid, 256, chS, chQ);
            __syncthreads();
            double q = 0.0;
            for (int c = tid; c < Cres; c += 256) q += chQ[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
            if (lane == 0) redq[wave] = q;
            __syncthreads();
            mr = 4.0f * sqrtf((float)(((redq[0] + redq[1]) + (redq[2] + redq[3])) / ((double)Cres * (double)p.H * p.W)));
        }
        const int kr_max = (mr > 0.f) ? rp_clamp_exp(4 - rp_exponent(mr)) : 0;
        int E = ka + p.w_rp_exp;
        if (kr_max + p.res_w_rp_exp < E) E = kr_max + p.res_w_rp_exp;
        ka = E - p.w_rp_exp;
        kr = E - p.res_w_rp_exp;
    }
    for (int c = tid; c < Cin; c += 256) {
        float4 o;
        if (gn) {
            const int cpg = Cin / p.gn_groups, g = c / cpg;
            float A = gRstd[g] * p.gn_gamma[c];
            float Bc = p.gn_beta[c] - gMean[g] * A;
            if (p.scale_shift) {
                const float* ss = p.scale_shift + (size_t)b * p.ss_stride + p.ss_off;
                const float sc = ss[c] + 1.0f, sh = ss[Cin + c];
                A *= sc;
                Bc = Bc * sc + sh;
            }
            A *= (c >= C0) ? p.in1.scale : p.in0.scale;
            o = make_float4(ldexpf(A, ka), ldexpf(Bc, ka), A * -1.44269504088896340736f, Bc * -1.44269504088896340736f);
        } else {
            o = make_float4(ldexpf((c >= C0) ? p.in1.scale : p.in0.scale, ka), 0.f, 0.f, 0.f);
        }
        reinterpret_cast<float4*>(p.gn_coef)[(size_t)b * Cin + c] = o;
    }
    if (tid == 0) { p.gn_exps[2 * b] = ka; p.gn_exps[2 * b + 1] = kr; }
}

template <int TH, int TW, int NJ, bool GN, bool HALF, int MODE, int KO, int RO>
int launch_rp(const mi_conv_params& p, hipStream_t st) {
    using CFG = RpCfg<TH, TW, NJ, GN, HALF, MODE, KO, RO>;
    const int tiles = ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    // tiles per workgroup (speed only; the per-tile statistics do not depend on it): tile_cfg bits 12..15, default such that the chip
    // still gets >= ~8 workgroups per CU
    int ntile = (p.tile_cfg >> 12) & 0xf;
    // strip length: up to 4 tiles per workgroup while that leaves >= 1024 workgroups (one full wave of 4 per CU); measured on the SR
    // U-Net's shapes (tools/gpu_rp_shapes.sh): 256^2 -> 4, 128^2 -> 2, <= 64^2 -> 1
    if (ntile == 0) {
        ntile = 1; while (ntile < 4 && (size_t)p.B * (tiles / (2 * ntile)) >= 1024) ntile *= 2;
    }
    const int strips = (tiles + ntile - 1) / ntile;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_rp_kernel<CFG>), dim3(strips * p.B), dim3(256), 0, st, p, (const uint4*)p.w_rp, (const uint4*)p.res_w_rp, ntile,
                       (const float4*)nullptr, (const int*)nullptr);
    return mi_check_launch("conv_rp_kernel");
}

// wide-channel regime: 8 x 32 (tile_cfg 7), 8 x 64 (tile_cfg 6) or 16 x 16 (tile_cfg 10; both MODE 0 only) pixel tiles, NJ N tiles per workgroup, the layer's
// output channels over blockIdx.y
template <int TWW, int NJ, bool GN, bool HALF, int MODE, int THH = 8>
int launch_rp_wide(const mi_conv_params& p, hipStream_t st) {
    using CFG = RpCfg<THH, TWW, NJ, GN, HALF, MODE, -1, -1, true>;
    const int tiles = ((p.H + THH - 1) / THH) * ((p.W + TWW - 1) / TWW);
    const int cpt = MODE == 2 ? 16 : 8, njt = (p.Cout + cpt - 1) / cpt;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_rp_kernel<CFG>), dim3(tiles * p.B, (njt + NJ - 1) / NJ), dim3(256), 0, st, p, (const uint4*)p.w_rp,
                       (const uint4*)p.res_w_rp, 1, (const float4*)p.gn_coef, (const int*)p.gn_exps);
    return mi_check_launch("conv_rp_kernel (wide)");
}

template <int MODE>
int launch_rp_wide_m(const mi_conv_params& p, hipStream_t st) {
    const bool half = (p.tile_cfg & MI_CONV_HALF) != 0;
    const int cpt = MODE == 2 ? 16 : 8, njt = (p.Cout + cpt - 1) / cpt;
    if constexpr (MODE == 0) {
        if ((p.tile_cfg & 0xff) == 6 && njt >= 4 && !half) {          // 8 x 64 tiles: the B fragments of a round serve twice the pixels
            if (p.gn_groups > 0) return launch_rp_wide<64, 4, true, false, MODE>(p, st);
            return launch_rp_wide<64, 4, false, false, MODE>(p, st);
        }
        if ((p.tile_cfg & 0xff) == 10 && njt >= 4 && !half) {         // 16 x 16 tiles: images no wider than 16 (half of an 8 x 32 tile would hang over the edge)
            if (p.gn_groups > 0) return launch_rp_wide<16, 4, true, false, MODE, 16>(p, st);
            return launch_rp_wide<16, 4, false, false, MODE, 16>(p, st);
        }
        if (p.gn_groups > 0) {
            if (njt >= 4) return half ? launch_rp_wide<32, 4, true, true, MODE>(p, st) : launch_rp_wide<32, 4, true, false, MODE>(p, st);
            return half ? launch_rp_wide<32, 1, true, true, MODE>(p, st) : launch_rp_wide<32, 1, true, false, MODE>(p, st);
        }
    }
    if (njt >= 4 && MODE != 2) return half ? launch_rp_wide<32, 4, false, true, MODE>(p, st) : launch_rp_wide<32, 4, false, false, MODE>(p, st);
    if (njt >= 2 && MODE == 2) return half ? launch_rp_wide<32, 2, false, true, MODE>(p, st) : launch_rp_wide<32, 2, false, false, MODE>(p, st);
    return half ? launch_rp_wide<32, 1, false, true, MODE>(p, st) : launch_rp_wide<32, 1, false, false, MODE>(p, st);
}

template <int TH, int TW, int NJ, int MODE, int KO, int RO>
int launch_rp_v(const mi_conv_params& p, hipStream_t st) {
    const bool half = (p.tile_cfg & MI_CONV_HALF) != 0;
    if constexpr (MODE == 0) {
        if (p.gn_groups > 0) return half ? launch_rp<TH, TW, NJ, true, true, MODE, KO, RO>(p, st) : launch_rp<TH, TW, NJ, true, false, MODE, KO, RO>(p, st);
    }
    if constexpr ((KO <= 1 && RO <= 0) || MODE != 0)        // without GroupNorm: the final conv (8 channels in), the up / down-sampling convs and the generic fall-back only
        return half ? launch_rp<TH, TW, NJ, false, true, MODE, KO, RO>(p, st) : launch_rp<TH, TW, NJ, false, false, MODE, KO, RO>(p, st);
    return launch_rp_v<TH, TW, NJ, MODE, -1, -1>(p, st);
}

// the layer shapes of the BASELINE U-Nets get a kernel with a compile-time round structure; everything else the generic one
template <int TH, int TW, int NJ, int MODE>
int launch_rp_kr(const mi_conv_params& p, hipStream_t st) {
    const int ko = (p.in0.C + (p.in1.data ? p.in1.C : 0)) >> 3;
    const int ro = (p.res0.data && p.res_w_rp) ? (p.res0.C + (p.res1.data ? p.res1.C : 0)) >> 3 : 0;
    if constexpr (MODE != 0 && NJ == 1) {
        if (ko == 1 && ro == 0) return launch_rp_v<TH, TW, NJ, MODE, 1, 0>(p, st);
        if (ko == 2 && ro == 0 && MODE == 1) return launch_rp_v<TH, TW, NJ, MODE, 2, 0>(p, st);
    }
    if constexpr (MODE == 0 && NJ == 1) {
        if (ko == 1 && ro == 0) return launch_rp_v<TH, TW, NJ, MODE, 1, 0>(p, st);
        if (ko == 2 && ro == 0) return launch_rp_v<TH, TW, NJ, MODE, 2, 0>(p, st);
        if (ko == 1 && ro == 2) return launch_rp_v<TH, TW, NJ, MODE, 1, 2>(p, st);
    }
    if constexpr (MODE == 0 && NJ == 2) {
        if (ko == 1 && ro == 0) return launch_rp_v<TH, TW, NJ, MODE, 1, 0>(p, st);
        if (ko == 2 && ro == 0) return launch_rp_v<TH, TW, NJ, MODE, 2, 0>(p, st);
        if (ko == 4 && ro == 0) return launch_rp_v<TH, TW, NJ, MODE, 4, 0>(p, st);
        if (ko == 2 && ro == 4) return launch_rp_v<TH, TW, NJ, MODE, 2, 4>(p, st);
        if (ko == 3 && ro == 0) return launch_rp_v<TH, TW, NJ, MODE, 3, 0>(p, st);      // 16 + 8 skip channels (base U-Net, 32^2 level)
        if (ko == 2 && ro == 3) return launch_rp_v<TH, TW, NJ, MODE, 2, 3>(p, st);
    }
    return launch_rp_v<TH, TW, NJ, MODE, -1, -1>(p, st);
}

template <int TH, int TW, int MODE>
int launch_rp_nj(const mi_conv_params& p, hipStream_t st) {
    const int nj = MODE == 2 ? (p.Cout + 15) / 16 : (p.Cout + 7) / 8;
    switch (nj) {
        case 1: return launch_rp_kr<TH, TW, 1, MODE>(p, st);
        case 2:
            if constexpr (TH * TW <= 512 && MODE == 0) return launch_rp_kr<TH, TW, 2, MODE>(p, st);
            break;
        case 3: case 4:
            if constexpr (MODE == 0 && TH * TW <= 256) return launch_rp_kr<TH, TW, 4, MODE>(p, st);
            break;
    }
    mi_set_error("mi_conv_fwd: row-paired path: %d output channels not instantiated for this tile", p.Cout);
    return MI_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int mi_gn_coef_fwd(const mi_conv_params* pp, void* stream) {
    const mi_conv_params& p = *pp;
    const int Cin = p.in0.C + (p.in1.data ? p.in1.C : 0);
    if (!p.gn_coef || !p.gn_exps || p.B <= 0) { mi_set_error("mi_gn_coef_fwd: gn_coef / gn_exps buffers missing or empty batch"); return MI_ERR_INVALID; }
    if (Cin > MI_COEF_MAXC || p.gn_groups > MI_MAX_GROUPS || (p.gn_groups > 0 && Cin % p.gn_groups)) { mi_set_error("mi_gn_coef_fwd: %d channels / %d groups unsupported", Cin, p.gn_groups); return MI_ERR_UNSUPPORTED; }
    if (p.gn_groups > 0 && (!p.in0.stats || (p.in1.data && !p.in1.stats))) { mi_set_error("mi_gn_coef_fwd: GroupNorm input without channel statistics"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(gn_coef_kernel, dim3(p.B), dim3(256), 0, (hipStream_t)stream, p);
    return mi_check_launch("gn_coef_kernel");
}

int mi_conv_rp_launch(const mi_conv_params& p, hipStream_t st) {
    const int C0 = p.in0.C, C1 = p.in1.data ? p.in1.C : 0, Cin = C0 + C1;
    const int mode = p.up2 ? 1 : (p.stride == 2 ? 2 : 0);
    if (!((p.ksize == 3 && p.stride == 1) || (p.ksize == 4 && p.stride == 2 && !p.up2))) { mi_set_error("mi_conv_fwd: row-paired path is k3 s1 (optionally nearest x2) or k4 s2"); return MI_ERR_UNSUPPORTED; }
    if (mode != 0 && p.res0.data) { mi_set_error("mi_conv_fwd: row-paired path: no residual on the up / down-sampling convs"); return MI_ERR_UNSUPPORTED; }
    if (mode == 1 && ((p.H | p.W) & 1)) { mi_set_error("mi_conv_fwd: up2 needs even output size"); return MI_ERR_INVALID; }
    const bool wide = p.gn_coef != nullptr;
    if ((p.W & 3) || (C0 & 7) || (C1 & 7) || Cin > (wide ? MI_COEF_MAXC : RP_MAXC)) { mi_set_error("mi_conv_fwd: row-paired path needs W %% 4 == 0 and channel counts in multiples of 8 up to %d", wide ? MI_COEF_MAXC : RP_MAXC); return MI_ERR_UNSUPPORTED; }
    if (mode != 0 && p.gn_groups > 0) { mi_set_error("mi_conv_fwd: GroupNorm prologue is only built for the k3 s1 family"); return MI_ERR_UNSUPPORTED; }
    if (p.res0.data && p.res_w) {
        const int Cres = p.res0.C + (p.res1.data ? p.res1.C : 0);
        if (!p.res_w_rp || (p.res0.C & 7) || (p.res1.data && (p.res1.C & 7)) || Cres > (wide ? MI_COEF_MAXC : RP_MAXC)) { mi_set_error("mi_conv_fwd: row-paired path needs res_w_rp and residual channels in multiples of 8"); return MI_ERR_INVALID; }
    }
    const size_t biggest = (size_t)(C0 > C1 ? C0 : C1) * p.H * p.W * (mode == 2 ? 4 : 1);
    if (biggest >= (1ull << 31)) { mi_set_error("mi_conv_fwd: row-paired path indexes one image with 32-bit offsets"); return MI_ERR_UNSUPPORTED; }
    if (wide) {
        if (!p.gn_exps) { mi_set_error("mi_conv_fwd: wide regime needs gn_exps (mi_gn_coef_fwd first)"); return MI_ERR_INVALID; }
        const int tc = p.tile_cfg & 0xff;
        if (tc != 7 && !((tc == 6 || tc == 10) && mode == 0)) { mi_set_error("mi_conv_fwd: the wide regime uses tile_cfg 7 (8x32), or 6 (8x64) / 10 (16x16) for the k3 s1 member"); return MI_ERR_INVALID; }
        return mode == 0 ? launch_rp_wide_m<0>(p, st) : (mode == 1 ? launch_rp_wide_m<1>(p, st) : launch_rp_wide_m<2>(p, st));
    }
    if (mode == 1) {
        if ((p.tile_cfg & 0xff) == 6) return launch_rp_nj<8, 64, 1>(p, st);
        mi_set_error("mi_conv_fwd: row-paired up-sampling conv uses tile_cfg 6 (8x64)");
        return MI_ERR_INVALID;
    }
    if (mode == 2) {
        if ((p.tile_cfg & 0xff) == 7) return launch_rp_nj<8, 32, 2>(p, st);
        mi_set_error("mi_conv_fwd: row-paired stride-2 conv uses tile_cfg 7 (8x32)");
        return MI_ERR_INVALID;
    }
    switch (p.tile_cfg & 0xff) {
        case 6: return launch_rp_nj<8, 64, 0>(p, st);
        case 7: return launch_rp_nj<8, 32, 0>(p, st);
    }
    mi_set_error("mi_conv_fwd: row-paired path uses tile_cfg 6 (8x64) or 7 (8x32) (5, 16x64 tiles, was removed in round 6: measured slower, profiles/r05_summary.md r05l)");
    return MI_ERR_INVALID;
}
