// Resident conv chain: a run of consecutive Block / ResnetBlock convolutions of one U-Net level (layers.py:131-145, 417-439; the
// <= 64 x 64 levels of Unet.forward, Unet.py:419-465) in ONE launch.
//
// Why: at <= 64^2 every conv launch of the row-paired kernel is one round of one-tile workgroups whose phases (issue loads -> statistics ->
// transform -> MFMA -> epilogue) serialise: 17.7 .. 19.6 us for 33 MB of traffic and ~2 us of matrix-core work (profiles/r03_summary.md),
// 17 such launches per super-resolution step and all 32 launches of a base step.  Here the activation never leaves the chip between two
// layers: every image is cut into horizontal slabs (16 rows of a 64-wide image, 8 rows of a 32-wide one), one 512-thread workgroup per
// slab keeps its slab of the running tensor in REGISTERS (the accumulators of the last conv), transforms it (GroupNorm affine ->
// scale/shift -> SiLU -> fp16 hi/lo split) straight into the LDS operand planes of the next conv and multiplies on the matrix cores with
// the row-paired scheme of conv_rp.hip (N = 2 output rows x 8 channels, K = 4 input rows x 8 channels per horizontal tap, 3-term fp16
// splits with power-of-two operand scalings).  The MFMA operands are swapped against conv_rp.hip (D^T = W^T A^T), so a lane ends up with
// FOUR CONSECUTIVE CHANNELS of one pixel -- exactly an 8-byte piece of the next layer's pixel-major operand chunk: no transposition.
//
// What the workgroups of an image must exchange between two layers is small: (a) per-channel partial (sum, sum of squares) of their slab
// (the next GroupNorm needs whole-image moments; 128 bytes), (b) their first and last row (the 3x3 halo of the neighbours; 2 x 4 KB).
// Both travel as 8-byte {tag, value} GRANULES (two per 16-byte write-through sc1 store; tag = launch number x 64 + layer + 1): the data is
// its own flag -- no release fence, no drain, no separate flag word; the consumer re-reads its granules with L1-bypassing (sc1) loads until
// every tag matches.  Valid under any workgroup -> XCD placement (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement &
// inter-workgroup visibility", form R2).  Exchange buffers are double-buffered by layer parity, and every layer waits for the partials
// of ALL slabs of its image, which is what makes two buffers enough.
// Co-residency: a workgroup takes its (image, slab) from a ticket counter AFTER it has become resident, so the slabs of an image belong to
// workgroups that are resident at the same time by construction -- whatever else runs on the GPU, with any number of such launches in
// flight on other streams, and without a cooperative launch: only the highest, still incomplete ticket group can wait for workgroups
// that have not started, and those start as soon as any workgroup of any kernel retires.  Every spin is bounded (error word in `sync`).
#include "common.hip.h"
#include <type_traits>

typedef _Float16 rs_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 rs_f16x2 __attribute__((ext_vector_type(2)));
typedef float rs_f32x2 __attribute__((ext_vector_type(2)));

#ifdef RS_TRACE
// development aid (tools/bench_resident.py with the -DRS_TRACE build): 100 MHz wall-clock stamps at the phase boundaries of every layer
// of the first 256 workgroups (by ticket) of the last launch
__device__ unsigned long long rs_trace_buf[256 * MI_RES_MAX_LAYERS * 16];
extern "C" int mi_debug_read_trace_rs(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(rs_trace_buf), bytes); }
__device__ unsigned long long rs_trace_w[64 * MI_RES_MAX_LAYERS * 16 * 4];
extern "C" int mi_debug_read_trace_rs_w(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(rs_trace_w), bytes); }
#define RS_WSTAMP(k) do { if (lane == 0 && local < 64) rs_trace_w[((local * MI_RES_MAX_LAYERS + li) * 16 + wave) * 4 + (k)] = wall_clock64(); } while (0)
#define RS_STAMP(k) do { if (tid == 0 && local < 256) rs_trace_buf[(local * MI_RES_MAX_LAYERS + li) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define RS_STAMP(k) do { } while (0)
#define RS_WSTAMP(k) do { } while (0)
#endif

namespace {

constexpr int RS_MAXCIN = 64;           // channels of a conv input / of a 1x1-residual input (after concat)
constexpr int RS_MAXS = 8;              // slabs per image
constexpr unsigned RS_SPIN_LIMIT = 1u << 22;

struct rs_layout { long long xstats, xhalo, total; };
__host__ __device__ inline rs_layout rs_sync_layout(int B, int S, int W) {
    rs_layout l;
    l.xstats = 64;                                                                          // [0] ticket counter (u64), [8] error word (u32)
    l.xhalo = l.xstats + (long long)2 * B * S * MI_RES_MAXC * 16;                           // [parity][B][S][16 ch] x {tag, sum, tag, sumsq}
    l.total = l.xhalo + (long long)2 * B * S * 2 * W * (MI_RES_MAXC / 2) * 16;              // [parity][B][S][side][W][8 pairs] x {tag, v, tag, v}
    return l;
}

__device__ __forceinline__ void rs_split8(const float (&y)[8], uint4& hi, uint4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const rs_f32x2 v = {y[2 * i], y[2 * i + 1]};
        const rs_f16x2 h2 = __builtin_convertvector(v, rs_f16x2);
        h[i] = __builtin_bit_cast(unsigned, h2);
        l[i] = mi_split_lo2(h[i], y[2 * i], y[2 * i + 1]);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ int rs_exponent(float m) {
    const unsigned u = __float_as_uint(m) & 0x7fffffffu;
    const int be = (int)(u >> 23);
    if (be == 0 || be == 255) return 0;
    return be - 126;
}
__device__ __forceinline__ int rs_clamp_exp(int k) { return k < -60 ? -60 : (k > 60 ? 60 : k); }

enum { RS_CUR = 0, RS_XR = 1, RS_GLB = 2 };

// waves per SIMD the register allocation must leave room for (2 = all 512 registers of a SIMD lane for the two waves of one workgroup).
// RS_LEAN=1: the light kernels (8 output channels, 32-wide images) at 128 registers, so that another stream's kernels can share the CU
#ifndef RS_LEAN
#define RS_LEAN 0
#endif
#define RS_WAVES_PER_EU(TW_, R_, NJ_) (((TW_) == 64 && (R_) == 8) ? 4 : ((RS_LEAN && ((NJ_) == 1 || (TW_) == 32)) ? 4 : 2))

// TW: image width (a slab spans it), R: rows per slab, NW: waves per workgroup, NJ: N tiles of 8 output channels the accumulators are
// sized for (NJX: every layer has exactly 8 NJ output channels), HALF: one fp16 term per product (tensors with st = 1 are bf16 in memory)
template <int TW, int R, int NW, int NJ, bool NJX, bool HALF>
__global__ __launch_bounds__(64 * NW, RS_WAVES_PER_EU(TW, R, NJ)) void resident_convs_kernel(const mi_resident_params p) {
    constexpr int T = 64 * NW;
    constexpr int IH = R + 2, PW = TW + 8, PLANE = IH * PW;                  // staged window (rows), LDS pitch / plane size in 16-byte chunks
    constexpr int GX = TW / 16, GY = R / 2, NG = GX * GY, GPW = NG / NW;     // 16-pixel x 2-row groups; per wave
    static_assert(NG % NW == 0 && GPW >= 1, "slab must split over the waves");
    constexpr int MPER = (R * TW + T - 1) / T;                               // staging units (one pixel x 8 channels) of a tensor in global memory:
    constexpr int UPER = MPER + 1;                                           // the slab's own rows, + one unit of the two halo rows
    static_assert(2 * TW <= T && 4 * TW <= T, "halo rows: one unit / one 4-channel piece per work-item");
    constexpr int WSLOT = 3 * NJ * 128;                                      // weight chunks of one octet (3 taps x NJ x 128 lanes x {hi, lo}); two octets per pass
    constexpr int HWV = 4 * TW / 64;                                         // waves that own a halo piece
    __shared__ __attribute__((aligned(16))) uint4 actH[2 * PLANE];
    __shared__ __attribute__((aligned(16))) uint4 actL[HALF ? 1 : 2 * PLANE];
    __shared__ __attribute__((aligned(16))) uint4 wl[2 * WSLOT];
    __shared__ __attribute__((aligned(8))) float2 chP[RS_MAXCIN];            // per input channel: (A 2^ka, B 2^ka) of y = SiLU(A x + B) 2^ka
    __shared__ double chS[RS_MAXCIN], chQ[RS_MAXCIN], chQ2[RS_MAXCIN], xQ[MI_RES_MAXC];
    __shared__ float red[NW][NJ * 16];
    __shared__ int sExp[4];
    __shared__ float sCexp;
    __shared__ mi_u64 sTicket;
    __shared__ int sAbort;

    // Work-item coordinates.  They are re-derived from an OPAQUE copy of the work-item id at every phase of every layer (derive()): the
    // compiler otherwise hoists the ~150 loop-invariant LDS / global offsets of all phases out of the layer loop and spills them
    // (measured: 650 bytes of scratch per work-item at 256 registers).
    int tid, wave, lane, lq, lg;
    int dyL, cq;                                               // this lane's output row inside a group, and channel quad inside an octet
    int perm;                                                  // lane group -> input row of the K block (0, 2, 1, 3), as conv_rp.hip
    int gyy[GPW], gxx[GPW];
    int hside, hx, hcq;                                        // halo piece of this work-item (waves < HWV): 4 channels (quad hcq of every octet) of pixel hx of the row above / below the slab
    auto derive = [&]() {
        int t = threadIdx.x;
        MI_OPAQUE(t);
        tid = t; wave = t >> 6; lane = t & 63; lq = lane & 15; lg = lane >> 4;
        dyL = lg >> 1; cq = lg & 1; perm = ((lg & 1) << 1) | (lg >> 1);
#pragma unroll
        for (int g = 0; g < GPW; ++g) { const int G = wave * GPW + g; gyy[g] = G / GX; gxx[g] = G % GX; }
        hside = t / (2 * TW); hx = (t >> 1) % TW; hcq = t & 1;
    };
    derive();
    const int H = p.H, W = p.W, S = H / R, HW = H * W;
    char* const sync = reinterpret_cast<char*>(p.sync);
    const rs_layout lay = rs_sync_layout(p.B, S, W);

    if (tid == 0) { sTicket = mi_agent_add_u64(reinterpret_cast<mi_u64*>(sync), 1ull); sAbort = 0; }
    for (int i = tid; i < 2 * PLANE; i += T) {                 // the zero padding (columns 0 / TW + 1, rows outside the image) is never written again
        actH[i] = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (!HALF) actL[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    // (image, slab) of this workgroup: consecutive tickets = the slabs of one image; launch number = ticket / grid size (launches on one
    // `sync` are serialised by their stream), so the tags of successive launches differ and nothing ever needs a reset
    const mi_u64 tk = sTicket, Gt = (mi_u64)p.B * (mi_u64)S, seq = tk / Gt;
    const int local = (int)(tk - seq * Gt), b = local / S, s = local - b * S;
    const mi_buf sbuf = mi_make_buf(sync + lay.xstats);
    const mi_buf hbuf = mi_make_buf(sync + lay.xhalo);
    auto fail = [&](unsigned code) {                            // one lane: a neighbour never showed up -- give up instead of hanging the GPU
        mi_agent_store_u32(reinterpret_cast<unsigned*>(sync + 8), code);
        sAbort = 1;
    };


    f32x4 cur[GPW][NJ], X[GPW][NJ];
#pragma unroll
    for (int g = 0; g < GPW; ++g)
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) { cur[g][jt] = (f32x4){0.f, 0.f, 0.f, 0.f}; X[g][jt] = cur[g][jt]; }
    int Ccur = 0, Cx = 0;
    bool prev_saved = false;
    // Weight fragments never touch a register: LDS-DMA (global_load_lds_dwordx4, 1 KB per instruction) of the two octets of one pass into
    // the two weight slots.  Pass 0 of a layer is requested right after the previous layer's epilogue barrier, pass n + 1 right after the
    // barrier that ends pass n's matrix-core loop -- when no wave reads the old ones any more -- and awaited (vmcnt) at the pass's
    // staging barrier, i.e. it flies under the operand staging.
    auto weights_dma = [&](const mi_res_layer& Ln, int ccur, int cx, int q0) {
        const int nj_ = (Ln.Cout + 7) >> 3;
        const int ko = ((Ln.src == 0 ? ccur : Ln.in0.C) >> 3) + (Ln.in1.data ? (Ln.in1.C >> 3) : 0);
        const int ro = Ln.res >= 3 ? (((Ln.res == 3 ? cx : Ln.res0.C) >> 3) + (Ln.res1.data ? (Ln.res1.C >> 3) : 0)) : 0;
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {
            const int q = q0 + slot;
            if (q >= ko + ro) continue;
            const uint4* src = q < ko ? reinterpret_cast<const uint4*>(Ln.w_rp) + (size_t)q * 3 * nj_ * 128 : reinterpret_cast<const uint4*>(Ln.res_w_rp) + (size_t)(q - ko) * nj_ * 128;
            const int rows = (q < ko ? 3 : 1) * nj_ * 2;                           // rows of 64 chunks
            for (int r = wave; r < rows; r += NW) {
#if defined(HIPEMU)
                wl[slot * WSLOT + r * 64 + lane] = src[r * 64 + lane];
#else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + r * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)&wl[slot * WSLOT + r * 64], 16, 0, 0);
#endif
            }
        }
    };
    weights_dma(p.layer[0], 0, 0, 0);

    for (int li = 0; li < p.n_layers; ++li) {
        const mi_res_layer& L = p.layer[li];
        const int Cout = L.Cout, njl = (Cout + 7) >> 3;
        const int nA = (L.src == 0 ? Ccur : L.in0.C) >> 3, nB = L.in1.data ? (L.in1.C >> 3) : 0, KO = nA + nB, Cin = 8 * KO, C0 = 8 * nA;
        const bool rconv = L.res >= 3;
        const int rA = rconv ? ((L.res == 3 ? Cx : L.res0.C) >> 3) : 0, rB = (rconv && L.res1.data) ? (L.res1.C >> 3) : 0;
        const int RO = rA + rB, Cres = 8 * RO, Cr0 = 8 * rA, nq = KO + RO;
        const bool gn = L.gn_groups > 0;
        const int par_in = (li - 1) & 1, par_out = li & 1;
        const unsigned tag_in = (unsigned)(seq * 64 + (mi_u64)li), tag_out = tag_in + 1u;       // never 0: (li + 1) % 64 != 0

        // octet q of the layer's operand sequence (conv octets, then 1x1-residual octets): where it comes from
        auto oct_kind = [&](int q) -> int {
            if (q < KO) return q < nA ? (L.src == 0 ? RS_CUR : RS_GLB) : RS_GLB;
            return (q - KO) < rA ? (L.res == 3 ? RS_XR : RS_GLB) : RS_GLB;
        };
        auto oct_tensor = [&](int q, int& c0) -> const mi_act& {
            if (q < KO) { if (q < nA) { c0 = 8 * q; return L.in0; } c0 = 8 * (q - nA); return L.in1; }
            const int r = q - KO;
            if (r < rA) { c0 = 8 * r; return L.res0; }
            c0 = 8 * (r - rA);
            return L.res1;
        };

        derive();
        RS_STAMP(0);
        // ---------------- (1) everything that does not depend on the neighbours: weights, affine parameters, bias, statistics and the
        // first pass's tensors in global memory
        float pg = 0.f, pb = 0.f, psc = 1.f, psh = 0.f;
        if (gn && wave == 0) {
            const int c = lane < Cin ? lane : 0;
            pg = L.gn_gamma[c];
            pb = L.gn_beta[c];
            if (L.ss_off >= 0) {
                const float* ss = p.scale_shift + (size_t)b * p.ss_stride + L.ss_off;
                psc = ss[c] + 1.0f;
                psh = ss[Cin + c];
            }
        }
        float bv[NJ][4];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = 8 * jt + 4 * cq + r;
                float v = (L.bias && ch < Cout) ? L.bias[ch] : 0.0f;
                if (rconv && L.res_b && ch < Cout) v += L.res_b[ch];
                bv[jt][r] = v;
            }
        // channel totals of the tensors in global memory: one work-item per channel adds the producer's partials in tile order (fp64)
        {
            auto totals = [&](const mi_act& a, int t0, int coff, double* Sd, double* Qd) {
                const int c = tid - t0;
                if (c < 0 || c >= a.C || !a.stats) return;
                const float* st = a.stats + ((size_t)(mi_row_of(b, a.bmod) * a.C + c) * a.nt) * 2;
                double sm = 0.0, qm = 0.0;
                for (int t = 0; t < a.nt; ++t) { sm += (double)st[2 * t]; qm += (double)st[2 * t + 1]; }
                if (Sd) Sd[coff + c] = sm * (double)a.scale;
                Qd[coff + c] = qm * (double)a.scale * (double)a.scale;
            };
            if (L.src != 0) totals(L.in0, 64, 0, chS, chQ);
            if (nB) totals(L.in1, 128, C0, chS, chQ);
            if (L.res == 4) totals(L.res0, 192, 0, nullptr, chQ2);
            if (rB) totals(L.res1, 256, Cr0, nullptr, chQ2);
        }
        float raw0[UPER][8], raw1[UPER][8];
        bool st0 = false, st1 = false;
        // unit u < MPER: pixel (tid + u T) of the slab's own rows; unit MPER: pixel tid of the two halo rows (work-items < 2 TW)
        auto unit_pos = [&](int u, int& iy, int& x) -> bool {
            if (u < MPER) { const int qd = tid + u * T; iy = 1 + qd / TW; x = qd % TW; return qd < R * TW; }
            iy = (tid / TW) ? R + 1 : 0;
            x = tid % TW;
            return tid < 2 * TW;
        };
        auto issue_global = [&](int q, float (&raw)[UPER][8], bool& s16) {
            int c0;
            const mi_act& t = oct_tensor(q, c0);
            const size_t e0 = ((size_t)mi_row_of(b, t.bmod) * t.C + c0) * HW;
            s16 = HALF && t.st != 0;
            const mi_gptr<const float> base = mi_global(t.data + (s16 ? 0 : e0));
            const mi_gptr<const unsigned short> base16 = mi_global(reinterpret_cast<const unsigned short*>(t.data) + e0);
            unsigned off[UPER];
#pragma unroll
            for (int u = 0; u < UPER; ++u) {
                int iy, x;
                const bool live = unit_pos(u, iy, x);
                const int gy = s * R - 1 + iy;
                off[u] = (live && gy >= 0 && gy < H) ? (unsigned)(gy * W + x) : 0u;
            }
            // channel planes through a uniform base: one 32-bit offset register per unit, no 64-bit address arithmetic per load
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int u = 0; u < UPER; ++u) {
                    if (HALF && s16) raw[u][j] = __uint_as_float((unsigned)*reinterpret_cast<mi_gptr<const unsigned short>>(reinterpret_cast<mi_gptr<const char>>(base16 + (size_t)j * HW) + 2u * off[u]));
                    else raw[u][j] = *reinterpret_cast<mi_gptr<const float>>(reinterpret_cast<mi_gptr<const char>>(base + (size_t)j * HW) + 4u * off[u]);
                }
        };
        // raw slot k belongs to LDS slot k of the next pass: pass 0's tensors from global memory are requested here (they fly while the
        // neighbours' partials are polled), pass n + 1's right before the matrix-core loop of pass n
        auto issue_pass = [&](int q0) {
            if (q0 < nq && oct_kind(q0) == RS_GLB) issue_global(q0, raw0, st0);
            if (q0 + 1 < nq && oct_kind(q0 + 1) == RS_GLB) issue_global(q0 + 1, raw1, st1);
        };
        issue_pass(0);

        RS_STAMP(1);
        // ---------------- (2) the resident tensor's channel totals: every slab's partial granules, re-read until their tags match
        if (L.src == 0 && wave == 0) {
            const int c = lane < Ccur ? lane : 0;
            double sm = 0.0, qm = 0.0;
            for (unsigned spins = 0;;) {
                f32x4 g[RS_MAXS];
#pragma unroll
                for (int sl = 0; sl < RS_MAXS; ++sl) {
                    const int slc = sl < S ? sl : 0;
                    g[sl] = mi_buf_load_f32x4_sc1(sbuf, (unsigned)((((size_t)par_in * p.B + b) * S + slc) * MI_RES_MAXC + c) * 16u);
                }
                bool ok = true;
                sm = 0.0; qm = 0.0;
#pragma unroll
                for (int sl = 0; sl < RS_MAXS; ++sl) {
                    if (sl < S) {
                        ok = ok && __float_as_uint(g[sl][0]) == tag_in && __float_as_uint(g[sl][2]) == tag_in;
                        sm += (double)g[sl][1];
                        qm += (double)g[sl][3];
                    }
                }
                if (__all(ok)) break;
                if (++spins > RS_SPIN_LIMIT) { if (lane == 0) fail(0x100u + (unsigned)li); break; }
                mi_sleep();
            }
            if (lane < Ccur) {
                chS[lane] = sm;
                chQ[lane] = qm;
                if (prev_saved) xQ[lane] = qm;
            }
        }
        __syncthreads();
        if (sAbort) return;
        derive();
        RS_STAMP(2);
        // ---------------- (3) GroupNorm moments, per-channel affine, power-of-two operand scalings (as conv_rp.hip), by one wave
        if (wave == 0) {
            const int c = lane;
            const float insc = (c >= C0) ? L.in1.scale : (L.src == 0 ? 1.0f : L.in0.scale);
            float A = 0.f, Bc = 0.f, m = 0.f;
            if (gn) {
                if (c < Cin) {
                    const int cpg = Cin / L.gn_groups, g0 = (c / cpg) * cpg;
                    double sg = 0.0, qg = 0.0;
                    for (int k = 0; k < cpg; ++k) { sg += chS[g0 + k]; qg += chQ[g0 + k]; }
                    const double inv_n = 1.0 / ((double)cpg * (double)HW);
                    const double mean = sg * inv_n;
                    double var = qg * inv_n - mean * mean;
                    var = var > 0.0 ? var : 0.0;
                    const float gmean = (float)mean, grstd = 1.0f / sqrtf((float)(var + (double)L.gn_eps));
                    float An = pg, Bn = pb;
                    A = grstd * pg;
                    Bc = pb - gmean * A;
                    if (L.ss_off >= 0) {
                        A *= psc;
                        Bc = Bc * psc + psh;
                        An *= psc;
                        Bn = Bn * psc + psh;
                    }
                    A *= insc;
                    m = 4.0f * fabsf(An) + fabsf(Bn);
                }
                m = mi_wave_max(m);
            } else {
                double q = (c < Cin) ? chQ[c] : 0.0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
                m = 4.0f * sqrtf((float)(q / ((double)Cin * (double)HW)));
            }
            int ka = (m > 0.f) ? rs_clamp_exp(4 - rs_exponent(m)) : 0;
            int E = ka + L.w_rp_exp, kr = 0;
            if (RO > 0) {
                double q = (c < Cres) ? ((L.res == 3 && c < Cr0) ? xQ[c] : chQ2[c]) : 0.0;      // X's totals were kept when its successor read them
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
                const float mr = 4.0f * sqrtf((float)(q / ((double)Cres * (double)HW)));
                const int kr_max = (mr > 0.f) ? rs_clamp_exp(4 - rs_exponent(mr)) : 0;
                const int Er = kr_max + L.res_w_rp_exp;
                if (Er < E) E = Er;
                ka = E - L.w_rp_exp;
                kr = E - L.res_w_rp_exp;
            }
            // one branch-free form for every operand: y = a / (1 + exp2(a c)), a = P.x x + P.y.  GroupNorm + SiLU: P = (A, B) 2^ka,
            // c = -log2(e) 2^-ka.  Plain scaling (no GroupNorm; 1x1-residual operands): P = (2 s, 0), c = 0 -> y = 2 s x / (1 + 1) = s x
            if (c < Cin) chP[c] = gn ? make_float2(ldexpf(A, ka), ldexpf(Bc, ka)) : make_float2(ldexpf(insc, ka + 1), 0.f);
            if (lane == 0) { sExp[0] = ka; sExp[1] = kr; sExp[2] = E; sCexp = gn ? ldexpf(-1.44269504088896340736f, -ka) : 0.0f; }
        }
        __syncthreads();
        derive();
        RS_STAMP(3);

        // ---------------- (4) operand staging + matrix-core passes, two octets at a time
        const float cexp = sCexp, rsc_x2 = ldexpf(1.0f, sExp[1] + 1);
        auto act1 = [&](float x, float2 P, float c) -> float {
            const float a = fmaf(x, P.x, P.y);
            return a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a * c));
        };
        auto put4 = [&](const float (&y)[4], int chunk, int half8) {      // 4 channels of one pixel -> 8 bytes of its hi (and lo) chunk
            mi_f16x4 hi, lo;
            if constexpr (HALF) {
                const rs_f32x2 v0 = {y[0], y[1]}, v1 = {y[2], y[3]};
                const rs_f16x2 h0 = __builtin_convertvector(v0, rs_f16x2), h1 = __builtin_convertvector(v1, rs_f16x2);
                hi = __builtin_bit_cast(mi_f16x4, make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)));
                lo = hi;
            } else {
                mi_split_f16(y, hi, lo);
            }
            reinterpret_cast<uint2*>(actH)[2 * chunk + half8] = __builtin_bit_cast(uint2, hi);
            if constexpr (!HALF) reinterpret_cast<uint2*>(actL)[2 * chunk + half8] = __builtin_bit_cast(uint2, lo);
        };
        // a tensor held in registers (the running activation, or X for a 1x1 residual conv): own rows straight from the accumulator layout
        auto stage_regs = [&](const f32x4 (&Tn)[GPW][NJ], auto jt_tag, int slot, bool conv) {
            constexpr int jt = decltype(jt_tag)::value;
            float2 P[4];
            float c = 0.0f;
            if (conv) {
#pragma unroll
                for (int r = 0; r < 4; ++r) P[r] = chP[8 * jt + 4 * cq + r];
                c = cexp;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) P[r] = make_float2(rsc_x2, 0.f);
            }
#pragma unroll
            for (int g = 0; g < GPW; ++g) {
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = act1(Tn[g][jt][r], P[r], c);
                put4(y, slot * PLANE + (2 * gyy[g] + dyL + 1) * PW + 16 * gxx[g] + lq + 1, cq);
            }
        };
        auto stage_global = [&](int q, const float (&raw)[UPER][8], bool s16, int slot) {
            const bool conv = q < KO;
            int c0;
            const mi_act& t = oct_tensor(q, c0);
            const float rsc2 = conv ? 0.f : ldexpf(t.scale, sExp[1] + 1), c = conv ? cexp : 0.0f;
            float2 P[8];
            if (conv) {
#pragma unroll
                for (int j = 0; j < 8; ++j) P[j] = chP[8 * q + j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) P[j] = make_float2(rsc2, 0.f);
            }
#pragma unroll
            for (int u = 0; u < UPER; ++u) {
                int iy, x;
                if (!unit_pos(u, iy, x)) continue;
                const int gy = s * R - 1 + iy;
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float xv = raw[u][j];
                    if constexpr (HALF) { if (s16) xv = __uint_as_float(__float_as_uint(xv) << 16); }
                    y[j] = act1(xv, P[j], c);
                }
                uint4 hv, lv = make_uint4(0u, 0u, 0u, 0u);
                if constexpr (HALF) {
                    unsigned h[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const rs_f32x2 v = {y[2 * i], y[2 * i + 1]};
                        h[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, rs_f16x2));
                    }
                    hv = make_uint4(h[0], h[1], h[2], h[3]);
                } else {
                    rs_split8(y, hv, lv);
                }
                if (gy < 0 || gy >= H) { hv = make_uint4(0u, 0u, 0u, 0u); lv = hv; }      // rows outside the image: zero padding follows the activation
                actH[slot * PLANE + iy * PW + x + 1] = hv;
                if constexpr (!HALF) actL[slot * PLANE + iy * PW + x + 1] = lv;
            }
        };
        // the neighbours' edge rows of the resident tensor -> window rows 0 / R + 1 of the octets' planes (octet jt sits in slot jt: pass 0)
        auto stage_halo = [&]() {
            if (wave >= HWV) return;
            const int hnbr = hside ? s + 1 : s - 1;
            const bool hvalid = tid < 4 * TW && hnbr >= 0 && hnbr < S;
            f32x4 v[NJ];
            for (unsigned spins = 0;;) {
                bool ok = true;
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    if (jt < nA && hvalid) {
                        const unsigned off = (unsigned)((((((size_t)par_in * p.B + b) * S + hnbr) * 2 + (1 - hside)) * W + hx) * (MI_RES_MAXC / 2) + 4 * jt + 2 * hcq) * 16u;
                        const f32x4 g0 = mi_buf_load_f32x4_sc1(hbuf, off), g1 = mi_buf_load_f32x4_sc1(hbuf, off + 16u);
                        ok = ok && __float_as_uint(g0[0]) == tag_in && __float_as_uint(g0[2]) == tag_in && __float_as_uint(g1[0]) == tag_in && __float_as_uint(g1[2]) == tag_in;
                        v[jt] = (f32x4){g0[1], g0[3], g1[1], g1[3]};
                    }
                }
                if (__all(ok)) break;
                if (++spins > RS_SPIN_LIMIT) { if (lane == 0) fail(0x200u + (unsigned)li); break; }
                mi_sleep();
            }
            if (!hvalid) return;
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                if (jt >= nA) continue;
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = act1(v[jt][r], chP[8 * jt + 4 * hcq + r], cexp);
                put4(y, jt * PLANE + (hside ? R + 1 : 0) * PW + hx + 1, hcq);
            }
        };
        auto stage = [&](int q, int slot) {
            const int kind = oct_kind(q);
            if (kind == RS_CUR) {                                   // q == slot: the resident octets are the first of the sequence
                if (q == 0) stage_regs(cur, std::integral_constant<int, 0>{}, slot, true);
                if constexpr (NJ > 1) { if (q == 1) stage_regs(cur, std::integral_constant<int, 1>{}, slot, true); }
            } else if (kind == RS_XR) {
                const int r = q - KO;
                if (r == 0) stage_regs(X, std::integral_constant<int, 0>{}, slot, false);
                if constexpr (NJ > 1) { if (r == 1) stage_regs(X, std::integral_constant<int, 1>{}, slot, false); }
            } else {
                if (slot == 0) stage_global(q, raw0, st0, slot);
                else stage_global(q, raw1, st1, slot);
            }
        };
        f32x4 acc[GPW][NJ];
#pragma unroll
        for (int g = 0; g < GPW; ++g)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) acc[g][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // D^T[(dy, co)][px] += W^T[(dy, co)][(r, ci)] . act^T[(r, ci)][px]: the weight fragment is the A operand, the activation chunk the B operand
        auto mma = [&](int q, int slot) {
            const bool isres = q >= KO;
            const int wbase = slot * WSLOT;
            const int ntap = isres ? 1 : 3;
            for (int st = 0; st < ntap; ++st) {
                const int sx = isres ? 1 : st;
                rs_f16x8 bh[NJ], bl[NJ];
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    const int f = wbase + (st * njl + (jt < njl ? jt : 0)) * 128 + lane * 2;
                    bh[jt] = __builtin_bit_cast(rs_f16x8, wl[f]);
                    if constexpr (!HALF) bl[jt] = __builtin_bit_cast(rs_f16x8, wl[f + 1]);
                }
                // all A chunks of the tap first, then the products TERM-major: consecutive matrix-core instructions never share an
                // accumulator (a dependent one waits for its predecessor's passes: GPW x NJ independent ones lie between two on the same tile)
                rs_f16x8 ah[GPW], al[GPW];
#pragma unroll
                for (int g = 0; g < GPW; ++g) {
                    const int idx = slot * PLANE + (2 * gyy[g] + perm) * PW + 16 * gxx[g] + lq + sx;
                    ah[g] = __builtin_bit_cast(rs_f16x8, actH[idx]);
                    if constexpr (!HALF) al[g] = __builtin_bit_cast(rs_f16x8, actL[idx]);
                }
                if constexpr (!HALF) {
#pragma unroll
                    for (int g = 0; g < GPW; ++g)
#pragma unroll
                        for (int jt = 0; jt < NJ; ++jt)
                            if (NJX || jt < njl) acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[jt], al[g], acc[g][jt], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < GPW; ++g)
#pragma unroll
                        for (int jt = 0; jt < NJ; ++jt)
                            if (NJX || jt < njl) acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[jt], ah[g], acc[g][jt], 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < GPW; ++g)
#pragma unroll
                    for (int jt = 0; jt < NJ; ++jt)
                        if (NJX || jt < njl) acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[jt], ah[g], acc[g][jt], 0, 0, 0);
            }
        };
        for (int q0 = 0; q0 < nq; q0 += 2) {
            if (q0 > 0) { __syncthreads(); derive(); weights_dma(L, Ccur, Cx, q0); }   // the planes and weight slots of the previous pass are consumed
            stage(q0, 0);
            if (q0 + 1 < nq) stage(q0 + 1, 1);
            if (q0 == 0) RS_STAMP(4);
            if (q0 == 0 && L.src == 0) stage_halo();
            mi_drain_vmem();                             // this wave's share of the pass's weight fragments has landed in LDS
            __syncthreads();
            if (sAbort) return;
            derive();
            if (q0 == 0) RS_STAMP(5);
            issue_pass(q0 + 2);
            mma(q0, 0);
            if (q0 + 1 < nq) mma(q0 + 1, 1);
        }

        derive();
        RS_STAMP(6);
        RS_WSTAMP(0);
        // ---------------- (5) epilogue: bias, residual; hand the edge rows over first, then statistics, then the copy for other launches
        const float unscale = ldexpf(1.0f, -sExp[2]);
        const bool last = li + 1 == p.n_layers;
        const int res_kind = L.res;
#pragma unroll
        for (int g = 0; g < GPW; ++g)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                f32x4 y = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (NJX || jt < njl) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = fmaf(acc[g][jt][r], unscale, bv[jt][r]);
                }
                cur[g][jt] = y;
            }
        if (res_kind == 1) {
#pragma unroll
            for (int g = 0; g < GPW; ++g)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cur[g][jt][r] += X[g][jt][r];
        } else if (res_kind == 2) {
            const mi_act& t = L.res0;
            const bool r16 = HALF && t.st != 0;
            const float rs = t.scale;
            const size_t eb = (size_t)mi_row_of(b, t.bmod) * t.C * HW;
            float rv[GPW][NJ][4];
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t e0 = eb + (size_t)(8 * (jt < njl ? jt : 0) + r) * HW;                                   // uniform
                    const mi_gptr<const char> pl = reinterpret_cast<mi_gptr<const char>>(r16 ? mi_global(reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(t.data) + e0)) : mi_global(t.data + e0));
#pragma unroll
                    for (int g = 0; g < GPW; ++g) {
                        const unsigned eo = (unsigned)((s * R + 2 * gyy[g] + dyL) * W + 16 * gxx[g] + lq + 4 * cq * HW);      // this lane's element inside plane (8 jt + r)
                        if (HALF && r16) rv[g][jt][r] = mi_bf16_to_f32((unsigned)*reinterpret_cast<mi_gptr<const unsigned short>>(pl + 2u * eo));
                        else rv[g][jt][r] = *reinterpret_cast<mi_gptr<const float>>(pl + 4u * eo);
                    }
                }
#pragma unroll
            for (int g = 0; g < GPW; ++g)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt)
                    if (NJX || jt < njl) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) cur[g][jt][r] = fmaf(rv[g][jt][r], rs, cur[g][jt][r]);
                    }
        }
        float cs[NJ][4], cqq[NJ][4];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = 0.f, q2 = 0.f;
#pragma unroll
                for (int g = 0; g < GPW; ++g) { const float y = cur[g][jt][r]; a += y; q2 = fmaf(y, y, q2); }
                cs[jt][r] = a;
                cqq[jt][r] = q2;
            }
        RS_STAMP(8);
        RS_WSTAMP(1);
        if (L.save_x) {
#pragma unroll
            for (int g = 0; g < GPW; ++g)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) X[g][jt] = cur[g][jt];
            Cx = Cout;
        }
        prev_saved = L.save_x != 0;
        Ccur = Cout;
        // per-channel (sum, sum of squares) of the slab: the 16 lanes of a row hold 8 NJ values each -> a transposing butterfly (every
        // step hands half of the remaining values to the partner lane: 8 NJ - 1 exchanges instead of 4 per value), then the other row
        {
            constexpr int NV = 8 * NJ;
            float v[NV];
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[(jt * 4 + r) * 2] = cs[jt][r]; v[(jt * 4 + r) * 2 + 1] = cqq[jt][r]; }
#define RS_FOLD(HH, MM) { const bool up = (lq & MM) != 0; _Pragma("unroll") for (int i = 0; i < HH; ++i) { \
                const float send = up ? v[i] : v[i + HH], keep = up ? v[i + HH] : v[i]; v[i] = keep + __shfl_xor(send, MM); } }
            if constexpr (NV == 16) { RS_FOLD(8, 8) RS_FOLD(4, 4) RS_FOLD(2, 2) RS_FOLD(1, 1) }
            else { RS_FOLD(4, 8) RS_FOLD(2, 4) RS_FOLD(1, 2) v[0] += __shfl_xor(v[0], 1); }
#undef RS_FOLD
            v[0] += __shfl_xor(v[0], 32);
            // this lane now holds value index (lq >> (NJ == 1 ? 1 : 0)) = (jt * 4 + r) * 2 + k of its channel quad
            const int vi = NJ == 1 ? (lq >> 1) : lq;
            if (lg < 2) red[wave][(8 * (vi >> 3) + 4 * lg + ((vi >> 1) & 3)) * 2 + (vi & 1)] = v[0];
        }
        RS_WSTAMP(2);
        __syncthreads();
        derive();
        RS_STAMP(9);
        RS_WSTAMP(3);
        if (!last) weights_dma(p.layer[li + 1], Ccur, Cx, 0);
        if (tid < Cout) {
            float ts = 0.f, tq = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { ts += red[w][2 * tid]; tq += red[w][2 * tid + 1]; }       // fixed order
            if (!last) {
                const float tg = __uint_as_float(tag_out);
                mi_buf_store_f32x4_sc1(sbuf, (unsigned)((((size_t)par_out * p.B + b) * S + s) * MI_RES_MAXC + tid) * 16u, (f32x4){tg, ts, tg, tq});
            }
            if (L.out_stats) {
                float* os = L.out_stats + ((size_t)(b * Cout + tid) * L.out_nt + s) * 2;
                os[0] = ts;
                os[1] = tq;
            }
        }
        RS_STAMP(10);
        if (!last) {
#pragma unroll
            for (int g = 0; g < GPW; ++g) {
                const bool top = gyy[g] == 0 && dyL == 0, bot = gyy[g] == GY - 1 && dyL == 1;
                if (top || bot) {
#pragma unroll
                    for (int jt = 0; jt < NJ; ++jt) {
                        if (!NJX && jt >= njl) continue;
                        const unsigned off = (unsigned)((((((size_t)par_out * p.B + b) * S + s) * 2 + (bot ? 1 : 0)) * W + 16 * gxx[g] + lq) * (MI_RES_MAXC / 2) + 4 * jt + 2 * cq) * 16u;
                        const float tg = __uint_as_float(tag_out);
                        mi_buf_store_f32x4_sc1(hbuf, off, (f32x4){tg, cur[g][jt][0], tg, cur[g][jt][1]});
                        mi_buf_store_f32x4_sc1(hbuf, off + 16u, (f32x4){tg, cur[g][jt][2], tg, cur[g][jt][3]});
                    }
                }
            }
        }

        if (L.out) {
#pragma unroll
            for (int g = 0; g < GPW; ++g)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    if (!NJX && jt >= njl) continue;
                    const unsigned eo = (unsigned)((s * R + 2 * gyy[g] + dyL) * W + 16 * gxx[g] + lq + 4 * cq * HW);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const size_t e0 = ((size_t)b * Cout + 8 * jt + r) * HW;                                              // uniform
                        if (HALF && L.out_st) *reinterpret_cast<mi_gptr<unsigned short>>(reinterpret_cast<mi_gptr<char>>(mi_global(reinterpret_cast<unsigned short*>(L.out) + e0)) + 2u * eo) = (unsigned short)mi_f32_to_bf16x2(cur[g][jt][r], 0.0f);
                        else *reinterpret_cast<mi_gptr<float>>(reinterpret_cast<mi_gptr<char>>(mi_global(L.out + e0)) + 4u * eo) = cur[g][jt][r];
                    }
                }
        }
        RS_STAMP(7);
    }
}

template <int TW, int R, int NW>
int launch_resident(const mi_resident_params& p, hipStream_t st, int nj, bool exact) {
    const int S = p.H / R;
    const dim3 grid((unsigned)(p.B * S)), block(64 * NW);
#define RS_LAUNCH(NJ_, X_, H_) hipLaunchKernelGGL(HIP_KERNEL_NAME(resident_convs_kernel<TW, R, NW, NJ_, X_, H_>), grid, block, 0, st, p)
    if (p.half) {
        if (nj == 2) { if (exact) RS_LAUNCH(2, true, true); else RS_LAUNCH(2, false, true); }
        else RS_LAUNCH(1, true, true);
    } else {
        if (nj == 2) { if (exact) RS_LAUNCH(2, true, false); else RS_LAUNCH(2, false, false); }
        else RS_LAUNCH(1, true, false);
    }
#undef RS_LAUNCH
    return mi_check_launch("resident_convs_kernel");
}

// rows per slab.  64-wide images: 16 (one workgroup per CU at 64 x 64 x 64 images).  MINIMAGEN_RES_ROWS64=8 selects 8-row slabs (two
// workgroups per CU; measured slower: the per-layer hand-over latencies do not shrink with the slab, profiles/r04_resident_phase_trace.txt)
int rs_rows(int H, int W) {
    static const int rows64 = [] { const char* e = getenv("MINIMAGEN_RES_ROWS64"); return (e && atoi(e) == 8) ? 8 : 16; }();
    if (W == 64 && H % rows64 == 0) return rows64;
    if (W == 32 && H % 8 == 0) return 8;
    return 0;
}

}  // namespace

extern "C" int mi_resident_slabs(int H, int W) {
    const int R = rs_rows(H, W);
    return (R && H / R <= RS_MAXS) ? H / R : 0;
}

extern "C" long long mi_resident_sync_bytes(int B, int H, int W) {
    const int S = mi_resident_slabs(H, W);
    if (S <= 0 || B <= 0) return 0;
    return rs_sync_layout(B, S, W).total;
}

extern "C" int mi_resident_error_offset(void) { return 8; }

extern "C" int mi_resident_convs_fwd(const mi_resident_params* pp, void* stream) {
    const mi_resident_params& p = *pp;
    const int R = rs_rows(p.H, p.W);
    if (!R || p.B <= 0) { mi_set_error("mi_resident_convs_fwd: %dx%d images are not supported (64 wide with H %% 16 == 0, or 32 wide with H %% 8 == 0)", p.H, p.W); return MI_ERR_UNSUPPORTED; }
    if (p.H / R > RS_MAXS) { mi_set_error("mi_resident_convs_fwd: more than %d slabs per image", RS_MAXS); return MI_ERR_UNSUPPORTED; }
    if (p.n_layers < 1 || p.n_layers > MI_RES_MAX_LAYERS || !p.sync) { mi_set_error("mi_resident_convs_fwd: 1..%d layers and a sync buffer are required", MI_RES_MAX_LAYERS); return MI_ERR_INVALID; }
    bool two = false, one = false;
    int Ccur = 0, Cx = 0;
    for (int i = 0; i < p.n_layers; ++i) {
        const mi_res_layer& L = p.layer[i];
        if (L.Cout != 8 && L.Cout != 16) { mi_set_error("mi_resident_convs_fwd: layer %d: Cout must be 8 or 16", i); return MI_ERR_UNSUPPORTED; }
        if (!L.w_rp) { mi_set_error("mi_resident_convs_fwd: layer %d: weight fragments missing", i); return MI_ERR_INVALID; }
        if (i == 0 && L.src == 0) { mi_set_error("mi_resident_convs_fwd: the first layer reads global memory"); return MI_ERR_INVALID; }
        if (L.src != 0 && (!L.in0.data || !L.in0.stats)) { mi_set_error("mi_resident_convs_fwd: layer %d: in0 needs data and statistics", i); return MI_ERR_INVALID; }
        if (L.in1.data && !L.in1.stats) { mi_set_error("mi_resident_convs_fwd: layer %d: in1 needs statistics", i); return MI_ERR_INVALID; }
        const int ca = L.src == 0 ? Ccur : L.in0.C, cb = L.in1.data ? L.in1.C : 0;
        if ((ca & 7) || (cb & 7) || ca + cb > 32 || ca > 16 * (L.src == 0 ? 1 : 2)) { mi_set_error("mi_resident_convs_fwd: layer %d: input channels %d + %d unsupported", i, ca, cb); return MI_ERR_UNSUPPORTED; }
        if (L.gn_groups > 0 && (L.gn_groups > MI_MAX_GROUPS || (ca + cb) % L.gn_groups || !L.gn_gamma || !L.gn_beta)) { mi_set_error("mi_resident_convs_fwd: layer %d: GroupNorm parameters", i); return MI_ERR_INVALID; }
        if (L.ss_off >= 0 && !p.scale_shift) { mi_set_error("mi_resident_convs_fwd: layer %d: scale_shift table missing", i); return MI_ERR_INVALID; }
        if (L.res < 0 || L.res > 4) { mi_set_error("mi_resident_convs_fwd: layer %d: residual kind", i); return MI_ERR_INVALID; }
        if ((L.res == 1 || L.res == 3) && Cx == 0) { mi_set_error("mi_resident_convs_fwd: layer %d: residual from X, but no layer saved it", i); return MI_ERR_INVALID; }
        if (L.res == 1 && Cx != L.Cout) { mi_set_error("mi_resident_convs_fwd: layer %d: identity residual with %d channels into %d", i, Cx, L.Cout); return MI_ERR_INVALID; }
        if (L.res == 2 && (!L.res0.data || L.res0.C != L.Cout)) { mi_set_error("mi_resident_convs_fwd: layer %d: identity residual tensor", i); return MI_ERR_INVALID; }
        if (L.res >= 3) {
            const int ra = L.res == 3 ? Cx : L.res0.C, rb = L.res1.data ? L.res1.C : 0;
            if (!L.res_w_rp || (ra & 7) || (rb & 7) || ra + rb > 32) { mi_set_error("mi_resident_convs_fwd: layer %d: 1x1 residual conv over %d + %d channels", i, ra, rb); return MI_ERR_UNSUPPORTED; }
            if (L.res == 4 && (!L.res0.data || !L.res0.stats)) { mi_set_error("mi_resident_convs_fwd: layer %d: res0 needs data and statistics", i); return MI_ERR_INVALID; }
            if (L.res1.data && !L.res1.stats) { mi_set_error("mi_resident_convs_fwd: layer %d: res1 needs statistics", i); return MI_ERR_INVALID; }
        }
        if (L.out_stats && L.out_nt < p.H / R) { mi_set_error("mi_resident_convs_fwd: layer %d: out_stats has %d slots per channel, %d slabs", i, L.out_nt, p.H / R); return MI_ERR_INVALID; }
        if (L.Cout > 8) two = true; else one = true;
        Ccur = L.Cout;
        if (L.save_x) Cx = L.Cout;
    }
    if (rs_sync_layout(p.B, p.H / R, p.W).total >= (1ll << 31)) { mi_set_error("mi_resident_convs_fwd: batch too large for 32-bit exchange offsets"); return MI_ERR_UNSUPPORTED; }
    const int nj = two ? 2 : 1;
    const bool exact = !(two && one);
    if (p.W == 64) return R == 16 ? launch_resident<64, 16, 8>(p, (hipStream_t)stream, nj, exact) : launch_resident<64, 8, 8>(p, (hipStream_t)stream, nj, exact);
    return launch_resident<32, 8, 8>(p, (hipStream_t)stream, nj, exact);
}
