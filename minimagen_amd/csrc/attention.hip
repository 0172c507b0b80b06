// K9: the bottleneck cross-attention of MinImagen's ResnetBlock (layers.py:220-251, 433-435) in folded form (see minimagen_hip.h).
// One wave owns 16 image tokens; for every head it computes S^T = G_h . x^^T (swapped operands, so a token's scores over all context rows j
// sit in ONE lane's accumulators + the 3 lanes 16/32/48 away -> row max / row sum are in-register plus two cross-lane steps),
// exponentiates in place, and feeds the accumulators straight back as the B operand of O^T += VW_h^T . P^T: the C/D layout of the score
// tiles IS the B layout of PV, so P never moves.  The whole (tokens x 261) score row lives in registers: no online-softmax rescaling.
// (Rounds 1-5 also carried exact-fp32 MFMA forms of this kernel -- `variant` 0 .. 5 of mi_cross_attn_fwd -- as A/B yardsticks; removed in
//  round 6: they were reachable through an environment knob only.  git history has them.)
#include "common.hip.h"

namespace {

// ---- variant 6: the same algorithm with both contractions as 3-term fp16 splits on the real matrix cores.
// Measured on MI355X (profiles/r01_mfma_valu_overlap_ubench.txt): v_mfma_f32_16x16x4_f32 does not overlap with VALU work
// (it runs at the VALU fp32 rate and serialises with it), v_mfma_f32_16x16x16_f16 takes ~0.6x its time for 4x the K and
// hides 2-3 VALU instructions.  x = hi + lo with hi = fp16(x), lo = fp16(x - hi) keeps 22 mantissa bits; the dropped
// lo*lo term is 2^-22 relative, so the result stays inside the fp32 parity tolerance.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// ATTN_QK32=1: QK^T as {G hi | G lo}.{x hi | x hi} (K = 32) + G hi . x lo (K = 16): two instructions per tile instead of three.  Correct on the
// SIMT emulator, WRONG on the MI355X as compiled -- root cause (round 3, profiles/r03_attention_qk32_root_cause.txt): hipcc emits
//     v_mfma_f32_16x16x16_f16 v[14:17], ...            (4 passes)
//     v_mfma_f32_16x16x32_f16 v[70:73], ..., v[14:17]  (8 passes; SrcC = the previous result, a DIFFERENT vDst)
// back to back with no wait state, and the hardware does not interlock a dependent SrcC read across the two instruction shapes: the second
// instruction reads the accumulator before it is written.  Either order fails; separate accumulators added on the VALU pass all 16 device cases, and
// so does the same-accumulator form with s_nop states between the two instructions (inline asm).  With the wait states the form is no faster than
// three K = 16 instructions (same 24 pipe cycles), so it stays off: this kernel chains only instructions of ONE shape per accumulator.
#ifndef ATTN_QPF
#define ATTN_QPF 1          // prefetch the next QK^T fragment (one tile ahead) in the 3-term kernel
#endif
#ifndef ATTN_QK32
#define ATTN_QK32 2         // 2: QK^T as two K = 32 instructions per tile (one shape, chained); 0: three K = 16 instructions; 1: the mixed-shape form (broken, see above)
#endif

__device__ __forceinline__ void split_f16(const float (&x)[4], f16x4& hi, f16x4& lo) { mi_split_f16(x, hi, lo); }

// HALF = true: the reduced-precision configuration (variant 7): single fp16 term per product (no lo halves), fp32 accumulate,
// fp32 softmax -- the BASELINE 'half-precision with MFMA attention' configurations; parity gate 3e-2 / 3e-3 instead of fp32's 1e-4 / 1e-5
// NWV waves of 16 tokens per workgroup: every workgroup copies the whole head's fragments, so larger workgroups divide that
// L2 -> LDS traffic (1.1 GB per SR launch at 4 waves); statistics stay per 64-token tile (4 waves)
template <int C, int JT, int WPS, bool HALF, int NWV>
__global__ __launch_bounds__(64 * NWV, WPS) void cross_attn_f16x3_kernel(const mi_cross_attn_params p) {
    constexpr int KC = (C + 15) / 16, MT = (C + 15) / 16, FRH = 8 * KC + 8 * MT, TOK_WG = 16 * NWV;
    constexpr int JP = (JT + 1) / 2;
    __shared__ double red[NWV][2 * 16 * MT];
    // One head's context fragments as they lie in global memory, [tile][chunk q][lane] 16 bytes: q < KC: G {4 hi | 4 lo}; q >= KC: the V
    // chunks, arranged per PAIR of tiles for the K = 32 PV instruction (even tile: {4 hi(t0) | 4 hi(t1)}, odd tile: {4 lo(t0) | 4 lo(t1)},
    // written that way by attn_fold_rows_kernel; tile count padded to even, the pad stays zero).  DOUBLE-buffered and filled by LDS-DMA
    // (global_load_lds_dwordx4: no registers, asynchronous): head h + 1 streams in under head h's MFMA / softmax work, one barrier per
    // head.  (Round 1 staged them with a synchronous barrier - copy - barrier per head: measured 19 % of the kernel.)
    constexpr int QC = KC + MT, JTS = (JT + 1) & ~1, ROWS = JTS * QC;
    __shared__ __attribute__((aligned(16))) uint4 frag[2][ROWS * 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int tiles = (p.HW + TOK_WG - 1) / TOK_WG;
    int b, tile;
    if ((p.B2 & 7) == 0) {
        const int L = blockIdx.x, k = L >> 3;
        b = (L & 7) + 8 * (k / tiles);
        tile = k % tiles;
    } else {
        b = blockIdx.x / tiles;
        tile = blockIdx.x % tiles;
    }
    const int bx = mi_row_of(b, p.x.bmod);
    const int i = (tile * NWV + wave) * 16 + lq;
    const bool ok = i < p.HW;
    const float* xb = p.x.data + (size_t)bx * C * p.HW;
    // single-term variant: x and out may be stored as bf16 (mi_act.st / out_st)
    const bool x16 = HALF && p.x.st != 0, o16 = HALF && p.out_st != 0;
    auto ldx = [&](int a) -> float {
        if (x16) return mi_bf16_to_f32(reinterpret_cast<const unsigned short*>(p.x.data)[((size_t)bx * C + a) * p.HW + i]);
        return xb[(size_t)a * p.HW + i];
    };

    // LayerNorm(x) -> B operand of QK^T: this lane supplies channels a = 16kc + 4lg + e of token lq
    f16x4 xhi[KC], xlo[KC];
    {
        float xf[KC][4];
        float s = 0.0f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = 16 * kc + 4 * lg + e;
                xf[kc][e] = (ok && a < C) ? ldx(a) * p.x.scale : 0.0f;
                s += xf[kc][e];
            }
        s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
        const float mean = s / (float)C;
        float v = 0.0f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = (16 * kc + 4 * lg + e < C) ? xf[kc][e] - mean : 0.0f; v = fmaf(d, d, v); }
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        const float rstd = 1.0f / sqrtf(v / (float)C + 1e-5f);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            float xn[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = 16 * kc + 4 * lg + e;
                xn[e] = (a < C) ? ldexpf((xf[kc][e] - mean) * rstd * p.n1_g[a] + p.n1_b[a], p.x_exp) : 0.0f;
            }
            split_f16(xn, xhi[kc], xlo[kc]);
        }
    }
    f32x4 oacc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) oacc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const _Float16* gvb = reinterpret_cast<const _Float16*>(p.gv) + (size_t)b * p.heads * JTS * 64 * FRH;
    const int jlast = p.J - 1;
    auto issue_head = [&](int h, int buf) {
        const _Float16* gvh = gvb + (size_t)h * JTS * 64 * FRH;
        for (int r = wave; r < ROWS; r += NWV) {             // one 1 KB row (64 lanes x 16 bytes) per instruction
            const int jt = r / QC, q = r % QC;
            const _Float16* src = gvh + ((size_t)(jt * 64 + lane) * QC + q) * 8;
#if defined(HIPEMU)
            frag[buf][r * 64 + lane] = *reinterpret_cast<const uint4*>(src);
#else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)&frag[buf][r * 64], 16, 0, 0);
#endif
        }
    };
    issue_head(0, 0);

    for (int h = 0; h < p.heads; ++h) {
        const int hb = h & 1;
#if !defined(HIPEMU)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of head h has landed in LDS ...
#endif
        __syncthreads();                                  // ... everybody's has, and nobody reads head h - 1's buffer any more
        if (h + 1 < p.heads) issue_head(h + 1, hb ^ 1);
        f32x4 s[JT];
        uint4 gq[2];                                       // 3-term, one channel chunk: the next tile's fragment is read while this tile multiplies
        constexpr bool QPF = !HALF && KC == 1 && ATTN_QPF;
        if constexpr (QPF) gq[0] = frag[hb][lane];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (QPF) {
                if (jt + 1 < JT) gq[(jt + 1) & 1] = frag[hb][((jt + 1) * QC) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                union { uint4 u; f16x4 h2[2]; uint2 u2; } g;
                if constexpr (HALF) g.u2 = reinterpret_cast<const uint2*>(&frag[hb][0])[2 * ((jt * QC + kc) * 64 + lane)];    // hi halves only
                else if constexpr (QPF) g.u = gq[jt & 1];
                else g.u = frag[hb][(jt * QC + kc) * 64 + lane];
                const f16x4 ghi = g.h2[0];
                if constexpr (!HALF) {
#if ATTN_QK32 == 2
                    // both instructions of ONE shape (K = 32), so the accumulator may be chained: {G hi | G lo} . {x lo | 0} + {G hi | G lo} . {x hi | x hi}.
                    // The A operand is the LDS chunk as it lies; a K = 16 instruction occupies the matrix pipe as long as a K = 32 one
                    // (profiles/r01_mfma_f16_chain_ubench.txt), so this is two pipe slots per tile instead of three.
                    const f16x4 z4 = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
                    const f16x8 xhh = __builtin_shufflevector(xhi[kc], xhi[kc], 0, 1, 2, 3, 4, 5, 6, 7);
                    const f16x8 xl0 = __builtin_shufflevector(xlo[kc], z4, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, g.u), xl0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, g.u), xhh, acc, 0, 0, 0);
#elif ATTN_QK32
                    // {G hi | G lo} . {x hi | x hi} on the K = 32 instruction (the LDS chunk IS that A operand) + G hi . x lo on the K = 16 one:
                    // two instructions per tile instead of three
                    const f16x8 xhh = __builtin_shufflevector(xhi[kc], xhi[kc], 0, 1, 2, 3, 4, 5, 6, 7);
                    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ghi, xlo[kc], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, g.u), xhh, acc, 0, 0, 0);
#else
                    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(g.h2[1], xhi[kc], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ghi, xlo[kc], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ghi, xhi[kc], acc, 0, 0, 0);
#endif
                } else {
                    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ghi, xhi[kc], acc, 0, 0, 0);
                }
            }
            s[jt] = acc;
            // single-term variant: one MFMA per tile leaves the scheduler free to hoist every tile's LDS read (spills at 128 registers)
            if constexpr (HALF) { if ((jt & 3) == 3) __builtin_amdgcn_sched_barrier(0); }
        }
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (jt == JT - 1 && (16 * jt + 4 * lg + r) > jlast) s[jt][r] = -INFINITY;
                m = fmaxf(m, s[jt][r]);
            }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const float sc = ldexpf(1.0f, -(p.x_exp + p.g_exp)), msc = -m * sc;      // undo the operand scalings inside the subtraction's FMA
        mi_f32x2 l2 = {0.0f, 0.0f};
        const mi_f32x2 sc2 = {sc, sc}, msc2 = {msc, msc};
        f32x4 oh[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) oh[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // PV on the K = 32 instruction, context tiles in pairs (t0, t1): {V hi(t0)|V hi(t1)} . {P hi(t0)|P hi(t1)} + {V hi}.{P lo} + {V lo}.{P hi}.
        // The B operand is the C/D layout of the two score tiles as it is: P[j = 16t + 4lg + e][token lq].
#pragma unroll
        for (int jp = 0; jp < JP; ++jp) {
            f16x4 ph[2], pl[2];
#pragma unroll
            for (int ts = 0; ts < 2; ++ts) {
                const int jt = 2 * jp + ts;
                float pe[4] = {0.f, 0.f, 0.f, 0.f};
                if (jt < JT) {
                    // exponent arguments and the running sum on packed pairs (v_pk_fma_f32 / v_pk_add_f32)
                    const f32x4 sv = s[jt < JT ? jt : 0];
                    const mi_f32x2 a01 = mi_pk_fma((mi_f32x2){sv[0], sv[1]}, sc2, msc2), a23 = mi_pk_fma((mi_f32x2){sv[2], sv[3]}, sc2, msc2);
                    pe[0] = __builtin_amdgcn_exp2f(a01[0]); pe[1] = __builtin_amdgcn_exp2f(a01[1]);
                    pe[2] = __builtin_amdgcn_exp2f(a23[0]); pe[3] = __builtin_amdgcn_exp2f(a23[1]);
                    l2 = mi_pk_add(l2, (mi_f32x2){pe[0], pe[1]});
                    l2 = mi_pk_add(l2, (mi_f32x2){pe[2], pe[3]});
                }
                if constexpr (HALF) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) ph[ts][r] = (_Float16)pe[r];
                } else {
                    split_f16(pe, ph[ts], pl[ts]);
                }
            }
            const f16x8 phi = __builtin_shufflevector(ph[0], ph[1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f16x8 vhi = __builtin_bit_cast(f16x8, frag[hb][((2 * jp) * QC + KC + mt) * 64 + lane]);         // {V hi(t0) | V hi(t1)} as stored
                if constexpr (!HALF) {
                    const f16x8 plo = __builtin_shufflevector(pl[0], pl[1], 0, 1, 2, 3, 4, 5, 6, 7);
                    const f16x8 vlo = __builtin_bit_cast(f16x8, frag[hb][((2 * jp + 1) * QC + KC + mt) * 64 + lane]); // {V lo(t0) | V lo(t1)}
                    oh[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vlo, phi, oh[mt], 0, 0, 0);
                    oh[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhi, plo, oh[mt], 0, 0, 0);
                }
                oh[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhi, phi, oh[mt], 0, 0, 0);
            }
        }
        float l = l2[0] + l2[1];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float linv = ldexpf(1.0f / l, -p.v_exp);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[mt][r] = fmaf(oh[mt][r], linv, oacc[mt][r]);
    }

    // to_out.1 LayerNorm + residual + statistics (same as the fp32 kernel)
    float yv[4 * MT];
#pragma unroll
    for (int e = 0; e < 4 * MT; ++e) yv[e] = 0.0f;
    {
        float s1 = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s1 += oacc[mt][r];
        s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
        const float mean = s1 / (float)C;
        float v = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int a = 16 * mt + 4 * lg + r; const float d = (a < C) ? oacc[mt][r] - mean : 0.0f; v = fmaf(d, d, v); }
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        const float rstd = 1.0f / sqrtf(v / (float)C + 1e-5f);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = 16 * mt + 4 * lg + r;
                if (a < C && ok) {
                    const float y = (oacc[mt][r] - mean) * rstd * p.n2_g[a] + p.n2_b[a] + ldx(a) * p.x.scale;
                    if (o16) reinterpret_cast<unsigned short*>(p.out)[((size_t)b * C + a) * p.HW + i] = (unsigned short)(mi_f32_to_bf16x2(y, 0.0f) & 0xffffu);
                    else p.out[((size_t)b * C + a) * p.HW + i] = y;
                    yv[4 * mt + r] = y;
                }
            }
    }
    if (p.out_stats) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = 16 * mt + 4 * lg + r;
                double S, Q;
                mi_stat_reduce16(yv[4 * mt + r], a < C && ok, lane, S, Q);
                if (lq == 0) { red[wave][2 * a] = S; red[wave][2 * a + 1] = Q; }
            }
        __syncthreads();
        const int nt64 = (p.HW + 63) / 64;               // statistics tiles are 64 tokens = 4 waves, whatever the workgroup size
        for (int sub = 0; sub < NWV / 4; ++sub) {
            const int st = tile * (NWV / 4) + sub;
            if (tid < 2 * C && st < nt64)
                p.out_stats[((size_t)(b * C + (tid >> 1)) * nt64 + st) * 2 + (tid & 1)] =
                    red[4 * sub][tid] + red[4 * sub + 1][tid] + red[4 * sub + 2][tid] + red[4 * sub + 3][tid];
        }
    }
}

}  // namespace

extern "C" int mi_cross_attn_fwd(const mi_cross_attn_params* pp, void* stream) {
    const mi_cross_attn_params& p = *pp;
    hipStream_t st = (hipStream_t)stream;
    const int JT = (p.J + 15) / 16;
    // MinImagen's contexts: null + time tokens + 256 text rows (17 tiles), or null + time tokens only when the U-Net is called
    // without text (Unet.py:572: text is optional; 1 tile, fp16 kernel only)
    if (JT != 17 && JT != 1) { mi_set_error("mi_cross_attn_fwd: context of %d rows (%d tiles) not instantiated (MinImagen: 1 + time tokens [+ 256])", p.J, JT); return MI_ERR_UNSUPPORTED; }
    if (p.B2 <= 0 || p.HW <= 0) { mi_set_error("mi_cross_attn_fwd: empty problem"); return MI_ERR_INVALID; }
    if ((p.x.st || p.out_st) && p.variant != 7) { mi_set_error("mi_cross_attn_fwd: bf16 activation storage is variant 7 only"); return MI_ERR_UNSUPPORTED; }
    if (p.variant == 6 || p.variant == 7) {      // fp16 MFMA (fragments from mi_attn_fold_rows with frag_f16 = 1): 6 = 3-term split (fp32-grade), 7 = single term
        // waves per workgroup (measured on MI355X): 8 for the SR bottleneck (4096 tokens: 0.53 -> 0.47 ms per pair of launches),
        // 16 for up to 1024 tokens (base U-Net)
        static const int force_nwv = getenv("MI_ATTN_WAVES") ? atoi(getenv("MI_ATTN_WAVES")) : 0;      // A/B knob: 8 or 16
        const int nwv = (force_nwv == 8 || force_nwv == 16) ? force_nwv : (p.HW <= 1024 ? 16 : 8);
        const dim3 g6(((p.HW + 16 * nwv - 1) / (16 * nwv)) * p.B2);
#define MI_ATTN16_LAUNCH(CC) \
        if (JT == 1) { \
            if (p.variant == 6) hipLaunchKernelGGL(HIP_KERNEL_NAME(cross_attn_f16x3_kernel<CC, 1, (CC <= 16 ? 4 : 2), false, 8>), dim3(((p.HW + 127) / 128) * p.B2), dim3(512), 0, st, p); \
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(cross_attn_f16x3_kernel<CC, 1, (CC <= 16 ? 4 : 2), true, 8>), dim3(((p.HW + 127) / 128) * p.B2), dim3(512), 0, st, p); \
        } else \
        if (p.variant == 6 && nwv == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(cross_attn_f16x3_kernel<CC, 17, (CC <= 16 ? 4 : 2), false, 8>), g6, dim3(512), 0, st, p); \
        else if (p.variant == 6) hipLaunchKernelGGL(HIP_KERNEL_NAME(cross_attn_f16x3_kernel<CC, 17, (CC <= 16 ? 4 : 2), false, 16>), g6, dim3(1024), 0, st, p); \
        else if (nwv == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(cross_attn_f16x3_kernel<CC, 17, (CC <= 16 ? 4 : 2), true, 8>), g6, dim3(512), 0, st, p); \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(cross_attn_f16x3_kernel<CC, 17, (CC <= 16 ? 4 : 2), true, 16>), g6, dim3(1024), 0, st, p);
        switch (p.C) {
            case 8: MI_ATTN16_LAUNCH(8) break;
            case 16: MI_ATTN16_LAUNCH(16) break;
            case 32: MI_ATTN16_LAUNCH(32) break;
            default: mi_set_error("mi_cross_attn_fwd: folded path instantiated for C in {8,16,32}, got %d", p.C); return MI_ERR_UNSUPPORTED;
        }
#undef MI_ATTN16_LAUNCH
        return mi_check_launch("cross_attn_f16x3_kernel");
    }
    mi_set_error("mi_cross_attn_fwd: variant %d (6 = 3-term fp16 split, fp32-grade; 7 = single fp16 term: the reduced-precision configuration)", p.variant);
    return MI_ERR_UNSUPPORTED;
}

// =====================================================================================================
// K10: folded multi-query self-attention (layers.py:52-104) + ChanFeedForward (layers.py:148-161) for narrow layers
namespace {

// LayerNorm over the channels of every token, NCHW in -> token-major [B][HW][C] out.  One workgroup = 64 tokens: work-item (token, channel
// quarter) reads the planes coalesced along the tokens (mean, then the centred second moment, like the reference's torch.var), and the
// normalised values go through a 64 x 64 LDS tile so that the token-major rows are written 64 bytes per work-item instead of one float
// per work-item and row (round 2's form: 233 us per launch at 64 x 64 x 128 channels).
__global__ __launch_bounds__(256) void ln_tokens_kernel(const mi_act x, int HW, const float* gamma, const float* beta, float* out) {
    __shared__ float part[4][64];
    __shared__ float sMean[64], sRstd[64];
    __shared__ float T[64 * 65];
    const int tid = threadIdx.x, tl = tid & 63, cq = tid >> 6;
    const int b = blockIdx.y, C = x.C;
    const int bx = mi_row_of(b, x.bmod);
    const int tok0 = blockIdx.x * 64, tok = tok0 + tl;
    const bool ok = tok < HW;
    const float* xb = x.data + (size_t)bx * C * HW + (ok ? tok : HW - 1);
    float s = 0.0f;
    for (int c = cq; c < C; c += 4) s += xb[(size_t)c * HW] * x.scale;
    part[cq][tl] = s;
    __syncthreads();
    if (cq == 0) sMean[tl] = ((part[0][tl] + part[1][tl]) + (part[2][tl] + part[3][tl])) / (float)C;
    __syncthreads();
    const float mean = sMean[tl];
    float v = 0.0f;
    for (int c = cq; c < C; c += 4) { const float d = xb[(size_t)c * HW] * x.scale - mean; v = fmaf(d, d, v); }
    part[cq][tl] = v;
    __syncthreads();
    if (cq == 0) sRstd[tl] = 1.0f / sqrtf(((part[0][tl] + part[1][tl]) + (part[2][tl] + part[3][tl])) / (float)C + 1e-5f);
    __syncthreads();
    const float rstd = sRstd[tl];
    const int orow = tid >> 2, oseg = (tid & 3) * 16;               // output role: token row, 16-channel segment of the 64-channel block
    // blockIdx.z: which share of the 64-channel blocks (few tokens per image -- 16 x 16 -- would otherwise leave half of the CUs without a workgroup);
    // every share computes the token moments for itself
    const int nblk = (C + 63) / 64, per = (nblk + (int)gridDim.z - 1) / (int)gridDim.z;
    const int c_lo = (int)blockIdx.z * per * 64, c_hi = c_lo + per * 64 < C ? c_lo + per * 64 : C;
    for (int c0 = c_lo; c0 < c_hi; c0 += 64) {
        for (int cc = cq; cc < 64; cc += 4) {
            const int c = c0 + cc;
            if (c < C) T[tl * 65 + cc] = (xb[(size_t)c * HW] * x.scale - mean) * rstd * gamma[c] + beta[c];
        }
        __syncthreads();
        if (tok0 + orow < HW) {
            float* o = out + ((size_t)b * HW + tok0 + orow) * C + c0 + oseg;
            if (c0 + oseg + 16 <= C && (C & 3) == 0) {
#pragma unroll
                for (int i = 0; i < 16; i += 4)
                    *reinterpret_cast<float4*>(o + i) = make_float4(T[orow * 65 + oseg + i], T[orow * 65 + oseg + i + 1], T[orow * 65 + oseg + i + 2], T[orow * 65 + oseg + i + 3]);
            } else {
                for (int i = 0; i < 16; ++i) if (c0 + oseg + i < C) o[i] = T[orow * 65 + oseg + i];
            }
        }
        __syncthreads();
    }
}

// Same operand scheme as cross_attn_folded_kernel (16 tokens per wave), but the context (HW+1 rows) is walked in
// chunks of JTC tiles with an online softmax: running max m, running sum l, O_h rescaled by 2^(m_old - m_new).
template <int C, int JTC>
__global__ __launch_bounds__(256) void self_attn_folded_kernel(const mi_self_attn_params p, const int JT) {
    constexpr int KK = C / 4, NGP = KK < 4 ? 4 : KK, MT = (C + 15) / 16, FR = NGP + 4 * MT;
    __shared__ double red[4][2 * 16 * MT];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int tiles = (p.HW + 63) / 64;
    const int b = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int bx = mi_row_of(b, p.x.bmod);
    const int i = (tile * 4 + wave) * 16 + lq;
    const bool ok = i < p.HW;
    const float* xb = p.x.data + (size_t)bx * C * p.HW;
    float xh[KK];
    {
        float s = 0.0f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) { xh[kk] = ok ? xb[(size_t)(4 * kk + lg) * p.HW + i] * p.x.scale : 0.0f; s += xh[kk]; }
        s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
        const float mean = s / (float)C;
        float v = 0.0f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) { const float d = xh[kk] - mean; v = fmaf(d, d, v); }
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        const float rstd = 1.0f / sqrtf(v / (float)C + 1e-5f);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) { const int a = 4 * kk + lg; xh[kk] = (xh[kk] - mean) * rstd * p.n1_g[a] + p.n1_b[a]; }
    }
    f32x4 oacc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) oacc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* gvb = p.gv + (size_t)b * p.heads * JT * 64 * FR + (size_t)lane * FR;
    const int jlast = p.J - 1;
    for (int h = 0; h < p.heads; ++h) {
        const float* gvh = gvb + (size_t)h * JT * 64 * FR;
        f32x4 oh[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) oh[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float m = -INFINITY, l = 0.0f;
        for (int jt0 = 0; jt0 < JT; jt0 += JTC) {
            f32x4 s[JTC];
#pragma unroll
            for (int t = 0; t < JTC; ++t) {
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (jt0 + t < JT) {
                    float g[NGP];
#pragma unroll
                    for (int v4 = 0; v4 < NGP / 4; ++v4) {
                        const float4 q4 = *reinterpret_cast<const float4*>(gvh + (size_t)(jt0 + t) * 64 * FR + 4 * v4);
                        g[4 * v4] = q4.x; g[4 * v4 + 1] = q4.y; g[4 * v4 + 2] = q4.z; g[4 * v4 + 3] = q4.w;
                    }
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(g[kk], xh[kk], acc, 0, 0, 0);
                }
                s[t] = acc;
            }
            float cm = -INFINITY;
#pragma unroll
            for (int t = 0; t < JTC; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (16 * (jt0 + t) + 4 * lg + r > jlast) s[t][r] = -INFINITY;
                    cm = fmaxf(cm, s[t][r]);
                }
            cm = fmaxf(cm, __shfl_xor(cm, 16));
            cm = fmaxf(cm, __shfl_xor(cm, 32));
            const float mn = fmaxf(m, cm);                    // finite from the first chunk on (row 0 is always valid)
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            float cl = 0.0f;
#pragma unroll
            for (int t = 0; t < JTC; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(s[t][r] - mn); s[t][r] = e; cl += e; }
            cl += __shfl_xor(cl, 16);
            cl += __shfl_xor(cl, 32);
            l = l * alpha + cl;
            m = mn;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) oh[mt][r] *= alpha;
#pragma unroll
            for (int t = 0; t < JTC; ++t) {
                if (jt0 + t < JT) {
                    float vw[4 * MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float4 q4 = *reinterpret_cast<const float4*>(gvh + (size_t)(jt0 + t) * 64 * FR + NGP + 4 * mt);
                        vw[4 * mt] = q4.x; vw[4 * mt + 1] = q4.y; vw[4 * mt + 2] = q4.z; vw[4 * mt + 3] = q4.w;
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) oh[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vw[4 * mt + r], s[t][r], oh[mt], 0, 0, 0);
                }
            }
        }
        const float linv = 1.0f / l;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[mt][r] = fmaf(oh[mt][r], linv, oacc[mt][r]);
    }
    // to_out.1 LayerNorm + residual + statistics (as in cross_attn_folded_kernel)
    float yv[4 * MT];
#pragma unroll
    for (int e = 0; e < 4 * MT; ++e) yv[e] = 0.0f;
    {
        float s1 = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s1 += oacc[mt][r];
        s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
        const float mean = s1 / (float)C;
        float v = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int a = 16 * mt + 4 * lg + r; const float d = (a < C) ? oacc[mt][r] - mean : 0.0f; v = fmaf(d, d, v); }
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        const float rstd = 1.0f / sqrtf(v / (float)C + 1e-5f);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = 16 * mt + 4 * lg + r;
                if (a < C && ok) {
                    const float y = (oacc[mt][r] - mean) * rstd * p.n2_g[a] + p.n2_b[a] + xb[(size_t)a * p.HW + i] * p.x.scale;
                    p.out[((size_t)b * C + a) * p.HW + i] = y;
                    yv[4 * mt + r] = y;
                }
            }
    }
    if (p.out_stats) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = 16 * mt + 4 * lg + r;
                double S, Q;
                mi_stat_reduce16(yv[4 * mt + r], a < C && ok, lane, S, Q);
                if (lq == 0) { red[wave][2 * a] = S; red[wave][2 * a + 1] = Q; }
            }
        __syncthreads();
        if (tid < 2 * C) p.out_stats[((size_t)(b * C + (tid >> 1)) * tiles + tile) * 2 + (tid & 1)] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// one work-item = one pixel; the (tiny) 1x1 weights are wave-uniform reads
template <int C, int CH>
__global__ __launch_bounds__(256) void chan_ff_kernel(const mi_chan_ff_params p, const float* __restrict__ w1, const float* __restrict__ w2,
                                                      const float* __restrict__ g1, const float* __restrict__ g2) {
    __shared__ double red[2 * C][4];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int bx = mi_row_of(b, p.x.bmod);
    const int i = blockIdx.x * 256 + tid;
    const bool ok = i < p.HW;
    const float* xb = p.x.data + (size_t)bx * C * p.HW + (ok ? i : 0);
    float x[C], hdn[CH];
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) { x[c] = xb[(size_t)c * p.HW] * p.x.scale; s += x[c]; }
    float mean = s / (float)C, v = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) { const float d = x[c] - mean; v = fmaf(d, d, v); }
    float rs = 1.0f / sqrtf(v / (float)C + 1e-5f);           // (x - mean) / sqrt(var + eps) * g   (layers.py:174-177)
    float xn[C];
#pragma unroll
    for (int c = 0; c < C; ++c) xn[c] = (x[c] - mean) * rs * g1[c];
    s = 0.0f;
#pragma unroll
    for (int o = 0; o < CH; ++o) {
        float a = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) a = fmaf(w1[o * C + c], xn[c], a);
        hdn[o] = gelu_erf(a);
        s += hdn[o];
    }
    mean = s / (float)CH; v = 0.0f;
#pragma unroll
    for (int o = 0; o < CH; ++o) { const float d = hdn[o] - mean; v = fmaf(d, d, v); }
    rs = 1.0f / sqrtf(v / (float)CH + 1e-5f);
#pragma unroll
    for (int o = 0; o < CH; ++o) hdn[o] = (hdn[o] - mean) * rs * g2[o];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float a = 0.0f;
#pragma unroll
        for (int o = 0; o < CH; ++o) a = fmaf(w2[c * CH + o], hdn[o], a);
        const float y = a + x[c];
        if (ok) p.out[((size_t)b * C + c) * p.HW + i] = y;
        if (p.out_stats) {
            double ys, yq;
            mi_stat_reduce64(ok ? y : 0.0f, ok, ys, yq);
            if ((tid & 63) == 0) { red[2 * c][tid >> 6] = ys; red[2 * c + 1][tid >> 6] = yq; }
        }
    }
    if (p.out_stats) {
        __syncthreads();
        if (tid < 2 * C) p.out_stats[((size_t)(b * C + (tid >> 1)) * gridDim.x + blockIdx.x) * 2 + (tid & 1)] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    }
}

}  // namespace

extern "C" int mi_ln_tokens_fwd(const mi_act* x, int B, int HW, const float* gamma, const float* beta, float* out, void* stream) {
    if (B <= 0 || HW <= 0) { mi_set_error("mi_ln_tokens_fwd: empty"); return MI_ERR_INVALID; }
    const int nt = (HW + 63) / 64, nblk = (x->C + 63) / 64;
    int split = 1;
    while (split < nblk && (long long)nt * B * split < 512) split *= 2;
    hipLaunchKernelGGL(ln_tokens_kernel, dim3(nt, B, split), dim3(256), 0, (hipStream_t)stream, *x, HW, gamma, beta, out);
    return mi_check_launch("ln_tokens_kernel");
}

extern "C" int mi_self_attn_fwd(const mi_self_attn_params* pp, void* stream) {
    const mi_self_attn_params& p = *pp;
    hipStream_t st = (hipStream_t)stream;
    if (p.B2 <= 0 || p.HW <= 0 || p.J != p.HW + 1) { mi_set_error("mi_self_attn_fwd: bad sizes (J must be HW+1)"); return MI_ERR_INVALID; }
    const int JT = (p.J + 15) / 16;
    const dim3 grid(((p.HW + 63) / 64) * p.B2);
    switch (p.C) {
        case 8: hipLaunchKernelGGL(HIP_KERNEL_NAME(self_attn_folded_kernel<8, 16>), grid, dim3(256), 0, st, p, JT); break;
        case 16: hipLaunchKernelGGL(HIP_KERNEL_NAME(self_attn_folded_kernel<16, 16>), grid, dim3(256), 0, st, p, JT); break;
        case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(self_attn_folded_kernel<32, 16>), grid, dim3(256), 0, st, p, JT); break;
        default: mi_set_error("mi_self_attn_fwd: folded path instantiated for C in {8,16,32}, got %d", p.C); return MI_ERR_UNSUPPORTED;
    }
    return mi_check_launch("self_attn_folded_kernel");
}

extern "C" int mi_chan_ff_fwd(const mi_chan_ff_params* pp, void* stream) {
    const mi_chan_ff_params& p = *pp;
    hipStream_t st = (hipStream_t)stream;
    if (p.B <= 0 || p.HW <= 0 || p.Chid != 2 * p.C) { mi_set_error("mi_chan_ff_fwd: bad sizes (hidden must be 2*C)"); return MI_ERR_INVALID; }
    const dim3 grid((p.HW + 255) / 256, p.B);
    switch (p.C) {
        case 8: hipLaunchKernelGGL(HIP_KERNEL_NAME(chan_ff_kernel<8, 16>), grid, dim3(256), 0, st, p, p.w1, p.w2, p.g1, p.g2); break;
        case 16: hipLaunchKernelGGL(HIP_KERNEL_NAME(chan_ff_kernel<16, 32>), grid, dim3(256), 0, st, p, p.w1, p.w2, p.g1, p.g2); break;
        case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(chan_ff_kernel<32, 64>), grid, dim3(256), 0, st, p, p.w1, p.w2, p.g1, p.g2); break;
        default: mi_set_error("mi_chan_ff_fwd: instantiated for C in {8,16,32}, got %d", p.C); return MI_ERR_UNSUPPORTED;
    }
    return mi_check_launch("chan_ff_kernel");
}
