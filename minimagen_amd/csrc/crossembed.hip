// K3: CrossEmbedLayer (MinImagen layers.py:254-305; Unet.py:169-172,396-400): three parallel
// stride-1 convolutions (k = 3, 7, 15) over the raw input image (x, or cat(x, lowres_cond_img)
// for super-resolution U-Nets), concatenated on the channel axis.  One kernel: the 15x15 halo
// tile is staged once per input channel in LDS and all three kernels consume it (the 7x7 and
// 3x3 windows are sub-windows of the rows already in registers); the two inputs are read from
// their own pointers (the concat is never materialised); per-channel partial statistics for the
// first GroupNorm come out of the epilogue.
#include "common.hip.h"
#include <type_traits>

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
    double ssum[COUT], ssq[COUT];               // fp64 statistics (common.hip.h)
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float* bp = co < 4 ? p.bias[0] : (co < 6 ? p.bias[1] : p.bias[2]);
        const float bv = bp ? bp[co < 4 ? co : (co < 6 ? co - 4 : co - 6)] : 0.0f;
        double s = 0.0, q = 0.0;
        if (row_ok) {
            float* dst = p.out + ((size_t)(b * COUT + co) * p.H + oy) * p.W + ox;
            const float* add = p.addend ? p.addend + ((size_t)(b * COUT + co) * p.H + oy) * p.W + ox : nullptr;
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const float v = acc[px][co] + bv + ((add && ox + px < p.W) ? add[px] : 0.0f);
                if (ox + px < p.W) { dst[px] = v; s += (double)v; q = fma((double)v, (double)v, q); }
            }
        }
        ssum[co] = s;
        ssq[co] = q;
    }
    if (p.out_stats) {
        static_assert(NT % 64 == 0 && 2 * COUT <= NT, "statistics reduction: whole waves");
        constexpr int NW = NT / 64;
        __syncthreads();
        double* const redd = reinterpret_cast<double*>(smem);          // [2 * COUT][NW]
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            double s = ssum[co], q = ssq[co];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
            if ((tid & 63) == 0) { redd[(2 * co) * NW + (tid >> 6)] = s; redd[(2 * co + 1) * NW + (tid >> 6)] = q; }
        }
        __syncthreads();
        if (tid < 2 * COUT) {
            double a = 0.0;
            for (int w = 0; w < NW; ++w) a += redd[tid * NW + w];
            p.out_stats[((size_t)(b * COUT + (tid >> 1)) * gridDim.x + tile) * 2 + (tid & 1)] = a;
        }
    }
}

// ---- matrix cores: dim_scales (4,2,2), kernel sizes (3,7,15), <= 4 input channels ----
// The three convs as ONE Toeplitz GEMM per group of 16 pixels x 8 output rows on v_mfma_f32_16x16x32_f16:
//   D[px m][(co2, dy)] += A[m][(r; lg, dx, ci)] . B[(r; lg, dx, ci)][(co2, dy)]        r = input row of the group's 22-row window
//   A = act[row r][column x0 - 7 + m + 2 lg + 8 h + dx][ci]    (16 bytes per lane: two adjacent pixels x 4 channels, fp16)
//   B = W[co][ci][ky = r - dy - off][kx = 2 lg + 8 h + dx - off], off = 7 - pad  -- depends on r - dy only, so the table in LDS has
//       one row per vertical tap and every lane reads its own row (packing.pack_crossembed_mfma)
// Four N tiles share every A fragment: k3 channels 0-1, k3 channels 2-3, k7, k15 (the narrower kernels skip the rows / halves where
// their taps are all zero).  fp32 operands as 3-term fp16 splits; the inputs are scaled per workgroup by a power of two taken from
// the staged tile's max |x| (exact, undone in the epilogue), the weights at pack time -- any input range is safe.
typedef _Float16 ce_f16x8 __attribute__((ext_vector_type(8)));

#ifdef MI_TRACE
// development aid (tools/bench_ce.py): shader-clock time per phase of the first workgroups of the last launch + wall-clock start / end
__device__ unsigned long long mi_trace_ce_buf[1024 * 8];
extern "C" int mi_debug_read_trace_ce(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mi_trace_ce_buf), bytes); }
#define CE_TSTART() const unsigned long long ce_w0 = wall_clock64(); unsigned long long ce_last = clock64(), ce_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define CE_TPHASE(k) do { const unsigned long long n_ = clock64(); ce_acc[k] += n_ - ce_last; ce_last = n_; } while (0)
#define CE_TEND() do { ce_acc[7] = (ce_w0 << 32) | (wall_clock64() & 0xffffffffull); const int wg_ = blockIdx.y * gridDim.x + blockIdx.x; if (threadIdx.x == 0 && wg_ < 1024) for (int k = 0; k < 8; ++k) mi_trace_ce_buf[wg_ * 8 + k] = ce_acc[k]; } while (0)
#else
#define CE_TSTART() do { } while (0)
#define CE_TPHASE(k) do { } while (0)
#define CE_TEND() do { } while (0)
#endif

template <int GY, int GX, bool HALF>
__global__ __launch_bounds__(256, 2) void crossembed_mfma_kernel(const mi_crossembed_params p, const uint4* __restrict__ wtab) {
    constexpr int TH = 8 * GY, TW = 16 * GX, G = GY * GX / 4;             // 4 waves x G groups of (16 px x 8 rows)
    constexpr int IH = TH + 14, PW = TW + 16, NW4 = PW / 4, NU = IH * NW4, PER = (NU + 255) / 256;
    constexpr int TP = 33;                                                // table row pitch in 16-byte chunks: 32 + 1, so the 8 rows
                                                                          // (r - dy) a wave reads at once fall into different banks
    __shared__ __attribute__((aligned(16))) uint2 actH[IH * PW];
    __shared__ __attribute__((aligned(16))) uint2 actL[HALF ? 2 : IH * PW];
    __shared__ __attribute__((aligned(16))) uint4 tab[32 * TP];
    __shared__ double red[4][16];
    __shared__ float smax[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int H = p.H, W = p.W, HW = H * W;
    const int tiles_x = (W + TW - 1) / TW, tile = blockIdx.x;
    const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
    const int b = blockIdx.y, Cin = p.C0;
    const int b0 = p.in0_batch_mod > 0 ? b % p.in0_batch_mod : b;
    const float* src = p.in0 + (size_t)b0 * Cin * HW;

    CE_TSTART();
    // ---- stage the (TH + 14) x (TW + 16) window: one unit = 4 pixels x all channels
    float4 raw[PER][4];
    unsigned inm = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int q = tid + u * 256, iy = q / NW4, xq = q - iy * NW4;
        const int gy = oy0 - 7 + iy, gx = ox0 - 8 + 4 * xq;
        const bool in = q < NU && gy >= 0 && gy < H && gx >= 0 && gx < W;          // W % 4 == 0: a float4 is inside or outside as a whole
        inm |= in ? (1u << u) : 0u;
        const int off = in ? gy * W + gx : 0;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) raw[u][ci] = ci < Cin ? mi_ldg4(src + (size_t)ci * HW + off) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = tid; i < 32 * 32; i += 256) tab[(i >> 5) * TP + (i & 31)] = mi_ldg4u(wtab + i);
    // the addend (the step-invariant low-res half of the convolution) is as many bytes as the output: requested now, consumed in the
    // epilogue, so that its latency hides under the GEMM
    const int dy = lq & 7, co2 = lq >> 3;
    float4 addv[G][4];
    if (p.addend) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int gi = wave * G + g, gyy = gi / GX, gxx = gi % GX;
            const int oy = oy0 + 8 * gyy + dy, ox = ox0 + 16 * gxx + 4 * lg;
            const bool ok = oy < H && ox < W;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const size_t o = ((size_t)(b * 8 + 2 * t + co2) * H + (ok ? oy : 0)) * W + (ok ? ox : 0);
                if (HALF && p.out_st) addv[g][t] = mi_bf16x4_to_f32(mi_ldg2u(reinterpret_cast<const unsigned short*>(p.addend) + o));     // bf16 storage
                else addv[g][t] = mi_ldg4(p.addend + o);
            }
        }
    }
    CE_TPHASE(0);
    float m = 0.0f;
#pragma unroll
    for (int u = 0; u < PER; ++u)
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
            const float4 v = raw[u][ci];
            const float a = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
            m = ((inm >> u) & 1u) ? fmaxf(m, a) : m;
        }
    m = mi_wave_max(m);
    if (lane == 0) smax[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    int ex = 0;                                     // max |x| 2^ex in [128, 256); no scaling for an all-zero or non-finite tile
    {
        const int be = (int)((__float_as_uint(m) & 0x7fffffffu) >> 23);
        if (be != 0 && be != 255) ex = 8 - (be - 126);
    }
    const float sc = ldexpf(1.0f, ex);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int q = tid + u * 256, iy = q / NW4, xq = q - iy * NW4;
        if (q >= NU) continue;
        const bool in = (inm >> u) & 1u;
        uint2 hh[4], ll[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y[4];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                const float4 v = raw[u][ci];
                const float x = j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w));
                y[ci] = in ? x * sc : 0.0f;                                              // zero padding of the conv
            }
            mi_f16x4 h4, l4;
            mi_split_f16(y, h4, l4);
            hh[j] = __builtin_bit_cast(uint2, h4);
            ll[j] = __builtin_bit_cast(uint2, l4);
        }
        const int d = iy * PW + 4 * xq;
        *reinterpret_cast<uint4*>(&actH[d]) = make_uint4(hh[0].x, hh[0].y, hh[1].x, hh[1].y);
        *reinterpret_cast<uint4*>(&actH[d + 2]) = make_uint4(hh[2].x, hh[2].y, hh[3].x, hh[3].y);
        if constexpr (!HALF) {
            *reinterpret_cast<uint4*>(&actL[d]) = make_uint4(ll[0].x, ll[0].y, ll[1].x, ll[1].y);
            *reinterpret_cast<uint4*>(&actL[d + 2]) = make_uint4(ll[2].x, ll[2].y, ll[3].x, ll[3].y);
        }
    }
    __syncthreads();
    CE_TPHASE(1);

    // ---- the GEMM.  One step = one K = 32 instruction triple per group and N tile:
    //   k15: (window row r, column half h): the 8 column positions 2 lg + dx + 8 h of the 16-column window, 4 channels each;
    //   k7:  rides on the k15 fragments of window rows 4 .. 17 (its taps are positions 4 .. 10: both halves);
    //   k3:  its own dense step per window row PAIR (r, r + 1) and N tile (channels 0-1, 2-3): lane groups 0-1 take row r, 2-3 row r + 1,
    //        positions 6 .. 9 (the 4th is a zero weight).  (On the k15 fragments the k3 member cost 40 steps x 2 tiles per group; dense: 5 x 2.)
    // The loop is bound by LDS reads (A fragments: 2 KB per step and group), then by the matrix pipe: 1344 -> 984 instructions and 576 -> 476 KB
    // of LDS reads per wave against round 3's all-on-k15 form; an own (aligned) step for k7 as well would cut the instructions to 816 but
    // read 640 KB -- measured slower.
    // The operands of step s + 1 are read from LDS while the MFMAs of step s run (two register sets); the rows are walked in row pairs with
    // a compile-time step list, so that there is no branch between the reads and their use.
    f32x4 acc[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[g][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int bch = co2 * 4 + lg;                           // this lane's chunk within a (row, h, hi | lo) block of the table
    struct Ops { ce_f16x8 bh[2], bl[2], ah[G], al[G]; };
    // KIND 0 / 1: k15 half 0 / 1 (+ k7 when W7), 3: k3
    auto load = [&](auto kind_tag, auto w7_tag, int r, Ops& o) {
        constexpr int KIND = decltype(kind_tag)::value;
        constexpr bool W7 = decltype(w7_tag)::value;
        constexpr int NB = (KIND == 3 || W7) ? 2 : 1;
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            int row, hsel = 0;
            if constexpr (KIND <= 1) {
                hsel = KIND;
                if (t == 0) { const int q = r - dy; row = 16 + ((unsigned)q < 15u ? q : 15); }
                else { const int q = r - dy - 4; row = 8 + ((unsigned)q < 7u ? q : 7); }
            } else { const int q = r - dy - 5; row = (unsigned)q < 4u ? 4 * t + q : 31; }      // q = q' + 1, q' = r - dy - 6 in -1 .. 2; else the zero row
            o.bh[t] = __builtin_bit_cast(ce_f16x8, tab[row * TP + (2 * hsel) * 8 + bch]);
            if constexpr (!HALF) o.bl[t] = __builtin_bit_cast(ce_f16x8, tab[row * TP + (2 * hsel + 1) * 8 + bch]);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int gi = wave * G + g, gyy = gi / GX, gxx = gi % GX;
            int idx = 16 * gxx + 1 + lq;
            if constexpr (KIND <= 1) idx += (8 * gyy + r) * PW + 2 * lg + 8 * KIND;
            else idx += (8 * gyy + r + (lg >> 1)) * PW + 6 + 2 * (lg & 1);
            const uint2 h0 = actH[idx], h1 = actH[idx + 1];
            o.ah[g] = __builtin_bit_cast(ce_f16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
            if constexpr (!HALF) {
                const uint2 l0 = actL[idx], l1 = actL[idx + 1];
                o.al[g] = __builtin_bit_cast(ce_f16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
            }
        }
    };
    auto mma = [&](auto kind_tag, auto w7_tag, const Ops& o) {
        constexpr int KIND = decltype(kind_tag)::value;
        constexpr bool W7 = decltype(w7_tag)::value;
        constexpr int NB = (KIND == 3 || W7) ? 2 : 1;
        // term-major over the wave's groups: consecutive MFMAs write different accumulators (no back-to-back dependency)
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int T = KIND == 3 ? t : 3 - t;                    // accumulator: k3 -> 0, 1; k15 -> 3, k7 -> 2
            if constexpr (!HALF) {
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g][T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.al[g], o.bh[t], acc[g][T], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g][T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.ah[g], o.bl[t], acc[g][T], 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g][T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.ah[g], o.bh[t], acc[g][T], 0, 0, 0);
        }
    };
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K3 = std::integral_constant<int, 3>;
    using Y = std::true_type; using N = std::false_type;
    // One row PAIR (r, r + 1), r even: steps k15(r, 0) k15(r, 1) [k3(r)] k15(r + 1, 0) k15(r + 1, 1).  `x` holds the operands of the first
    // step on entry; the operands of k15(rn, 0) -- the next pair's first step, with k7 iff W7N -- are loaded under the last step's MFMAs and
    // end up in `x` again (even step count), or in `y` for the 5-step form.
    auto pair4 = [&](auto w7, auto w7n, int r, int rn, Ops& x, Ops& y) {
        load(K1{}, w7, r, y);        mma(K0{}, w7, x);
        load(K0{}, w7, r + 1, x);    mma(K1{}, w7, y);
        load(K1{}, w7, r + 1, y);    mma(K0{}, w7, x);
        load(K0{}, w7n, rn, x);      mma(K1{}, w7, y);
    };
    auto pair5 = [&](auto w7n, int r, int rn, Ops& x, Ops& y) {
        load(K1{}, Y{}, r, y);        mma(K0{}, Y{}, x);
        load(K3{}, N{}, r, x);        mma(K1{}, Y{}, y);
        load(K0{}, Y{}, r + 1, y);    mma(K3{}, N{}, x);
        load(K1{}, Y{}, r + 1, x);    mma(K0{}, Y{}, y);
        load(K0{}, w7n, rn, y);       mma(K1{}, Y{}, x);
    };
    {
        Ops oa, ob;
        load(K0{}, N{}, 0, oa);
        pair4(N{}, N{}, 0, 2, oa, ob);                          // rows that hold taps of the 15x15 kernel only
        pair4(N{}, Y{}, 2, 4, oa, ob);
        pair4(Y{}, Y{}, 4, 6, oa, ob);                          // + the 7x7 kernel (window rows 4 .. 17)
#pragma unroll 1
        for (int r = 6; r < 14; r += 4) {                       // + the 3x3 kernel (window rows 6 .. 15, one step per row pair)
            pair5(Y{}, r, r + 2, oa, ob);
            pair5(Y{}, r + 2, r + 4, ob, oa);
        }
        pair5(Y{}, 14, 16, oa, ob);
        pair4(Y{}, N{}, 16, 18, ob, oa);
        pair4(N{}, N{}, 18, 20, ob, oa);
        pair4(N{}, N{}, 20, 20, ob, oa);                        // (the last load is a harmless re-read: no branch around the reads)
    }
    CE_TPHASE(2);

    // ---- epilogue: lane (lq = 8 co2 + dy, lg) holds pixels 4 lg .. 4 lg + 3 of output row dy, channel 2 t + co2 of every group
    // statistics about a per-channel shift (the first value of the channel's first lane): common.hip.h mi_stat_acc; reduced per N tile
    // right away (nothing kept across the loop); the pixel count is wave-uniform geometry
    int wcnt = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int gi = wave * G + g, gyy = gi / GX, gxx = gi % GX;
        const int wy = H - (oy0 + 8 * gyy), wx = W - (ox0 + 16 * gxx);
        wcnt += (wy < 0 ? 0 : (wy > 8 ? 8 : wy)) * (wx < 0 ? 0 : (wx > 16 ? 16 : wx));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = t < 2 ? 0 : t - 1;                                   // conv of this N tile
        const int c = 2 * t + co2;                                          // channel in the concatenated output
        const float us = ldexpf(1.0f, -(ex + p.w_mfma_exp[k]));
        const float* bp = p.bias[k];
        const float bv = bp ? bp[t == 1 ? 2 + co2 : co2] : 0.0f;
        float s = 0.0f, q2 = 0.0f;
        float sh = 0.0f;
        if (p.out_stats) sh = __shfl(fmaf(acc[0][t][0], us, bv) + (p.addend ? addv[0][t].x : 0.0f), 8 * co2);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int gi = wave * G + g, gyy = gi / GX, gxx = gi % GX;
            const int oy = oy0 + 8 * gyy + dy, ox = ox0 + 16 * gxx + 4 * lg;
            const bool ok = oy < H && ox < W;
            const size_t o = ((size_t)(b * 8 + c) * H + (ok ? oy : 0)) * W + (ok ? ox : 0);
            float4 y = make_float4(fmaf(acc[g][t][0], us, bv), fmaf(acc[g][t][1], us, bv), fmaf(acc[g][t][2], us, bv), fmaf(acc[g][t][3], us, bv));
            if (p.addend) { const float4 a = addv[g][t]; y.x += a.x; y.y += a.y; y.z += a.z; y.w += a.w; }
            if (ok) {
                if (HALF && p.out_st) mi_stg2u(reinterpret_cast<unsigned short*>(p.out) + o, mi_f32x4_to_bf16(y));
                else mi_stg4(p.out + o, y);
                const float d0 = y.x - sh, d1 = y.y - sh, d2 = y.z - sh, d3 = y.w - sh;
                s += (d0 + d1) + (d2 + d3);
                q2 += fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, d3 * d3)));
            }
        }
        if (p.out_stats) {
#pragma unroll
            for (int o = 1; o <= 4; o <<= 1) { s += __shfl_xor(s, o); q2 += __shfl_xor(q2, o); }
            s += __shfl_xor(s, 16); q2 += __shfl_xor(q2, 16);
            s += __shfl_xor(s, 32); q2 += __shfl_xor(q2, 32);
            if (dy == 0 && lg == 0) {
                mi_stat_acc a; a.c = sh; a.s = s; a.q = q2; a.n = wcnt;
                mi_stat_finish(a, red[wave][2 * (2 * t + co2)], red[wave][2 * (2 * t + co2) + 1]);
            }
        }
    }
    if (p.out_stats) {
        __syncthreads();
        if (tid < 16) p.out_stats[((size_t)(b * 8 + (tid >> 1)) * gridDim.x + tile) * 2 + (tid & 1)] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
    CE_TPHASE(3);
    CE_TEND();
}

// ---- generic (any dim / kernel sizes): one work-item per output pixel, CPT output channels of one member per blockIdx.z; taps read
// through L1/L2.  Correct for every constructor argument.  CPT = 8 where every member's channel count is a multiple of 8 (the wide presets:
// dim 128 -> 64 + 32 + 32): the tap's load and bounds logic serve eight FMAs, the weights of a tap are one wave-uniform 32-byte load -- Unet()'s
// CrossEmbed 0.80 -> 0.2 ms; the accumulation order per output is unchanged (channel, row, column).
template <int NT, int TW, int CPT>
__global__ __launch_bounds__(NT) void crossembed_generic_kernel(const mi_crossembed_params p, const int Ctot) {
    constexpr int TH = NT / TW;
    __shared__ double red[2][CPT][NT / 64];
    const int tid = threadIdx.x;
    const int tiles_x = (p.W + TW - 1) / TW;
    const int tile = blockIdx.x;
    const int oy = (tile / tiles_x) * TH + tid / TW, ox = (tile % tiles_x) * TW + tid % TW;
    const int b = blockIdx.y, co_g = blockIdx.z * CPT;
    int ki = 0, co = co_g;
    while (ki < p.n_kernels - 1 && co >= p.cout[ki]) { co -= p.cout[ki]; ++ki; }
    const int K = p.ksize[ki], pad = (K - 1) / 2, CO = p.cout[ki];
    const int C0 = p.C0, C1 = p.in1 ? p.C1 : 0, Cin = C0 + C1;
    const int b0 = p.in0_batch_mod > 0 ? b % p.in0_batch_mod : b;
    const int b1 = p.in1_batch_mod > 0 ? b % p.in1_batch_mod : b;
    float acc[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) acc[j] = 0.0f;
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
                    const float x = src[(size_t)gy * p.W + gx];
                    const float* wt = wc + (ky * K + kx) * CO;
#pragma unroll
                    for (int j = 0; j < CPT; ++j) acc[j] = fmaf(x, wt[j], acc[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            if (p.bias[ki]) acc[j] += p.bias[ki][co + j];
            if (p.addend) acc[j] += p.addend[((size_t)(b * Ctot + co_g + j) * p.H + oy) * p.W + ox];
            p.out[((size_t)(b * Ctot + co_g + j) * p.H + oy) * p.W + ox] = acc[j];
        }
    }
    if (p.out_stats) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            double s = ok ? (double)acc[j] : 0.0, q = s * s;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
            if ((tid & 63) == 0) { red[0][j][tid >> 6] = s; red[1][j][tid >> 6] = q; }
        }
        __syncthreads();
        if (tid < 2 * CPT) {
            const int j = tid >> 1, which = tid & 1;
            double a = 0.0;
            for (int w = 0; w < NT / 64; ++w) a += red[which][j][w];
            p.out_stats[((size_t)(b * Ctot + co_g + j) * gridDim.x + tile) * 2 + which] = a;
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
    if (mi_conv_tile_shape(p.tile_cfg & 0xff, &th, &tw) != MI_OK) { mi_set_error("mi_crossembed_fwd: bad tile_cfg"); return MI_ERR_INVALID; }
    const int tiles = ((p.H + th - 1) / th) * ((p.W + tw - 1) / tw);
    if (p.out_st && !(p.w_mfma && (p.tile_cfg & 0x400))) { mi_set_error("mi_crossembed_fwd: bf16 storage needs the matrix-core kernel with tile_cfg | 0x400"); return MI_ERR_UNSUPPORTED; }
    if (p.w_mfma) {             // matrix-core kernel: tile_cfg 8 (32 x 64) / 9 (16 x 32), | 0x400 = single fp16 term
        const int cfg = p.tile_cfg & 0xff;
        if (!fast || p.in1 || p.C0 < 1 || p.C0 > 4 || (p.W & 3) || (cfg != 8 && cfg != 9)) {
            mi_set_error("mi_crossembed_fwd: the matrix-core kernel is built for dim_scales (4,2,2), kernel sizes (3,7,15), one input of <= 4 channels, W %% 4 == 0, tile_cfg 8 / 9");
            return MI_ERR_UNSUPPORTED;
        }
        const bool half = (p.tile_cfg & 0x400) != 0;
        const uint4* wt = (const uint4*)p.w_mfma;
        if (cfg == 8) {
            if (half) hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_mfma_kernel<4, 4, true>), dim3(tiles, p.B), dim3(256), 0, st, p, wt);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_mfma_kernel<4, 4, false>), dim3(tiles, p.B), dim3(256), 0, st, p, wt);
        } else {
            if (half) hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_mfma_kernel<2, 2, true>), dim3(tiles, p.B), dim3(256), 0, st, p, wt);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_mfma_kernel<2, 2, false>), dim3(tiles, p.B), dim3(256), 0, st, p, wt);
        }
        return mi_check_launch("crossembed_mfma");
    }
    if (fast) {
        switch (p.tile_cfg) {
            case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_422_kernel<256, 64>), dim3(tiles, p.B), dim3(256), 0, st, p, p.w[0], p.w[1], p.w[2]); break;
            case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_422_kernel<256, 32>), dim3(tiles, p.B), dim3(256), 0, st, p, p.w[0], p.w[1], p.w[2]); break;
            default: hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_422_kernel<64, 32>), dim3(tiles, p.B), dim3(64), 0, st, p, p.w[0], p.w[1], p.w[2]); break;
        }
    } else {
        // generic tiles: same (th x tw) footprint so out_nt matches mi_conv_tile_shape; th*tw work-items <= 1024
        bool oct = true;
        for (int i = 0; i < p.n_kernels; ++i) oct = oct && (p.cout[i] % 8) == 0;
        if (oct) {
            switch (p.tile_cfg) {
                case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_generic_kernel<1024, 64, 8>), dim3(tiles, p.B, Ctot / 8), dim3(1024), 0, st, p, Ctot); break;
                case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_generic_kernel<1024, 32, 8>), dim3(tiles, p.B, Ctot / 8), dim3(1024), 0, st, p, Ctot); break;
                default: hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_generic_kernel<256, 32, 8>), dim3(tiles, p.B, Ctot / 8), dim3(256), 0, st, p, Ctot); break;
            }
        } else {
            switch (p.tile_cfg) {
                case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_generic_kernel<1024, 64, 1>), dim3(tiles, p.B, Ctot), dim3(1024), 0, st, p, Ctot); break;
                case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_generic_kernel<1024, 32, 1>), dim3(tiles, p.B, Ctot), dim3(1024), 0, st, p, Ctot); break;
                default: hipLaunchKernelGGL(HIP_KERNEL_NAME(crossembed_generic_kernel<256, 32, 1>), dim3(tiles, p.B, Ctot), dim3(256), 0, st, p, Ctot); break;
            }
        }
    }
    return mi_check_launch("crossembed");
}
