// Micro-benchmark (dev tool, round 6): the DATA-MOVEMENT + PIPELINE skeleton of a full-width-stripe, wave-specialised 3x3 conv for the
// narrow layers, measured before building the real kernel.
//
//   in / res / out: [B][8][H][W] fp32 (NCHW).  One workgroup owns RS consecutive output rows of one image at FULL width: no horizontal
//   halo, vertical halo (RS + 2) / RS.  It walks the stripe two output rows at a time (the row-paired MFMA form of conv_rp.hip: 4 input
//   rows -> 2 output rows), keeping a ring of 6 transformed input rows in LDS (fp16 hi / lo planes, pixel-major 16-byte chunks).
//   Waves are specialised:
//     loader waves (NLW = W / 128): a lane owns 4 consecutive pixels of one new input row and ALL 8 channels = 8 dwordx4 loads per step,
//       each wave-instruction a whole 1 KB row run of one channel plane; NB steps in flight in registers; GroupNorm-affine + SiLU + fp16
//       split in registers, 4 pixel chunks (16 B hi + 16 B lo) written to the ring;
//     consumer waves (4): A fragments from the ring (conflict-free 16-byte reads, lane group <-> row permutation 0,2,1,3), weights in
//       registers, 9 x v_mfma_f32_16x16x32_f16 per 16-pixel group, identity-residual loads one step ahead, bias + residual + stores in
//       the MFMA layout (64-byte segments: measured equal to coalesced rows in profiles/r05_conv_dma_ablation.txt).
//   One s_barrier per step; per-workgroup time per step = max(loader, consumer), not their sum.
//
// Variants (template / runtime): RES (identity residual), XF (0: copy only, no transform arithmetic; 1: full transform), MF (MFMA loop on / off).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <class T> using gptr = T __attribute__((address_space(1)))*;

__device__ __forceinline__ unsigned split_lo2(unsigned hb, float x0, float x1) {
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hb), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hb), "v"(x1));
    const f16x2 l2 = {(_Float16)l0, (_Float16)l1};
    return __builtin_bit_cast(unsigned, l2);
}

// KO = input-channel octets, NJ = output-channel octets (N tiles); B fragments in LDS ([KO][3 taps][NJ][hi | lo][64 lanes]) as in conv_rp.hip
template <int W, int KO, int NJ, int RS, int NB, bool RES, int XF, bool MF>
__global__ __launch_bounds__((((2 * (W / 4) * KO + 63) / 64) + (W / 16 >= 4 ? 4 : W / 16)) * 64) void stripe_k(
        const float* __restrict__ in_, const float* __restrict__ res_, float* __restrict__ out_, const double* __restrict__ stats,
        const uint4* __restrict__ wfrag, int H) {
    constexpr int QPR = W / 4, UNITS = 2 * QPR * KO, NLW = (UNITS + 63) / 64;
    constexpr int NG = W / 16, NCW = NG >= 4 ? 4 : NG, GPW = NG / NCW;
    constexpr int PW = W + 8, RING = 6, CI = 8 * KO, CO = 8 * NJ;
    constexpr int NSTEP = RS / 2;
    static_assert(NSTEP % NB == 0 && NSTEP % 2 == 0, "steps per stripe must be a multiple of the unroll factors");
    __shared__ __attribute__((aligned(16))) uint4 actH[KO * RING * PW];
    __shared__ __attribute__((aligned(16))) uint4 actL[KO * RING * PW];
    __shared__ __attribute__((aligned(16))) uint4 wl[KO * 3 * NJ * 128];
    __shared__ float4 chP[CI];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int stripes = H / RS;
    const int b = blockIdx.x / stripes, y0 = (blockIdx.x % stripes) * RS;
    const size_t HW = (size_t)H * W;
    const gptr<const float> in = (gptr<const float>)(in_ + (size_t)b * CI * HW);
    const gptr<const float> res = (gptr<const float>)(res_ + (size_t)b * CO * HW);
    const gptr<float> out = (gptr<float>)(out_ + (size_t)b * CO * HW);
    // zero the left / right pad chunk of every ring row once; B fragments to LDS
    if (tid < KO * RING * 2) {
        const int r = tid >> 1, c = (tid & 1) ? W + 1 : 0;
        actH[r * PW + c] = make_uint4(0, 0, 0, 0);
        actL[r * PW + c] = make_uint4(0, 0, 0, 0);
    }
    for (int k = tid; k < KO * 3 * NJ * 128; k += (NLW + NCW) * 64) wl[k] = wfrag[k];
    if (wave < NLW) {
        // ------------------------------------------------ loader / transform waves
        const int u = wave * 64 + lane;
        const bool live = u < UNITS;
        const int uu = live ? u : 0;
        const int oct = uu / (2 * QPR), lrow = (uu / QPR) & 1, q = uu % QPR;        // channel octet, which of the step's 2 new rows, pixel quad
        f32x4 raw[NB][8];
        bool inimg[NB];
        auto issue = [&](int s, auto buf_tag) {          // step s loads input rows y0 + 2 s + 1 + lrow  (s = -1 .. NSTEP - 1)
            constexpr int buf = decltype(buf_tag)::value;
            int y = y0 + 2 * s + 1 + lrow;
            y = y > y0 + RS ? y0 + RS : y;              // steps past the stripe (issued unconditionally: a CONDITIONAL issue makes the compiler's wait counts conservative) re-read its last halo row
            const bool ok = y >= 0 && y < H;
            inimg[buf] = ok;
            const unsigned off = ok ? (unsigned)(y * W + 4 * q) : 0u;
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[buf][j] = *reinterpret_cast<gptr<const f32x4>>(in + (size_t)(8 * oct + j) * HW + off);
        };
        // the first steps' bulk loads go out BEFORE the statistics round trip (vmcnt is in order: the statistics are issued first)
        double2 sv = *reinterpret_cast<const double2*>(stats + 2 * ((size_t)(b * CI + (lane % CI)) * 8 + lane / CI % 8));
        issue(-1, std::integral_constant<int, NB - 1>{});
        [&]<int... I>(std::integer_sequence<int, I...>) { (issue(I, std::integral_constant<int, I % NB>{}), ...); }(std::make_integer_sequence<int, NB - 1>{});
        // statistics -> per-channel affine (emulated: one dependent global round trip + a shuffle reduction + LDS + barrier among the loaders)
        double sx = sv.x, sy = sv.y;
#pragma unroll
        for (int o = 32; o >= CI; o >>= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); }
        if (wave == 0 && lane < CI) {
            const float mean = (float)(sx * 1e-30), rstd = 1.0f / sqrtf((float)(sy * 1e-30) + 1.0f);
            chP[lane] = make_float4(1.5f * rstd, 0.1f + mean, -2.1f * rstd, -0.14f + mean);
        }
        __syncthreads();                                 // barrier #0 (all waves): chP, wl, pads visible
        float4 P[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) P[j] = chP[8 * oct + j];
        auto transform = [&](int s, auto buf_tag) {
            constexpr int buf = decltype(buf_tag)::value;
            const int k = 2 * s + 2 + lrow;             // ring row index relative to y0 - 1
            const int slot = k % RING;
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = raw[buf][j][px];
                    if (XF) {
                        const float a = fmaf(x, P[j].x, P[j].y);
                        const float ex = __builtin_amdgcn_exp2f(fmaf(x, P[j].z, P[j].w));
                        y[j] = a * __builtin_amdgcn_rcpf(1.0f + ex);
                    } else {
                        y[j] = x;
                    }
                }
                unsigned h[4], l[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2 v = {y[2 * i], y[2 * i + 1]};
                    h[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
                    l[i] = XF ? split_lo2(h[i], y[2 * i], y[2 * i + 1]) : __float_as_uint(y[2 * i]);
                }
                uint4 hv = make_uint4(h[0], h[1], h[2], h[3]), lv = make_uint4(l[0], l[1], l[2], l[3]);
                if (!inimg[buf]) { hv = make_uint4(0, 0, 0, 0); lv = hv; }
                if (live) {
                    actH[(oct * RING + slot) * PW + 1 + 4 * q + px] = hv;
                    actL[(oct * RING + slot) * PW + 1 + 4 * q + px] = lv;
                }
            }
        };
        // prologue: steps -1 .. NB - 1 in flight, steps -1 and 0 transformed before the first compute step
        transform(-1, std::integral_constant<int, NB - 1>{});
        issue(NB - 1, std::integral_constant<int, NB - 1>{});
        transform(0, std::integral_constant<int, 0>{});
        issue(NB, std::integral_constant<int, 0>{});
        __syncthreads();
        // step `it`: the consumers compute rows of step it; we transform step it + 1 (buffer (it + 1) % NB) and re-issue it with step it + 1 + NB
        for (int it0 = 0; it0 < NSTEP; it0 += NB) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ([&] {
                    const int it = it0 + I;
                    constexpr int buf = (I + 1) % NB;
                    transform(it + 1, std::integral_constant<int, buf>{});          // (the step past the last one lands in ring rows nobody reads)
                    issue(it + 1 + NB, std::integral_constant<int, buf>{});
                    __syncthreads();
                }(), ...);
            }(std::make_integer_sequence<int, NB>{});
        }
    } else {
        // ------------------------------------------------ MFMA / epilogue waves
        const int cw = wave - NLW;
        const int perm = ((lg & 1) << 1) | (lg >> 1);
        const int dy = lq >> 3;
        f32x4 rv[2][GPW][NJ];
        auto issue_res = [&](int it, auto buf_tag) {
            constexpr int buf = decltype(buf_tag)::value;
            if (!RES) return;
#pragma unroll
            for (int g = 0; g < GPW; ++g)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    const int co = 8 * jt + (lq & 7);
                    const int oy = y0 + 2 * (it < NSTEP ? it : NSTEP - 1) + dy, ox = 16 * (cw * GPW + g) + 4 * lg;
                    rv[buf][g][jt] = *reinterpret_cast<gptr<const f32x4>>(res + (size_t)co * HW + (unsigned)(oy * W + ox));
                }
        };
        float csum[NJ], csq[NJ];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) { csum[jt] = 0.f; csq[jt] = 0.f; }
        auto compute = [&](int it, auto buf_tag) {
            constexpr int buf = decltype(buf_tag)::value;
            f32x4 acc[GPW][NJ];
#pragma unroll
            for (int g = 0; g < GPW; ++g)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) acc[g][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int slot = (2 * it + perm) % RING;
#pragma unroll
            for (int o = 0; o < KO; ++o)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    f16x8 bh[NJ], bl[NJ];
#pragma unroll
                    for (int jt = 0; jt < NJ; ++jt) {
                        bh[jt] = __builtin_bit_cast(f16x8, wl[((o * 3 + s) * NJ + jt) * 128 + lane]);
                        bl[jt] = __builtin_bit_cast(f16x8, wl[((o * 3 + s) * NJ + jt) * 128 + 64 + lane]);
                    }
#pragma unroll
                    for (int g = 0; g < GPW; ++g) {
                        const int idx = (o * RING + slot) * PW + 16 * (cw * GPW + g) + lq + s;
                        const f16x8 ah = __builtin_bit_cast(f16x8, actH[idx]);
                        const f16x8 al = __builtin_bit_cast(f16x8, actL[idx]);
#pragma unroll
                        for (int jt = 0; jt < NJ; ++jt) {
                            if (MF) {
                                acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[jt], acc[g][jt], 0, 0, 0);
                                acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[jt], acc[g][jt], 0, 0, 0);
                                acc[g][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[jt], acc[g][jt], 0, 0, 0);
                            } else {
                                acc[g][jt][0] += (float)ah[0] + (float)al[0] + (float)bh[jt][0]; acc[g][jt][1] += (float)ah[1]; acc[g][jt][2] += (float)ah[2]; acc[g][jt][3] += (float)al[3] + (float)bl[jt][1];
                            }
                        }
                    }
                }
#pragma unroll
            for (int g = 0; g < GPW; ++g)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    const int co = 8 * jt + (lq & 7);
                    const int oy = y0 + 2 * it + dy, ox = 16 * (cw * GPW + g) + 4 * lg;
                    f32x4 y = acc[g][jt] * 0.25f + 0.5f;
                    if (RES) y += rv[buf][g][jt];
                    *reinterpret_cast<gptr<f32x4>>(out + (size_t)co * HW + (unsigned)(oy * W + ox)) = y;
                    csum[jt] += (y[0] + y[1]) + (y[2] + y[3]);
                    csq[jt] += fmaf(y[0], y[0], fmaf(y[1], y[1], fmaf(y[2], y[2], y[3] * y[3])));
                }
        };
        issue_res(0, std::integral_constant<int, 0>{});
        __syncthreads();                                 // barrier #0
        __syncthreads();                                 // steps -1, 0 transformed
        for (int it0 = 0; it0 < NSTEP; it0 += 2) {
            issue_res(it0 + 1, std::integral_constant<int, 1>{});
            compute(it0, std::integral_constant<int, 0>{});
            __syncthreads();
            issue_res(it0 + 2, std::integral_constant<int, 0>{});
            compute(it0 + 1, std::integral_constant<int, 1>{});
            __syncthreads();
        }
        // per-stripe partial statistics (one store per channel and wave: the real kernel reduces the waves through LDS first)
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            csum[jt] += __shfl_xor(csum[jt], 8); csq[jt] += __shfl_xor(csq[jt], 8);
            csum[jt] += __shfl_xor(csum[jt], 16); csq[jt] += __shfl_xor(csq[jt], 16);
            csum[jt] += __shfl_xor(csum[jt], 32); csq[jt] += __shfl_xor(csq[jt], 32);
            if (lane < 8 && csum[jt] == 123.456f) out_[lane] = csq[jt];
        }
    }
}

// reference point: the plainest possible copy of the same bytes (float4 per work-item, grid-stride)
__global__ __launch_bounds__(256) void copy_k(const float4* __restrict__ a, const float4* __restrict__ r, float4* __restrict__ o, size_t n, int res) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = a[i];
        if (res) { const float4 w = r[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        o[i] = v;
    }
}

template <int W, int KO, int NJ, int RS, int NB, bool RES, int XF, bool MF>
void run(int B, const float* in, const float* res, float* out, const double* stats, const uint4* wf) {
    const int H = W;
    constexpr int NT = (((2 * (W / 4) * KO + 63) / 64) + (W / 16 >= 4 ? 4 : W / 16)) * 64;
    auto kern = stripe_k<W, KO, NJ, RS, NB, RES, XF, MF>;
    dim3 grid(B * (H / RS));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(NT), 0, 0, in, res, out, stats, wf, H);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(NT), 0, 0, in, res, out, stats, wf, H);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mb = (double)B * H * W * 4 * (8 * KO + (RES ? 2 : 1) * 8 * NJ) / 1e6;
    int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NT, 0);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
    printf("stripe %d->%d @%dx%d B=%d RS=%d NB=%d res=%d xf=%d mfma=%d: %7.1f us  (%6.1f MB -> %.2f TB/s)  grid %d x %d thr, %d WG/CU, %d VGPR, %zu B LDS, %zu B scratch\n", 8 * KO, 8 * NJ, W, W, B, RS, NB,
           (int)RES, XF, (int)MF, ms / reps * 1e3, mb, mb / (ms / reps * 1e-3) * 1e-6, grid.x, NT, nb, fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    const int B = 64;
    const size_t n = (size_t)B * 8 * 256 * 256;          // = B x 16 ch x 128^2 x 2 = B x 32 ch x 64^2 x 4: every shape below fits
    float *in, *res, *out; double* stats; uint4* wf;
    hipMalloc(&in, n * 4); hipMalloc(&res, n * 4); hipMalloc(&out, n * 4);
    hipMalloc(&stats, (size_t)B * 64 * 8 * 2 * sizeof(double)); hipMalloc(&wf, 4 * 3 * 2 * 128 * sizeof(uint4));
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(res, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(stats, 0, (size_t)B * 64 * 8 * 2 * sizeof(double));
    hipMemset(wf, 0x3c, 4 * 3 * 2 * 128 * sizeof(uint4));
    // plain copies of the same byte counts
    for (int r = 0; r < 2; ++r) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(copy_k, dim3(256 * 16), dim3(256), 0, 0, (const float4*)in, (const float4*)res, (float4*)out, n / 4, r);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(copy_k, dim3(256 * 16), dim3(256), 0, 0, (const float4*)in, (const float4*)res, (float4*)out, n / 4, r);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mb = (double)n * 4 * (r ? 3 : 2) / 1e6;
        printf("plain copy res=%d: %7.1f us  (%6.1f MB -> %.2f TB/s)\n", r, ms / 20 * 1e3, mb, mb / (ms / 20 * 1e-3) * 1e-6);
    }
    // 256^2, B = 64, 8 -> 8 (the final-resolution layers of the SR U-Net); conv_rp_kernel today: 97.7 us with / 68.3 us without the identity residual
    run<256, 1, 1, 32, 2, true, 1, true>(B, in, res, out, stats, wf);
    run<256, 1, 1, 32, 2, false, 1, true>(B, in, res, out, stats, wf);
    run<256, 1, 1, 32, 4, true, 1, true>(B, in, res, out, stats, wf);
    run<256, 1, 1, 64, 2, true, 1, true>(B, in, res, out, stats, wf);
    run<256, 1, 1, 16, 2, true, 1, true>(B, in, res, out, stats, wf);
    run<256, 1, 1, 32, 2, true, 0, true>(B, in, res, out, stats, wf);       // no transform arithmetic
    run<256, 1, 1, 32, 2, true, 1, false>(B, in, res, out, stats, wf);      // no MFMA
    run<256, 1, 1, 32, 2, true, 0, false>(B, in, res, out, stats, wf);      // pure movement through the pipeline
    run<256, 1, 1, 32, 2, false, 0, false>(B, in, res, out, stats, wf);
    // 128^2, B = 64: 8 -> 8 (24.1 us today with the identity residual) and 16 -> 8 (37-42 us today)
    run<128, 1, 1, 32, 2, true, 1, true>(B, in, res, out, stats, wf);
    run<128, 1, 1, 16, 2, true, 1, true>(B, in, res, out, stats, wf);
    run<128, 1, 1, 16, 4, true, 1, true>(B, in, res, out, stats, wf);
    run<128, 1, 1, 16, 2, false, 1, true>(B, in, res, out, stats, wf);
    run<128, 1, 1, 8, 2, true, 1, true>(B, in, res, out, stats, wf);
    run<128, 1, 1, 8, 4, true, 1, true>(B, in, res, out, stats, wf);
    run<128, 1, 1, 16, 2, true, 0, false>(B, in, res, out, stats, wf);
    run<128, 2, 1, 16, 2, false, 1, true>(B, in, res, out, stats, wf);
    run<128, 2, 1, 8, 2, false, 1, true>(B, in, res, out, stats, wf);
    // 64^2, B = 64: 16 -> 16 (18.6-20.8 us today), 32 -> 16 (26 us today)
    run<64, 2, 2, 8, 2, true, 1, true>(B, in, res, out, stats, wf);
    run<64, 2, 2, 8, 4, true, 1, true>(B, in, res, out, stats, wf);
    run<64, 2, 2, 8, 2, false, 1, true>(B, in, res, out, stats, wf);
    run<64, 2, 2, 4, 2, true, 1, true>(B, in, res, out, stats, wf);
    run<64, 2, 2, 4, 2, false, 1, true>(B, in, res, out, stats, wf);
    run<64, 2, 2, 16, 2, true, 1, true>(B, in, res, out, stats, wf);
    run<64, 2, 2, 8, 2, true, 0, false>(B, in, res, out, stats, wf);
    run<64, 4, 2, 8, 2, false, 1, true>(B, in, res, out, stats, wf);
    run<64, 4, 2, 4, 2, false, 1, true>(B, in, res, out, stats, wf);
    // 64^2, 8 -> 8 (the base U-Net's first level: 10-13 us today) and 32^2 16 -> 16 (its second level)
    run<64, 1, 1, 8, 2, true, 1, true>(B, in, res, out, stats, wf);
    run<64, 1, 1, 4, 2, true, 1, true>(B, in, res, out, stats, wf);
    run<32, 2, 2, 4, 2, true, 1, true>(B, in, res, out, stats, wf);
    run<32, 2, 2, 8, 2, true, 1, true>(B, in, res, out, stats, wf);
    return 0;
}
