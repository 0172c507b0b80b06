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

template <int W, int RS, int NB, bool RES, int XF, bool MF>
__global__ __launch_bounds__((W / 128 + 4) * 64) void stripe_k(const float* __restrict__ in_, const float* __restrict__ res_, float* __restrict__ out_,
                                                              const float4* __restrict__ coef, const uint4* __restrict__ wfrag, int H) {
    constexpr int NLW = W / 128, NCW = 4, PW = W + 8, RING = 6, NG = W / 16, GPW = NG / NCW;
    constexpr int NSTEP = RS / 2;
    static_assert(NSTEP % NB == 0 && NSTEP % 2 == 0, "steps per stripe must be a multiple of the unroll factors");
    __shared__ __attribute__((aligned(16))) uint4 actH[RING * PW];
    __shared__ __attribute__((aligned(16))) uint4 actL[RING * PW];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int stripes = H / RS;
    const int b = blockIdx.x / stripes, y0 = (blockIdx.x % stripes) * RS;
    const size_t HW = (size_t)H * W;
    const gptr<const float> in = (gptr<const float>)(in_ + (size_t)b * 8 * HW);
    const gptr<const float> res = (gptr<const float>)(res_ + (size_t)b * 8 * HW);
    const gptr<float> out = (gptr<float>)(out_ + (size_t)b * 8 * HW);
    // zero the left / right pad chunk of every ring row once
    if (tid < RING * 2) {
        const int r = tid >> 1, c = (tid & 1) ? W + 1 : 0;
        actH[r * PW + c] = make_uint4(0, 0, 0, 0);
        actL[r * PW + c] = make_uint4(0, 0, 0, 0);
    }
    if (wave < NLW) {
        // ------------------------------------------------ loader / transform waves
        constexpr int QPR = W / 4;                       // quads per row
        const int u = wave * 64 + lane, lrow = u / QPR, q = u % QPR;        // which of the step's 2 new rows, which pixel quad
        float4 P[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) P[j] = coef[b * 8 + j];
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
            for (int j = 0; j < 8; ++j) raw[buf][j] = *reinterpret_cast<gptr<const f32x4>>(in + (size_t)j * HW + off);
        };
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
                actH[slot * PW + 1 + 4 * q + px] = hv;
                actL[slot * PW + 1 + 4 * q + px] = lv;
            }
        };
        // prologue: steps -1 .. NB - 1 in flight, steps -1 and 0 transformed before the first compute step
        issue(-1, std::integral_constant<int, NB - 1>{});
        [&]<int... I>(std::integer_sequence<int, I...>) { (issue(I, std::integral_constant<int, I % NB>{}), ...); }(std::make_integer_sequence<int, NB - 1>{});
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
        f16x8 bh[3], bl[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            bh[s] = __builtin_bit_cast(f16x8, wfrag[(s * 2) * 64 + lane]);
            bl[s] = __builtin_bit_cast(f16x8, wfrag[(s * 2 + 1) * 64 + lane]);
        }
        const int co = lq & 7, dy = lq >> 3;
        f32x4 rv[2][GPW];
        auto issue_res = [&](int it, auto buf_tag) {
            constexpr int buf = decltype(buf_tag)::value;
            if (!RES) return;
#pragma unroll
            for (int g = 0; g < GPW; ++g) {
                const int oy = y0 + 2 * (it < NSTEP ? it : NSTEP - 1) + dy, ox = 16 * (cw * GPW + g) + 4 * lg;
                rv[buf][g] = *reinterpret_cast<gptr<const f32x4>>(res + (size_t)co * HW + (unsigned)(oy * W + ox));
            }
        };
        float csum = 0.f, csq = 0.f;
        auto compute = [&](int it, auto buf_tag) {
            constexpr int buf = decltype(buf_tag)::value;
            f32x4 acc[GPW];
#pragma unroll
            for (int g = 0; g < GPW; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int slot = (2 * it + perm) % RING;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
#pragma unroll
                for (int g = 0; g < GPW; ++g) {
                    const int idx = slot * PW + 16 * (cw * GPW + g) + lq + s;
                    const f16x8 ah = __builtin_bit_cast(f16x8, actH[idx]);
                    const f16x8 al = __builtin_bit_cast(f16x8, actL[idx]);
                    if (MF) {
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[s], acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[s], acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[s], acc[g], 0, 0, 0);
                    } else {
                        acc[g][0] += (float)ah[0] + (float)al[0]; acc[g][1] += (float)ah[1]; acc[g][2] += (float)ah[2]; acc[g][3] += (float)al[3];
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < GPW; ++g) {
                const int oy = y0 + 2 * it + dy, ox = 16 * (cw * GPW + g) + 4 * lg;
                f32x4 y = acc[g] * 0.25f + 0.5f;
                if (RES) y += rv[buf][g];
                *reinterpret_cast<gptr<f32x4>>(out + (size_t)co * HW + (unsigned)(oy * W + ox)) = y;
                csum += (y[0] + y[1]) + (y[2] + y[3]);
                csq += fmaf(y[0], y[0], fmaf(y[1], y[1], fmaf(y[2], y[2], y[3] * y[3])));
            }
        };
        issue_res(0, std::integral_constant<int, 0>{});
        __syncthreads();
        for (int it0 = 0; it0 < NSTEP; it0 += 2) {
            issue_res(it0 + 1, std::integral_constant<int, 1>{});
            compute(it0, std::integral_constant<int, 0>{});
            __syncthreads();
            issue_res(it0 + 2, std::integral_constant<int, 0>{});
            compute(it0 + 1, std::integral_constant<int, 1>{});
            __syncthreads();
        }
        if (csum == 123.456f && csq == 1.0f) out[0] = csum;      // keep the statistics arithmetic alive
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

template <int W, int RS, int NB, bool RES, int XF, bool MF>
void run(int B, const float* in, const float* res, float* out, const float4* coef, const uint4* wf) {
    const int H = W;
    constexpr int NT = (W / 128 + 4) * 64;
    dim3 grid(B * (H / RS));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(HIP_KERNEL_NAME(stripe_k<W, RS, NB, RES, XF, MF>), grid, dim3(NT), 0, 0, in, res, out, coef, wf, H);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(HIP_KERNEL_NAME(stripe_k<W, RS, NB, RES, XF, MF>), grid, dim3(NT), 0, 0, in, res, out, coef, wf, H);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mb = (double)B * 8 * H * W * 4 * (RES ? 3 : 2) / 1e6;
    int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, HIP_KERNEL_NAME(stripe_k<W, RS, NB, RES, XF, MF>), NT, 0);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&stripe_k<W, RS, NB, RES, XF, MF>));
    printf("stripe W=%d B=%d RS=%d NB=%d res=%d xf=%d mfma=%d: %7.1f us  (%6.1f MB -> %.2f TB/s)  grid %d x %d thr, %d WG/CU, %d VGPR, %zu B LDS, %zu B scratch\n", W, B, RS, NB,
           (int)RES, XF, (int)MF, ms / reps * 1e3, mb, mb / (ms / reps * 1e-3) * 1e-6, grid.x, NT, nb, fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    const int B = 64;
    const size_t n = (size_t)B * 8 * 256 * 256;
    float *in, *res, *out; float4* coef; uint4* wf;
    hipMalloc(&in, n * 4); hipMalloc(&res, n * 4); hipMalloc(&out, n * 4);
    hipMalloc(&coef, B * 8 * sizeof(float4)); hipMalloc(&wf, 6 * 64 * sizeof(uint4));
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(res, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float4> hc(B * 8, make_float4(1.5f, 0.1f, -2.1f, -0.14f));
    hipMemcpy(coef, hc.data(), hc.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipMemset(wf, 0x3c, 6 * 64 * sizeof(uint4));
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
    // 256^2, B = 64 (the final-resolution layers of the SR U-Net)
    run<256, 32, 2, true, 1, true>(B, in, res, out, coef, wf);
    run<256, 32, 2, false, 1, true>(B, in, res, out, coef, wf);
    run<256, 32, 4, true, 1, true>(B, in, res, out, coef, wf);
    run<256, 64, 2, true, 1, true>(B, in, res, out, coef, wf);
    run<256, 16, 2, true, 1, true>(B, in, res, out, coef, wf);
    run<256, 32, 2, true, 0, true>(B, in, res, out, coef, wf);       // no transform arithmetic
    run<256, 32, 2, true, 1, false>(B, in, res, out, coef, wf);      // no MFMA
    run<256, 32, 2, true, 0, false>(B, in, res, out, coef, wf);      // pure movement through the pipeline
    run<256, 32, 2, false, 0, false>(B, in, res, out, coef, wf);
    // 128^2, B = 64
    run<128, 32, 2, true, 1, true>(B, in, res, out, coef, wf);
    run<128, 16, 2, true, 1, true>(B, in, res, out, coef, wf);
    run<128, 16, 4, true, 1, true>(B, in, res, out, coef, wf);
    run<128, 16, 2, false, 1, true>(B, in, res, out, coef, wf);
    run<128, 8, 2, true, 1, true>(B, in, res, out, coef, wf);
    run<128, 16, 2, true, 0, false>(B, in, res, out, coef, wf);
    return 0;
}
