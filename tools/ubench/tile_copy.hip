// Micro-benchmark (dev tool): which global-memory access pattern of a conv-shaped tile kernel reaches streaming bandwidth?
// B x 8ch x 256 x 256 fp32 in -> out, tile 8x64 per 256-thread workgroup, NT tiles per workgroup, depth-1 register prefetch.
//   LOADP 0: one pixel per work-item, 8 channels as dword loads, 1-pixel halo (10 x 66 units)   (conv_rp.hip)
//   LOADP 1: float4 per work-item per channel, aligned window with 4-pixel halo (10 x 18 float4 x 8 ch)
//   LOADP 2: no halo, float4, 8 x 16 float4 x 8ch (pure copy)
//   STOREP 0: MFMA-layout store: lane (lq, lg) -> channel lq&7, row parity lq>>3, 4 px at 16*gx + 4*lg (64-byte fragments)
//   STOREP 1: row-major float4: 16 lanes cover 256 B of one row of one channel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int LOADP, int STOREP>
__global__ __launch_bounds__(256, 4) void k(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int ntile, int ilv) {
    constexpr int TH = 8, TW = 64;
    __shared__ float lds[10 * 72 * 8];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lg = lane >> 4;
    const int tiles_x = W / TW, tiles = tiles_x * (H / TH), strips = tiles / ntile;
    const int L = blockIdx.x, kk = L >> 3;
    const int b = (L & 7) + 8 * (kk / strips), strip = kk % strips;
    const int HW = H * W;
    const float* ib = in + (size_t)b * 8 * HW;
    float* ob = out + (size_t)b * 8 * HW;
    constexpr int PER = LOADP == 0 ? 3 : (LOADP == 1 ? 6 : 4);
    float4 r4[LOADP == 0 ? 1 : PER];
    float r1[LOADP == 0 ? PER * 8 : 1];
    auto load = [&](int tile) {
        const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
        if (LOADP == 0) {
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int q = tid + u * 256, iy = q / 66, c = q - iy * 66;
                const int gy = oy0 - 1 + iy, gx = ox0 - 1 + c;
                const bool ok = q < 660 && gy >= 0 && gy < H && gx >= 0 && gx < W;
                const unsigned off = ok ? gy * W + gx : 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) r1[u * 8 + j] = ib[(size_t)j * HW + off];
            }
        } else if (LOADP == 1) {
#pragma unroll
            for (int u = 0; u < PER; ++u) {       // units: 8 ch x 10 rows x 18 float4 = 1440
                const int q = tid + u * 256, xg = q % 18, r = q / 18, iy = r % 10, ch = r / 10;
                const int gy = oy0 - 1 + iy, gx = ox0 - 4 + 4 * xg;
                const bool ok = q < 1440 && gy >= 0 && gy < H && gx >= 0 && gx < W;
                const unsigned off = ok ? (ch * HW + gy * W + gx) : 0;
                r4[u] = *reinterpret_cast<const float4*>(ib + off);
            }
        } else {
#pragma unroll
            for (int u = 0; u < PER; ++u) {       // 8 ch x 8 rows x 16 float4 = 1024
                const int q = tid + u * 256, xg = q % 16, r = q / 16, iy = r % 8, ch = r / 8;
                r4[u] = *reinterpret_cast<const float4*>(ib + ch * HW + (oy0 + iy) * W + ox0 + 4 * xg);
            }
        }
    };
    // ilv 1: workgroup s walks tiles s, s + strips, s + 2 strips, ... instead of a contiguous strip
    auto tile_of = [&](int t) { return ilv ? strip + t * strips : strip * ntile + t; };
    load(tile_of(0));
    for (int t = 0; t < ntile; ++t) {
        const int tile = tile_of(t);
        const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
        __syncthreads();
        // stage through LDS as [ch][row 10][col 72] floats
        if (LOADP == 0) {
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int q = tid + u * 256, iy = q / 66, c = q - iy * 66;
                if (q < 660)
#pragma unroll
                    for (int j = 0; j < 8; ++j) lds[(j * 10 + iy) * 72 + c + 3] = r1[u * 8 + j];
            }
        } else if (LOADP == 1) {
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int q = tid + u * 256, xg = q % 18, r = q / 18;
                if (q < 1440) *reinterpret_cast<float4*>(&lds[r * 72 + 4 * xg]) = r4[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int q = tid + u * 256, xg = q % 16, r = q / 16, iy = r % 8, ch = r / 8;
                *reinterpret_cast<float4*>(&lds[(ch * 10 + iy + 1) * 72 + 4 + 4 * xg]) = r4[u];
            }
        }
        __syncthreads();
        if (t + 1 < ntile) load(tile_of(t + 1));
        if (STOREP == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int G = wave * 4 + g, gyy = G / 4, gxx = G % 4;
                const int co = lq & 7, dy = lq >> 3;
                const int oy = 2 * gyy + dy, ox = 16 * gxx + 4 * lg;
                const float4 v = *reinterpret_cast<const float4*>(&lds[(co * 10 + oy + 1) * 72 + 4 + ox]);
                *reinterpret_cast<float4*>(ob + (size_t)co * HW + (oy0 + oy) * W + ox0 + ox) = v;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = tid + u * 256, xg = q % 16, r = q / 16, iy = r % 8, ch = r / 8;
                const float4 v = *reinterpret_cast<const float4*>(&lds[(ch * 10 + iy + 1) * 72 + 4 + 4 * xg]);
                *reinterpret_cast<float4*>(ob + (size_t)ch * HW + (oy0 + iy) * W + ox0 + 4 * xg) = v;
            }
        }
    }
}

template <int LOADP, int STOREP>
void run(int ntile, int ilv = 0) {
    const int B = 64, H = 256, W = 256;
    const size_t n = (size_t)B * 8 * H * W;
    float *in, *out;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 4);
    hipMemset(in, 0, n * 4);
    const int tiles = (H / 8) * (W / 64);
    dim3 grid(B * tiles / ntile);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(HIP_KERNEL_NAME(k<LOADP, STOREP>), grid, dim3(256), 0, 0, in, out, B, H, W, ntile, ilv);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(HIP_KERNEL_NAME(k<LOADP, STOREP>), grid, dim3(256), 0, 0, in, out, B, H, W, ntile, ilv);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("load pattern %d, store pattern %d, %d tiles/WG%s: %.1f us  (%.2f TB/s of 268 MB)\n", LOADP, STOREP, ntile, ilv ? " interleaved" : "", ms / 20 * 1e3, 2.0 * n * 4 / (ms / 20 * 1e-3) * 1e-12);
    hipFree(in); hipFree(out);
}
int main() {
    for (int nt : {1, 4, 8}) {
        run<0, 0>(nt); run<0, 1>(nt); run<1, 0>(nt); run<1, 1>(nt); run<2, 1>(nt); run<2, 0>(nt);
    }
    for (int nt : {4, 8}) { run<0, 0>(nt, 1); run<2, 1>(nt, 1); }
    return 0;
}
