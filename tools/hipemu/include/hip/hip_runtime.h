// hipemu -- a small CPU SIMT emulator for the HIP kernel sources of this repo.
//
// DEVELOPMENT / TEST TOOL ONLY.  It is NOT a product path and NOT a CPU fallback: the
// shipped library (libminimagen_hip.so) is built by hipcc for gfx950 from the same
// .hip sources and never contains or loads any of this.  The build container has no
// GPU, so kernel *logic* (indexing, LDS staging, wave shuffles, MFMA fragment layouts)
// is exercised here before spending GPU minutes.  This header shadows
// <hip/hip_runtime.h> when the sources are compiled as plain C++ for x86 with
//     clang++ -x c++ -I tools/hipemu/include ...
//
// Model: one workgroup = a set of fibers (one per work-item) on one OS thread;
// __syncthreads() / wave collectives are cooperative yields.  Wave = 64 lanes.
// MFMA builtins follow the fragment layouts documented in the CDNA4 guide
// (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D: col=l&15,row=4*(l>>4)+reg for 16x16x4;
//  A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D: col=l&31,row=(r&3)+8*(r>>2)+4*(l>>5) for 32x32x2).
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hipemu_uint3 { unsigned x, y, z; };

struct alignas(8) float2 { float x, y; };
struct alignas(16) double2 { double x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline double2 make_double2(double x, double y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorUnknown = 999 };
struct hipemu_stream;
typedef hipemu_stream* hipStream_t;
struct hipemu_graph;
typedef hipemu_graph* hipGraph_t;
typedef hipemu_graph* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };

namespace hipemu {

enum FiberState { F_READY = 0, F_WAIT_BLOCK = 1, F_WAIT_WAVE = 2, F_DONE = 3 };

struct Fiber {
    void* sp;
    char* stack;
    int state;
    hipemu_uint3 tid;
    int lane, wave;
};

struct Wave {
    alignas(16) uint32_t buf[64][8];
    bool part[64];
};

struct BlockCtx {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    void* sched_sp;
    const std::function<void()>* body;
    hipemu_uint3 bid, bdim, gdim;
    char* stack_pool;
    size_t stack_pool_sz;
};

extern thread_local BlockCtx* g_blk;
extern thread_local Fiber* g_cur;

void yield_to_scheduler(int new_state);
void launch(dim3 grid, dim3 block, hipStream_t stream, std::function<void()> body);
void enqueue(hipStream_t stream, std::function<void()> op);   // honours stream capture

inline Wave& cur_wave() { return g_blk->waves[g_cur->wave]; }
inline void wave_sync() { yield_to_scheduler(F_WAIT_WAVE); }

template <class T> inline T wave_read(T v, int src_lane) {
    static_assert(sizeof(T) <= 16, "wave_read payload too large");
    Wave& w = cur_wave();
    std::memcpy(w.buf[g_cur->lane], &v, sizeof(T));
    wave_sync();
    T r;
    std::memcpy(&r, w.buf[src_lane & 63], sizeof(T));
    wave_sync();
    return r;
}
}  // namespace hipemu

#define threadIdx (hipemu::g_cur->tid)
#define blockIdx (hipemu::g_blk->bid)
#define blockDim (hipemu::g_blk->bdim)
#define gridDim (hipemu::g_blk->gdim)
#define warpSize 64

static inline void __syncthreads() { hipemu::yield_to_scheduler(hipemu::F_WAIT_BLOCK); }

template <class T> static inline T __shfl(T v, int src, int width = 64) {
    int lane = hipemu::g_cur->lane;
    int base = lane & ~(width - 1);
    return hipemu::wave_read(v, base + (src & (width - 1)));
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    int lane = hipemu::g_cur->lane;
    int src = lane ^ mask;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu::wave_read(v, src);
}
template <class T> static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    int lane = hipemu::g_cur->lane;
    int src = lane + (int)delta;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu::wave_read(v, src);
}
template <class T> static inline T __shfl_up(T v, unsigned delta, int width = 64) {
    int lane = hipemu::g_cur->lane;
    int src = lane - (int)delta;
    if (src < 0 || (src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu::wave_read(v, src);
}
static inline unsigned long long __ballot(int pred) {
    hipemu::Wave& w = hipemu::cur_wave();
    int lane = hipemu::g_cur->lane;
    w.buf[lane][0] = pred ? 1u : 0u;
    w.part[lane] = true;
    hipemu::wave_sync();
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (w.part[l] && w.buf[l][0]) m |= (1ull << l);
    hipemu::wave_sync();
    w.part[lane] = false;
    return m;
}
static inline int __all(int pred) {
    hipemu::Wave& w = hipemu::cur_wave();
    int lane = hipemu::g_cur->lane;
    w.buf[lane][0] = pred ? 1u : 0u;
    w.part[lane] = true;
    hipemu::wave_sync();
    int r = 1;
    for (int l = 0; l < 64; ++l)
        if (w.part[l] && !w.buf[l][0]) r = 0;
    hipemu::wave_sync();
    w.part[lane] = false;
    return r;
}
static inline int __any(int pred) { return __ballot(pred) != 0ull; }

// ---- builtins used by the kernels ------------------------------------------------------
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));

static inline hipemu_f32x4 hipemu_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    hipemu::Wave& w = hipemu::cur_wave();
    int l = hipemu::g_cur->lane;
    std::memcpy(&w.buf[l][0], &a, 4);
    std::memcpy(&w.buf[l][1], &b, 4);
    hipemu::wave_sync();
    hipemu_f32x4 d = c;
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            std::memcpy(&av, &w.buf[row + 16 * k][0], 4);   // A[i=row][k] lives in lane i + 16k
            std::memcpy(&bv, &w.buf[col + 16 * k][1], 4);   // B[k][j=col] lives in lane j + 16k
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    hipemu::wave_sync();
    return d;
}
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
    hipemu::Wave& w = hipemu::cur_wave();
    int l = hipemu::g_cur->lane;
    std::memcpy(&w.buf[l][0], &a, 4);
    std::memcpy(&w.buf[l][1], &b, 4);
    hipemu::wave_sync();
    hipemu_f32x16 d = c;
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            std::memcpy(&av, &w.buf[row + 32 * k][0], 4);
            std::memcpy(&bv, &w.buf[col + 32 * k][1], 4);
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    hipemu::wave_sync();
    return d;
}
typedef _Float16 hipemu_f16x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x16_f16: A[i = l&15][k = 4*(l>>4) + e], B[k = 4*(l>>4) + e][j = l&15], D as the other 16x16 shapes
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x16f16(hipemu_f16x4 a, hipemu_f16x4 b, hipemu_f32x4 c, int, int, int) {
    hipemu::Wave& w = hipemu::cur_wave();
    int l = hipemu::g_cur->lane;
    std::memcpy(&w.buf[l][0], &a, 8);
    std::memcpy(&w.buf[l][2], &b, 8);
    hipemu::wave_sync();
    hipemu_f32x4 d = c;
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            hipemu_f16x4 av, bv;
            std::memcpy(&av, &w.buf[row + 16 * (k >> 2)][0], 8);
            std::memcpy(&bv, &w.buf[col + 16 * (k >> 2)][2], 8);
            acc += (float)av[k & 3] * (float)bv[k & 3];      // exact products of halves, fp32 accumulation
        }
        d[r] = acc;
    }
    hipemu::wave_sync();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x16f16 hipemu_mfma_f32_16x16x16f16
typedef _Float16 hipemu_f16x8 __attribute__((ext_vector_type(8)));
// v_mfma_f32_16x16x32_f16 (gfx950): A[i = l&15][k = 8*(l>>4) + e], B[k = 8*(l>>4) + e][j = l&15], e = 0..7
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x32_f16(hipemu_f16x8 a, hipemu_f16x8 b, hipemu_f32x4 c, int, int, int) {
    hipemu::Wave& w = hipemu::cur_wave();
    int l = hipemu::g_cur->lane;
    std::memcpy(&w.buf[l][0], &a, 16);
    std::memcpy(&w.buf[l][4], &b, 16);
    hipemu::wave_sync();
    hipemu_f32x4 d = c;
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            hipemu_f16x8 av, bv;
            std::memcpy(&av, &w.buf[row + 16 * (k >> 3)][0], 16);
            std::memcpy(&bv, &w.buf[col + 16 * (k >> 3)][4], 16);
            acc += (float)av[k & 7] * (float)bv[k & 7];
        }
        d[r] = acc;
    }
    hipemu::wave_sync();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 hipemu_mfma_f32_16x16x32_f16
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_f32_16x16x4f32
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu_mfma_f32_32x32x2f32
typedef __fp16 hipemu_h2 __attribute__((ext_vector_type(2)));
static inline _Float16 hipemu_f16_rtz(float f) {           // fp32 -> fp16, round toward zero (v_cvt_pkrtz_f16_f32)
    _Float16 h = (_Float16)f;                             // RNE first, then step back if it rounded away from zero
    if (std::isfinite(f) && std::fabs((float)h) > std::fabs(f)) {
        unsigned short b; std::memcpy(&b, &h, 2); b -= 1; std::memcpy(&h, &b, 2);
    }
    return h;
}
static inline hipemu_h2 hipemu_cvt_pkrtz(float a, float b) { hipemu_h2 r; r[0] = (__fp16)hipemu_f16_rtz(a); r[1] = (__fp16)hipemu_f16_rtz(b); return r; }
#define __builtin_amdgcn_cvt_pkrtz(a, b) hipemu_cvt_pkrtz((a), (b))
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
template <class T> static inline T hipemu_readfirstlane(T v) {
    // first ACTIVE lane: emulate with the lowest participating lane
    hipemu::Wave& w = hipemu::cur_wave();
    int lane = hipemu::g_cur->lane;
    std::memcpy(w.buf[lane], &v, sizeof(T));
    w.part[lane] = true;
    hipemu::wave_sync();
    int first = lane;
    for (int l = 0; l < 64; ++l)
        if (w.part[l]) { first = l; break; }
    T r;
    std::memcpy(&r, w.buf[first], sizeof(T));
    hipemu::wave_sync();
    w.part[lane] = false;
    return r;
}
#define __builtin_amdgcn_readfirstlane(x) hipemu_readfirstlane(x)

static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
#define __expf(x) expf(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }

template <class T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline float atomicAdd(float* p, float v) {
    unsigned* up = reinterpret_cast<unsigned*>(p);
    unsigned old = __atomic_load_n(up, __ATOMIC_RELAXED), nw;
    do { float f; std::memcpy(&f, &old, 4); f += v; std::memcpy(&nw, &f, 4); }
    while (!__atomic_compare_exchange_n(up, &old, nw, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    float r; std::memcpy(&r, &old, 4); return r;
}
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ---- host API ---------------------------------------------------------------------------
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
const char* hipGetErrorString(hipError_t);
hipError_t hipMemsetAsync(void* p, int value, size_t bytes, hipStream_t s);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode mode);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* graph);
hipError_t hipGraphInstantiate(hipGraphExec_t* exec, hipGraph_t graph, void*, void*, size_t);
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t s);
hipError_t hipGraphExecDestroy(hipGraphExec_t exec);
hipError_t hipGraphDestroy(hipGraph_t graph);

#define HIP_KERNEL_NAME(...) __VA_ARGS__
namespace hipemu {
template <class K, class... A> inline void launch_k(dim3 g, dim3 b, hipStream_t s, K kernel, A... args) {
    launch(g, b, s, [=]() { kernel(args...); });
}
}  // namespace hipemu
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch_k((grid), (block), (stream), kernel, __VA_ARGS__)
