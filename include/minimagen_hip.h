/* minimagen_hip.h -- C ABI of libminimagen_hip.so (MI355X / gfx950).
 *
 * The reference (AssemblyAI-Community/MinImagen) has no FFI: its hot path is a chain of
 * ATen ops behind Python classes.  Each entry point below replaces the group of reference
 * ops named in its comment (file:line relative to the reference tree); the Python host
 * (minimagen_amd/ Python modules) keeps the reference's class API and calls these through ctypes with
 * raw device pointers.  No torch types cross this boundary.
 *
 * Conventions
 *   - all tensors fp32, contiguous; activations NCHW
 *   - every function only ENQUEUES work on `stream` (a hipStream_t passed as void*);
 *     no allocation, no host synchronisation -> safe inside HIP-graph capture
 *   - return 0 on success, a negative mi_status otherwise; mi_last_error() gives the
 *     message of the calling thread's last failure
 *   - "stats" buffers hold per-channel partial sums for the NEXT GroupNorm:
 *       float stats[B][C][nt][2] = (sum, sum of squares) over one producer tile,
 *     written by the producing kernel's epilogue, reduced in a fixed order by the
 *     consumer (deterministic; no float atomics)
 */
#ifndef MINIMAGEN_HIP_H
#define MINIMAGEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_ABI_VERSION 1

enum mi_status {
    MI_OK = 0,
    MI_ERR_INVALID = -1,      /* bad argument / unsupported shape */
    MI_ERR_LAUNCH = -2,       /* HIP launch error */
    MI_ERR_UNSUPPORTED = -3
};

int mi_abi_version(void);
const char* mi_last_error(void);
/* "hip-gfx950" for the product library.  (The CPU SIMT emulator used by the dev tests reports "hipemu".) */
const char* mi_backend(void);
/* sizeof() of the parameter structs, for binding self-checks: 0 mi_act, 1 mi_conv_params, 2 mi_crossembed_params, ... */
int mi_struct_size(int which);

/* ---- activation view ------------------------------------------------------------------ */
typedef struct mi_act {
    const float* data;   /* [B][C][H][W] */
    int C;
    const float* stats;  /* [B][C][nt][2] partial (sum, sumsq); may be NULL when no GroupNorm consumes it */
    int nt;
    float scale;         /* multiplies the data when consumed (skip connections: 2^-1/2, Unet.py:445) */
} mi_act;

/* ---- K4/K6/K7/K8/K11: the conv family --------------------------------------------------
 * out = conv_k(act(concat(in0, in1*scale))) + bias [+ residual]
 *   act  = GroupNorm(groups) -> [x*(scale+1)+shift] -> SiLU   when gn_groups>0  (Block, layers.py:131-145)
 *   conv = k3 s1 p1 | k4 s2 p1 (Downsample, layers.py:319) | nearest x2 then k3 (Upsample, layers.py:512-515)
 *   residual = identity add of res0, or 1x1 conv (res_w) of concat(res0,res1*scale)  (ResnetBlock, layers.py:415,439)
 * Weights are pre-packed by the host: w[Cin][k][k][Cout_pad], Cout_pad = Cout rounded up to cout_tile.
 * The epilogue also emits the per-channel partial stats of `out` when out_stats != NULL.
 */
typedef struct mi_conv_params {
    int B, H, W;            /* OUTPUT height/width */
    mi_act in0, in1;        /* in1.data == NULL -> single input */
    int Cout;
    int ksize, stride, up2;
    const float* w;
    const float* bias;      /* [Cout] or NULL */
    int gn_groups;          /* 0 = no norm/activation on the input */
    const float* gn_gamma;  /* [Cin] */
    const float* gn_beta;   /* [Cin] */
    float gn_eps;
    const float* scale_shift; /* [B][ss_stride]; scale at ss_off+c, shift at ss_off+Cin+c; NULL = none */
    int ss_stride, ss_off;
    mi_act res0, res1;      /* res0.data == NULL -> no residual */
    const float* res_w;     /* [Cres][Cout_pad] 1x1 weights, NULL = identity */
    const float* res_b;     /* [Cout] or NULL */
    float* out;             /* [B][Cout][H][W] */
    float* out_stats;       /* [B][Cout][out_nt][2] or NULL */
    int tile_cfg;           /* see mi_conv_tile_shape */
} mi_conv_params;

/* tile_cfg -> output tile (th x tw) handled by one workgroup; out_nt = ceil(H/th)*ceil(W/tw) */
int mi_conv_tile_shape(int tile_cfg, int* th, int* tw);
int mi_conv_cout_tile(int Cout);               /* channel tile (4, 8 or 16) the kernels use for this Cout */
int mi_conv_fwd(const mi_conv_params* p, void* stream);

/* ---- K3: CrossEmbedLayer (layers.py:298-305): parallel k=3,7,15 convs, concat on channels --- */
typedef struct mi_crossembed_params {
    int B, H, W;
    const float* in0; int C0;      /* x */
    const float* in1; int C1;      /* lowres_cond_img (Unet.py:396-397) or NULL */
    int in1_batch_mod;             /* in1 is indexed by (b % in1_batch_mod); 0 = B */
    int in0_batch_mod;
    int n_kernels;                 /* <= 3 */
    int ksize[3];                  /* sorted ascending, odd */
    int cout[3];                   /* channels per kernel (dim_scales) */
    const float* w[3];             /* packed [Cin][k][k][cout_i] */
    const float* bias[3];
    float* out; float* out_stats;  /* [B][sum cout][H][W], [B][C][nt][2] */
    int tile_cfg;
} mi_crossembed_params;
int mi_crossembed_fwd(const mi_crossembed_params* p, void* stream);

#ifdef __cplusplus
}
#endif
#endif
