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
 *       double stats[B][C][nt][2] = (sum, sum of squares) over one producer tile.  fp64 since ABI 8: every producer accumulates its
 *       tile about a local shift (a value of the tile) in fp32 and converts to the plain sums in fp64, so that var = E[x^2] - mean^2
 *       survives activations whose mean is orders of magnitude above their deviation (a large DC offset, a constant image),
 *     written by the producing kernel's epilogue, reduced in a fixed order by the
 *     consumer (deterministic; no float atomics)
 */
#ifndef MINIMAGEN_HIP_H
#define MINIMAGEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_ABI_VERSION 12

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
    const double* stats; /* [B][C][nt][2] partial (sum, sumsq), fp64; may be NULL when no GroupNorm consumes it */
    int nt;
    float scale;         /* multiplies the data when consumed (skip connections: 2^-1/2, Unet.py:445) */
    int bmod;            /* >0: the tensor has only `bmod` batch rows, row b%bmod is used (tensors shared by the
                            conditional and null halves of a classifier-free-guidance batch) */
    int st;              /* storage: 0 = fp32; 1 = bf16 (2 bytes per element, `data` then points at bf16 values) -- the
                            reduced-precision configuration only (BASELINE configs 3-5), honoured by the single-term kernels
                            (tile_cfg | 0x400, attention variant 7); statistics stay fp32 */
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
    int out_st;             /* storage of `out` (as mi_act.st; single-term kernels only) */
    double* out_stats;      /* [B][Cout][out_nt][2] or NULL */
    int tile_cfg;           /* see mi_conv_tile_shape; | MI_CONV_SPLIT16: 16-channel outputs as two 8-channel workgroups */
    /* row-paired matrix-core path (conv_rp.hip; tile_cfg 5..7): B-operand fragments of v_mfma_f32_16x16x32_f16 with
       N = (output row parity dy, 8 output channels), K = (4 input rows, 8 input channels) per horizontal tap:
       [ceil(Cin/8)][steps][ceil(Cout/8)][64 lanes][8 hi | 8 lo halves] of w * 2^w_rp_exp (power-of-two pre-scaling keeps the
       fp16 hi/lo split in the normal range whatever the magnitude of the weights); res_w_rp likewise for the 1x1 residual */
    const void* w_rp;
    const void* res_w_rp;
    int w_rp_exp, res_w_rp_exp;
    /* wide-channel regime (more than 64 input or 32 output channels: Unet() default, Base, Super): the per-channel affine of the fused
       GroupNorm / scale-shift and the power-of-two operand exponents are computed per image by mi_gn_coef_fwd(p) into these buffers
       BEFORE mi_conv_fwd(p); non-NULL gn_coef selects the wide kernel (output channels tiled over the grid) */
    float* gn_coef;         /* [B][Cin][4] = {A 2^ka, B 2^ka, -log2(e) A, -log2(e) B} */
    int* gn_exps;           /* [B][2] = {ka, largest safe exponent of the 1x1-residual input} */
    /* wide GEMM kernel (conv_wide.hip; tile_cfg 11, k3 s1, input / residual channels in multiples of 32, output channels of 64): the conv input after
       GroupNorm / scale-shift / SiLU (and the 1x1-residual input) as fp16 hi / lo operand planes, written once per layer by mi_conv_prep_fwd(p) --
       after mi_gn_coef_fwd(p), before mi_conv_fwd(p); mi_conv_prep_bytes() bytes; w_rp / res_w_rp then hold packing.pack_conv_weight_ig fragments
       ([Cin/32][taps][Cout/16][hi | lo][64 lanes][8 halves]: lane (lq, lg) = W[16 nt + lq][32 g + 8 lg .. + 7][tap]) */
    void* act_prep; long long act_prep_bytes;
} mi_conv_params;
#define MI_CONV_SPLIT16 0x100
#define MI_CONV_SPLIT8  0x800   /* 8-channel outputs as two 4-channel workgroups (small, latency-bound launches) */
#define MI_CONV_HALF    0x400   /* row-paired matrix-core path: single fp16 term per product (reduced-precision configuration; parity gate 3e-2) */
#define MI_CONV_REVERSE 0x200  /* tile_cfg | MI_CONV_REVERSE (row-paired path): workgroups take the images in reverse order (speed only) */
#define MI_CONV_RP_FIRST 6      /* tile_cfg 6: 8x64, 7: 8x32 output tiles of the row-paired matrix-core path (5, 16x64, removed in ABI 10); 10: 16x16 (wide k3 s1 convs on images <= 16 wide); 11: 8x16 tiles of the wide GEMM kernel (conv_wide.hip) */

/* tile_cfg -> output tile (th x tw) handled by one workgroup; out_nt = ceil(H/th)*ceil(W/tw) */
int mi_conv_tile_shape(int tile_cfg, int* th, int* tw);
/* tile_cfg 12 (round 6, conv_stripe.hip): the narrow k3 s1 layers on images 32 / 64 / 128 / 256 pixels wide -- and the k4 s2 Downsample conv over 8
   channels to 64 / 128 output columns -- as full-width row stripes with specialised loader and MFMA waves (same arithmetic and bits as tile_cfg 6 / 7;
   fp32 storage only).  Returns the rows per statistics block (W / 8 of the OUTPUT: out_nt = H / rows) when mi_conv_fwd(p) with tile_cfg 12 takes the launch described by p (every field but
   tile_cfg / out / out_stats is looked at), else 0.  tile_cfg bits 12..15 = statistics blocks per workgroup (0 = the library's choice; speed only). */
int mi_conv_stripe_rows(const mi_conv_params* p);
int mi_conv_prep_fwd(const mi_conv_params* p, void* stream);      /* operand planes of the wide GEMM kernel (tile_cfg 11) */
long long mi_conv_prep_bytes(int B, int Cin, int Cres, int H, int W);
int mi_conv_cout_tile(int Cout);               /* channel tile (4, 8 or 16) the kernels use for this Cout */
int mi_conv_fwd(const mi_conv_params* p, void* stream);
int mi_gn_coef_fwd(const mi_conv_params* p, void* stream);     /* fills p->gn_coef / p->gn_exps (wide-channel regime) */

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
    float* out; double* out_stats;  /* [B][sum cout][H][W], [B][C][nt][2] */
    int out_st;                    /* storage of `out` AND of `addend` (as mi_act.st; matrix-core kernel with tile_cfg | 0x400 only) */
    int tile_cfg;
    const float* addend;           /* [B][sum cout][H][W] added to the result (the step-invariant low-res half of the
                                      convolution, computed once per sample()), or NULL; bias[] entries may be NULL */
    /* non-NULL: the matrix-core kernel (dim_scales (4,2,2), kernel sizes (3,7,15), in1 == NULL, C0 <= 4, W % 4 == 0, tile_cfg 8 =
       32x64 or 9 = 16x32 tiles, | 0x400 for single-term fp16): the Toeplitz weight table of packing.pack_crossembed_mfma,
       [32 rows][256] fp16, and the power-of-two exponent each conv's weights were pre-scaled by.  w[] is then unused. */
    const void* w_mfma;
    int w_mfma_exp[3];
} mi_crossembed_params;
int mi_crossembed_fwd(const mi_crossembed_params* p, void* stream);

/* ---- conditioning --------------------------------------------------------------------- */
typedef struct mi_linear {   /* torch nn.Linear layout: w[out][in], b[out] (b may be NULL) */
    const float* w; const float* b; int in, out;
} mi_linear;

/* K2: text conditioning, once per sample() (Unet._text_condition, Unet.py:571-634, minus the
 * time-token rows): text_to_cond -> truncate/pad to max_len -> mask / CFG-drop to null_text_embed
 * -> mean-pool -> to_text_non_attn_cond (LayerNorm, Linear, SiLU, Linear) -> null_text_hidden
 * select; and norm_cond (LayerNorm, per row) of the text rows of c.  Row b' of the outputs uses
 * text row b' % B and keep[b'] (the all-True / all-False prob_mask_like of Unet.py:587). */
typedef struct mi_text_cond_params {
    int B2, B, L, E, cd, tcd, max_len;
    const float* text_embeds;       /* [B][L][E] */
    const uint8_t* text_mask;       /* [B][L] or NULL (all valid) */
    const uint8_t* keep;            /* [B2] */
    mi_linear text_to_cond;
    const float* null_text_embed;   /* [max_len][cd] */
    const float* ln_w; const float* ln_b;   /* to_text_non_attn_cond.0 */
    mi_linear h1, h2;               /* to_text_non_attn_cond.1 / .3 */
    const float* null_text_hidden;  /* [tcd] */
    const float* norm_w; const float* norm_b;  /* norm_cond */
    float* c_text;                  /* out [B2][max_len][cd] */
    float* text_hiddens;            /* out [B2][tcd] */
} mi_text_cond_params;
int mi_text_cond_fwd(const mi_text_cond_params* p, void* stream);

/* K1 + K5, every denoising step: sinusoidal embedding -> to_time_hiddens -> {to_time_cond,
 * to_time_tokens} (+ the lowres trio, Unet.py:508-536), t += text_hiddens (Unet.py:626),
 * norm_cond of the time-token rows (Unet.py:629-632), and every ResnetBlock's
 * time_mlp = Linear(SiLU(t)) (layers.py:395-399,425-429) in one launch. */
typedef struct mi_cond_step_params {
    int B2, B, dim, cd, tcd, ntok;
    const int64_t* time;            /* [B] */
    const int64_t* lowres_time;     /* [B] or NULL */
    const float* freq;              /* [dim/2] = exp(arange(dim/2) * -(ln 1e4/(dim/2-1)))  (layers.py:461-463) */
    mi_linear th, tc, tt;           /* to_time_hiddens.1, to_time_cond.0, to_time_tokens.0 */
    mi_linear lth, ltc, ltt;        /* to_lowres_time_* (w == NULL when not lowres_cond) */
    const float* text_hiddens;      /* [B2][tcd] or NULL */
    const float* norm_w; const float* norm_b;
    mi_linear time_mlps;            /* all ResnetBlock time_mlp.1 stacked: [R][tcd] */
    float* ss;                      /* out [B2][R] */
    float* c_time;                  /* out [B2][ntok_total][cd], ntok_total = ntok * (1 + lowres) */
    float* t_out;                   /* out [B2][tcd] or NULL */
    float* silu_out;                /* out [B2][tcd] or NULL: SiLU(t), the input of every time_mlp -- for callers that run the stacked time-MLP as one GEMM
                                       (ss == NULL skips it here: the wide presets, where it is [~10 K][512] per row) */
} mi_cond_step_params;
int mi_cond_step_fwd(const mi_cond_step_params* p, void* stream);

/* ---- K9: bottleneck cross-attention, folded form ---------------------------------------
 * CrossAttention (layers.py:220-251) inside ResnetBlock (layers.py:433-435) when
 * C (=dim_out) and cond_dim are below dim_head=64: with x^ = LN(x),
 *     sim_h = x^ . (scale * Wq_h^T Wk_h) . c^T           (C x cond_dim matrix per head)
 *     out   = sum_h softmax(sim_h) . (c . Wv_h^T Wo_h^T)
 * so the 512-wide q/k/v/out tensors never exist.  mi_attn_fold_rows turns context rows into
 * MFMA A-operand fragments ("gv"); mi_cross_attn_fwd does LN -> QK^T -> softmax -> PV ->
 * to_out.1 LayerNorm -> + residual with v_mfma_f32_16x16x4_f32 (exact fp32).
 *   gv layout: [B2][heads][JT][64 lanes][FR], FR = max(4, C/4) + 4*ceil(C/16); context row j
 *   lives in tile j/16.  Row 0 is the null key/value, rows 1.. the time tokens, then 256 text rows. */
#define MI_ATTN_MAX_BLOCKS 8
typedef struct mi_attn_fold_params {
    int B2, C, cd, heads, JT;
    const float* c_rows;            /* [B2][nrows][cd] rows (already norm_cond-ed) */
    int c_stride_b;                 /* floats between consecutive b' in c_rows */
    int row0, nrows;                /* they become context rows row0 .. row0+nrows-1 */
    int write_null;                 /* also write context row 0 from g0 / v0 */
    int frag_f16;                   /* 1: fragments for the fp16x3 kernel (variant 6): every value split hi + lo into two halves,
                                       per lane [G hi x4][G lo x4][VW hi x4][VW lo x4] per 16-channel chunk (same bytes as fp32) */
    int n_blocks;
    struct {
        const float* mg;            /* [heads][C][cd]  log2(e) * scale * Wq_h^T Wk_h */
        const float* mv;            /* [heads][C][cd]  Wo_h Wv_h */
        const float* g0;            /* [heads][C]      log2(e) * scale * Wq_h^T null_k */
        const float* v0;            /* [heads][C]      Wo_h null_v */
        float* gv;
        float* table;               /* mode 1 / 2: compact folded rows [steps * B2][nrows][heads][C][2] = (g, vw) */
        int g_exp, v_exp;           /* frag_f16: the fragments hold g * 2^g_exp and vw * 2^v_exp (exact power-of-two scalings chosen by
                                       the host from the weights' magnitude so that the fp16 hi/lo split stays in the normal range) */
    } blk[MI_ATTN_MAX_BLOCKS];
    /* The timestep sequence of a sampling loop is known in advance (T-1 .. 0, the same for every sample), so everything that depends
       only on (timestep, text) is computed ONCE per sample() for all T steps and the per-step work shrinks to a scatter:
         mode 0: fold c_rows and write the fragments (one step; Unet.forward)
         mode 1: fold c_rows of B2 = steps * rows virtual batch rows into blk[].table (no fragments written)
         mode 2: scatter the rows of step *t_state from blk[].table into the fragments of the B2 real batch rows, and copy that
                 step's ss_n scale/shift values per row from ss_all to ss (the ResnetBlocks' time_mlp outputs)
         mode 3: as mode 2 with a table of ONE row per timestep shared by all batch rows (the time tokens depend on the timestep only; built by a
                 mode-1 call over T pseudo-rows); the scale/shift table ss_all stays per (timestep, batch row) */
    int mode;
    const int* t_state;
    int t_off;              /* modes 2 / 3: the step is *t_state - t_off */
    const float* ss_all; float* ss; int ss_n;
} mi_attn_fold_params;
int mi_attn_fold_rows(const mi_attn_fold_params* p, void* stream);
int mi_attn_fragment_floats(int C);     /* FR */

typedef struct mi_cross_attn_params {
    int B2, C, HW, heads, J;        /* J = 1 + time tokens + 256 */
    mi_act x;                       /* block1 output h [B2][C][HW]; also the residual */
    const float* gv;
    const float* n1_g; const float* n1_b;   /* CrossAttention.norm   gamma / beta */
    const float* n2_g; const float* n2_b;   /* to_out.1              gamma / beta */
    float* out; double* out_stats;   /* [B2][C][HW]; stats [B2][C][ceil(HW/tok)][2], tok = 64 tokens */
    int out_st;                     /* storage of `out` (as mi_act.st; variant 7 only) */
    int x_exp, g_exp, v_exp;        /* variants 6 / 7: LayerNorm(x) is scaled by 2^x_exp before its fp16 split, the fragments carry 2^g_exp /
                                       2^v_exp (mi_attn_fold_params); the kernel undoes all three exactly (scores, output) */
    int variant;                    /* 6: 16 tokens per wave with the contractions as 3-term fp16 splits on v_mfma_f32_16x16x32_f16
                                       (hi*hi + hi*lo + lo*hi, ~2^-21: fp32-grade); 7: as 6 with a single fp16 term (reduced-precision
                                       configuration, same frag_f16 fragments).  (0 .. 5, the exact-fp32 MFMA yardsticks of rounds 1-5, are gone: ABI 10) */
} mi_cross_attn_params;
int mi_cross_attn_fwd(const mi_cross_attn_params* p, void* stream);
#define MI_ATTN_TOKENS_PER_WG 128

/* ---- sampler (Imagen._p_mean_variance / _p_sample, Imagen.py:261-370) ------------------
 * Per-stage device state so that one HIP graph can be replayed for every timestep:
 *   int   t_state[1]   current timestep t (T-1 .. 0)
 *   float coef[T][8]   { sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, posterior_mean_coef1,
 *                        posterior_mean_coef2, [t != 0] * exp(0.5 * posterior_log_variance_clipped), 0, 0, 0 }
 *                      (tables built by the host in fp64 then cast, exactly as diffusion_model.py:28-66)
 */
/* K11 epilogue: classifier-free-guidance combine (Unet.py:506) + predict_start_from_noise
 * (diffusion_model.py:159-162).  pred2 holds the conditional rows [0,B) and, when cond_scale != 1,
 * the null rows [B,2B). */
typedef struct mi_cfg_x0_params {
    int B, n;                     /* n = C*H*W elements per image */
    const float* pred2;
    int two;                      /* 1: pred2 has 2B rows and the combine runs */
    float cond_scale;
    const float* x_t;             /* [B][n]; may be NULL when only pred_out is wanted */
    const float* coef; const int* t_state;
    float* pred_out;              /* [B][n] guided prediction or NULL */
    float* x0;                    /* [B][n] or NULL */
    unsigned* hist0;              /* optional: pass-0 slice [B][2][MI_Q_BINS] of the quantile histograms (zero on entry): the radix
                                     select's first pass over |x0| is accumulated here while x0 is produced (see mi_quantile_params.pass0_done) */
    int t_off;                    /* the step is *t_state - t_off: several steps of one captured graph share one advance of t_state */
} mi_cfg_x0_params;
int mi_cfg_x0_fwd(const mi_cfg_x0_params* p, void* stream);

/* K12: dynamic-threshold quantile (Imagen.py:313-317 -> torch.quantile, linear interpolation):
 * exact selection of the order statistics k_lo and k_hi of |x0| per image by a 3-pass radix
 * select over the fp32 bit patterns (11+11+9 bits), integer atomics only -> deterministic and
 * bit-exact.  The rank arithmetic (k_lo, w in fp32) is torch's and is done by the host. */
#define MI_Q_BINS 2048
typedef struct mi_quantile_params {
    int B, n;
    const float* x0;              /* [B][n] */
    int k_lo, k_hi; float w;
    unsigned* hist;               /* [3][B][2][MI_Q_BINS], zeroed by mi_quantile_fwd itself unless self_cleaning */
    float* s_out;                 /* [B]  lerp(v_lo, v_hi, w) -- NOT yet clamped to >= 1 */
    float* v_out;                 /* [B][2] the two selected order statistics, or NULL */
    int pass0_done;               /* 1: mi_cfg_x0_fwd already accumulated pass 0 into hist (its hist0 field); skips that pass */
    int self_cleaning;            /* 1: hist is zero on entry (caller's guarantee) and is left zeroed on exit: no memset launch */
} mi_quantile_params;
int mi_quantile_fwd(const mi_quantile_params* p, void* stream);

/* K13: threshold, posterior mean and the reverse-diffusion draw (Imagen.py:320-326,361-370):
 *   s = max(1, s_q); x0 = clamp(x0,-s,s)/s; mean = c1*x0 + c2*x_t; x_{t-1} = mean + sigma_t * eps
 * eps comes from `noise` (parity runs: host-generated, step k = T-1-t at noise + k*B*n) or, when
 * noise == NULL, from Philox4x32-10 + Box-Muller keyed by (seed, global sample index, stream, element). */
typedef struct mi_posterior_params {
    int B, n, T;
    const float* x0; const float* s_q;
    float* x;                     /* in: x_t, out: x_{t-1} */
    const float* coef; const int* t_state;
    const float* noise;
    uint64_t seed; int sample0; int stream_base;
    const uint64_t* seed_dev;     /* when non-NULL the seed is read from device memory (lets one captured graph serve every call) */
    int t_off;                    /* the step is *t_state - t_off */
} mi_posterior_params;
int mi_posterior_fwd(const mi_posterior_params* p, void* stream);

/* K11 epilogue + K12 + K13 of one denoising step in ONE launch, for images that fit a workgroup's registers (n <= MI_SAMPLER_SMALL_N, the
 * base stage): guidance combine, x0, the exact radix select of the two order statistics (histograms in LDS, no global atomics), the
 * threshold and the posterior draw.  Same operations in the same order as mi_cfg_x0_fwd -> mi_quantile_fwd -> mi_posterior_fwd (bit-identical
 * results); c->x0 / c->pred_out / q->s_out / q->v_out are written when non-NULL, c->hist0 / q->hist are not used.  MI_ERR_UNSUPPORTED for
 * larger images. */
#define MI_SAMPLER_SMALL_N 16384
int mi_sampler_step_small_fwd(const mi_cfg_x0_params* c, const mi_quantile_params* q, const mi_posterior_params* pp, void* stream);

/* The same fused tail for LARGE images (n % 4 == 0): mi_sampler_group_size(n) workgroups of 1024 work-items per image keep their share of
 * x0 in registers across the three radix passes; the passes' histograms are combined with integer agent-scope atomics and a counter barrier
 * among the workgroups of an image (claimed by ticket once resident: no cooperative launch; every spin bounded).
 * Replaces mi_cfg_x0_fwd + mi_quantile_fwd (4 launches) + mi_posterior_fwd; bit-identical results.  `sync`: mi_sampler_group_sync_bytes(B, n)
 * bytes, zero-filled once, private to one stream's launches.  c->x0 / c->pred_out / q->s_out / q->v_out are written when non-NULL; c->hist0 /
 * q->hist / pp->x0 / pp->s_q are not used.
 * Header of `sync`: [0] u64 ticket | [8] u32 error word (sticky) | [12] u32 knobs: bits 0..30 spin limit (0 = the built-in 2^22), bit 31
 * fault injection (tests: the last workgroup of image 0 skips its arrival at radix pass 1).
 * FAIL-STOP: the workgroups of an image wait for each other.  One launch cannot deadlock (work is claimed by ticket once resident), but
 * launches in flight on different streams can starve each other when their waiting workgroups together fill the chip (observed with 128
 * workgroups per image at 1024^2).  A workgroup whose wait runs out stores 0x300 + pass into the error word, overwrites ITS part of pp->x with
 * NaN and leaves; every workgroup that finds the word set (peers at their next poll, every later launch on this buffer at its start) does
 * the same without waiting: a failed step is never a stale or half-written image, it is NaN from there on.  The caller polls the word at a
 * synchronisation point (the Python host copies it to pinned memory behind every call and raises at the next API entry), zero-fills `sync`
 * again and should stay with the separate kernels afterwards.  Keep mi_sampler_group_size(n) small against the number of CUs -- the Python
 * host uses this entry point for at most 8 workgroups per image (256^2). */
int mi_sampler_group_size(int n);                          /* workgroups per image; 0: unsupported */
long long mi_sampler_group_sync_bytes(int B, int n);
int mi_sampler_step_group_fwd(const mi_cfg_x0_params* c, const mi_quantile_params* q, const mi_posterior_params* pp, void* sync, void* stream);

/* t -= 1 ; times[b] = t  (diffusion_model.py:81-87, one step of the list) */
int mi_step_advance(int* t_state, int64_t* times, int B, void* stream);
/* t -= n ; times[b] = t  (the steps of one captured graph address *t_state - k and advance once) */
int mi_step_advance_by(int* t_state, int64_t* times, int B, int n, void* stream);
/* t = value ; times[b] = value */
int mi_step_set(int* t_state, int64_t* times, int B, int value, void* stream);

/* N(0,1) fill with the same generator as mi_posterior_fwd (x_T, low-res augmentation noise) */
int mi_randn_fill(float* out, int B, int n, uint64_t seed, int sample0, int stream_id, void* stream);

/* img.clamp_(-1,1); then (img+1)*0.5 when unnormalize != 0   (Imagen.py:418-420) */
int mi_finalize_images(const float* x, float* out, int64_t total, int unnormalize, void* stream);

/* K14: separable cubic resize with host-built tap tables (helpers.py:138-164 -> resize_right), H pass
 * then W pass; idx tables already contain the reflect-padded source indices. */
typedef struct mi_resize_params {
    int planes, Hin, Win, Hout, Wout, KH, KW;
    const float* in; float* out;
    const int* idx_h; const float* w_h;   /* [Hout][KH] */
    const int* idx_w; const float* w_w;   /* [Wout][KW] */
} mi_resize_params;
int mi_resize_fwd(const mi_resize_params* p, void* stream);

/* low-res conditioning augmentation (Imagen.py:483-485 q_sample, then the *2-1 of Imagen.py:393):
 * out = (a*img + b*noise)*2 - 1 */
int mi_lowres_augment(const float* img, const float* noise, float* out, int64_t total, float a, float b, int normalize, void* stream);

/* ---- K10: self-attention TransformerBlock for narrow layers (C in {8,16,32}) -------------
 * Attention (layers.py:52-104, multi-query: one shared 64-wide k/v head, null k/v) in the same folded form as K9 with
 * the image tokens as their own context: sim_h = x^ (s Wq_h^T Wk) x^^T, out = sum_h softmax(sim_h) (x^ Wv^T Wo_h^T).
 * Context length = H*W + 1, so the score row is processed in chunks with an online softmax (running max / sum). */
/* x^ = LayerNorm_C(x) written token-major: out[B][HW][C]  (Attention.norm, layers.py:57) */
int mi_ln_tokens_fwd(const mi_act* x, int B, int HW, const float* gamma, const float* beta, float* out, void* stream);
typedef struct mi_self_attn_params {
    int B2, C, HW, heads, J;        /* J = HW + 1 context rows (null k/v first) */
    mi_act x;                       /* [B2][C][HW]; also the residual (TransformerBlock :497 / Residual :368) */
    const float* gv;                /* fragments from mi_attn_fold_rows with JT = ceil(J/16) */
    const float* n1_g; const float* n1_b;   /* Attention.norm */
    const float* n2_g; const float* n2_b;   /* to_out.1 */
    float* out; double* out_stats;   /* stats [B2][C][ceil(HW/64)][2] or NULL */
} mi_self_attn_params;
int mi_self_attn_fwd(const mi_self_attn_params* p, void* stream);
/* ChanFeedForward + residual (layers.py:148-161, 498): y = x + W2 . CLN(gelu(W1 . CLN(x)))  (1x1 convs, no bias) */
typedef struct mi_chan_ff_params {
    int B, C, Chid, HW;
    mi_act x;
    const float* g1; const float* w1;   /* ChanLayerNorm g [C]; conv [Chid][C] */
    const float* g2; const float* w2;   /* ChanLayerNorm g [Chid]; conv [C][Chid] */
    float* out; double* out_stats;       /* stats [B][C][ceil(HW/256)][2] or NULL */
} mi_chan_ff_params;
int mi_chan_ff_fwd(const mi_chan_ff_params* p, void* stream);

/* ---- wide-channel attention (C > 32: Unet() default, Base, Super): the reference's own factorisation as token-major building blocks,
 * exact fp32 on the matrix cores: mi_ln_tokens_fwd -> mi_gemm_f32 (to_q / to_kv) -> mi_flash_attn_fwd -> mi_gemm_f32 (to_out.0)
 * -> mi_tokens_to_nchw_fwd (to_out.1 LayerNorm + residual + NCHW + statistics); ChanFeedForward: mi_ln_tokens_fwd -> mi_gemm_f32
 * (GELU) -> mi_ln_rows_fwd -> mi_gemm_f32 -> mi_tokens_to_nchw_fwd. */
typedef struct mi_flash_attn_params {
    int B, HW, heads, kv_heads;     /* dim_head 64; kv_heads = heads (CrossAttention, layers.py:226-233) or 1 (multi-query Attention, :42) */
    const float* q; float q_scale;  /* [B][HW][heads*64]; q_scale = dim_head^-0.5 * log2(e) */
    const float* null_k; const float* null_v;   /* [64] each: the null key / value prepended to every head's context, or NULL */
    /* up to two context segments after the null row (time tokens | text tokens; or the image tokens themselves):
       row r of batch b at k + b * bs + r * ld (+ 64 * head when kv_heads > 1) */
    const float* k0; const float* v0; int n0, ld0; long long bs0;
    const float* k1; const float* v1; int n1, ld1; long long bs1;
    float* out;                     /* [B][HW][heads*64] */
    /* optional workspace of mi_flash_kv_prep_bytes(B * kv_heads, J) bytes (J = context rows incl. the null row); with it the K / V operands are prepared
       once per launch (a small kernel) and streamed global -> LDS by LDS-DMA instead of every workgroup re-staging the whole context (multi-query
       form with heads % 4 == 0, or one k / v head per head); NULL: the self-staging kernels */
    void* kv_prep; long long kv_prep_bytes;
} mi_flash_attn_params;
int mi_flash_attn_fwd(const mi_flash_attn_params* p, void* stream);
long long mi_flash_kv_prep_bytes(int B, int J);
typedef struct mi_tokens_to_nchw_params {
    int B, HW, C;
    const float* tokens;            /* [B][HW][C] */
    const float* gamma; const float* beta; float eps;   /* LayerNorm over C first (gamma NULL: none; beta may be NULL) */
    mi_act res;                     /* NCHW residual added (data NULL: none) */
    float* out; double* out_stats;   /* [B][C][HW]; stats [B][C][ceil(HW/64)][2] or NULL */
} mi_tokens_to_nchw_params;
int mi_tokens_to_nchw_fwd(const mi_tokens_to_nchw_params* p, void* stream);
/* y[row] = LayerNorm(x[row]) * gamma (+ beta, may be NULL) over the last dimension of [rows][dim] */
int mi_ln_rows_fwd(const float* x, const float* gamma, const float* beta, float* y, int rows, int dim, float eps, void* stream);

/* ---- K16: T5 encoder (minimagen/t5.py:71-84 -> transformers T5Stack, third party) -------
 * fp32; dense contractions on v_mfma_f32_16x16x4_f32.  All matrices in torch nn.Linear layout. */
/* C[M][N] = act(A[M][K] . W[N][K]^T) + R[M][N]   (R may be NULL; act: 0 none, 1 ReLU, 2 gelu_new (tanh form), 3 exact GELU (erf form, nn.GELU))
 * gate != NULL (gated-gelu FF of T5 v1.1): C = act(A.W^T) * (A.gate^T).   K % 16 == 0 (N % 64 == 0 for the gated form). */
int mi_gemm_f32(const float* A, const float* W, const float* gate, const float* R, float* Cout, int M, int N, int K, int act, void* stream);
/* The same product with T5's RMSNorm folded in (round 6: 44 -> 32 launches per encode): rowsq_in [M][nparts] (may be NULL) holds partial sums of
 * squares of A's rows; row m of A.W^T is scaled by rsqrt(sum_p rowsq_in[m][p] / K + eps) before the activation -- with the norm's weight folded into W
 * by the caller this is act(RMSNorm(A) . W0^T).  rowsq_out [M][ceil(N/64)] (may be NULL) receives, per 64-column tile, the sum of squares of every
 * output row after the residual: the next projection's rowsq_in.  K % 32 == 0. */
int mi_gemm_rms_f32(const float* A, const float* W, const float* gate, const float* R, float* Cout, int M, int N, int K, int act,
                    const float* rowsq_in, int nparts, float eps, float* rowsq_out, void* stream);
/* y = x * rsqrt(mean(x^2) + eps) * w per row (T5LayerNorm); zero_mask != NULL: rows with zero_mask[row]==0 are zeroed (t5.py:82) */
int mi_rmsnorm(const float* x, const float* w, float* y, int rows, int dim, float eps, const uint8_t* zero_mask, void* stream);
/* out[row][:] = table[ids[row]][:] */
int mi_embed_rows(const int64_t* ids, const float* table, float* out, int rows, int dim, void* stream);
/* ... and rowsq[row][0] = sum of squares of the row, rowsq[row][1 .. nparts-1] = 0 (the record mi_gemm_rms_f32 reads) */
int mi_embed_rows_sq(const int64_t* ids, const float* table, float* out, float* rowsq, int nparts, int rows, int dim, void* stream);
/* T5 self-attention core for one layer: qkv [B*L][3*inner] (q | k | v, head h at columns h*64), unscaled q.k^T
 * + bias_tab[heads][2L-1] (relative position bias indexed by (j - i) + L-1) + key mask, softmax, .v -> ctx [B*L][inner] */
int mi_t5_attention(const float* qkv, const float* bias_tab, const uint8_t* key_mask, float* ctx, int B, int L, int heads, void* stream);

/* ---- training path: parameter gradients of the 3x3 stride-1 convolutions -------
 * The backward of Block.project (layers.py:126,145; Imagen.forward -> loss.backward(), Imagen.py:512-573).  The data gradient of a
 * 3x3 stride-1 conv is mi_conv_fwd itself on the transposed, flipped kernel; this entry computes
 *   dw[co][ci][ky][kx] = sum_{b,y,x} dy[b][co][y][x] * a[b][ci][y+ky-1][x+kx-1]   and   db[co] = sum dy[b][co][y][x]
 * (a = the conv's input, i.e. the activated tensor) as a split-K matrix-core GEMM; partial sums go through `partial`
 * (mi_conv_wgrad_workspace(Cin, Cout, nwg) floats) and are added in a fixed order: deterministic. */
typedef struct mi_conv_wgrad_params {
    int B, Cin, Cout, H, W;
    const float* a;                 /* [B][Cin][H][W] */
    const float* dy;                /* [B][Cout][H][W] */
    float* dw;                      /* [Cout][Cin][3][3] */
    float* db;                      /* [Cout] or NULL */
    float* partial; int nwg;        /* workspace and the number of workgroups walking the pixel tiles (e.g. 512) */
    /* a_stats != NULL: `a` is the RAW input of a Block and the kernel applies SiLU(GroupNorm(a) * (scale + 1) + shift) while it stages
       the tiles (the activated tensor is never materialised): statistics [B][Cin][a_nt][2], affine, scale|shift table as in mi_conv_params */
    const double* a_stats; int a_nt;
    const float* gamma; const float* beta; int groups; float eps;
    const float* ss; int ss_stride, ss_off;
} mi_conv_wgrad_params;
long long mi_conv_wgrad_workspace(int Cin, int Cout, int nwg);
int mi_conv_wgrad(const mi_conv_wgrad_params* p, void* stream);

/* the pointwise half of Block's backward, fused (the normalised / activated tensors are recomputed in registers): with
 * y1 = gamma * xhat + beta, y2 = y1 * (scale + 1) + shift, a = silu(y2) and da = dL/da (the data gradient of the conv):
 * dx (GroupNorm backward included), dgamma, dbeta, d(scale | shift).  Three launches: row sums, apply, parameter gradients. */
typedef struct mi_block_bwd_params {
    int B, C, HW, groups, nt, nchunk;   /* nchunk: row chunks per (image, channel) (grid.y of the two streaming kernels) */
    float eps;
    const float* x; const float* da;    /* [B][C][HW] */
    const double* x_stats;              /* [B][C][nt][2] */
    const float* gamma; const float* beta;
    const float* ss; int ss_stride, ss_off;   /* [B][ss_stride]: scale at ss_off + c, shift at ss_off + C + c; or NULL */
    float* uv;                          /* workspace [B][C][nchunk][2] */
    float* dx;                          /* [B][C][HW] */
    float* dgamma; float* dbeta;        /* [C] */
    float* dss;                         /* [B][2C] (d scale | d shift), required when ss != NULL */
    double* dx_stats;                   /* [B][C][nchunk][2] (sum, sum of squares) of dx per row chunk, or NULL */
} mi_block_bwd_params;
int mi_block_bwd(const mi_block_bwd_params* p, void* stream);
/* per-(image, channel) sum and sum of squares of x [rows][HW] -> stats [rows][2] (a tensor no HIP producer left statistics for) */
int mi_chan_stats_fwd(const float* x, double* stats, int rows, int HW, void* stream);
/* LayerNorm over the last dimension in the training graph (ABI 11) -- the reference's `LayerNorm` (layers.py:333-343: the token norms around the
 * bottleneck cross-attention, 131 072 rows of 16 channels at B = 32) and the nn.LayerNorm members of the conditioning stack (Unet.py:107-112,
 * layers.py:137):  y = (x - mean) rstd gamma + beta (beta may be NULL), biased variance, dim <= 1024; stat [rows][2] = (mean, rstd) is what the
 * backward reads (may be NULL when no backward follows).
 * Backward: dx = rstd (g - mean(g) - xh mean(g xh)) with g = dy gamma, xh = (x - mean) rstd; dgamma = sum_rows dy xh, dbeta = sum_rows dy through
 * `partial` ([mi_layernorm_bwd_nwg(rows, dim)][2][dim] floats), added in workgroup order by a second launch: deterministic.  dgamma == NULL: dx only. */
int mi_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stat, int rows, int dim, float eps, void* stream);
int mi_layernorm_bwd_nwg(int rows, int dim);
int mi_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* stat, float* dx, float* partial, float* dgamma, float* dbeta,
                     int rows, int dim, void* stream);
/* a [Cout][Cin][3][3] weight (adjoint != 0: its adjoint W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx]) times 2^exp -> the row-paired fp16 hi|lo
 * fragments of mi_conv_params.w_rp and the direct-conv layout [Cin'][3][3][cout_pad] of mi_conv_params.w, in one launch on the device
 * (the training path re-packs after every optimiser step).  mi_pack_conv3_floats(…, which): element counts (0: fp16 fragments, 1: fp32). */
long long mi_pack_conv3_floats(int Cout, int Cin, int adjoint, int cout_pad, int which);
int mi_pack_conv3(const float* w, int Cout, int Cin, int adjoint, int exp, void* frag, float* generic, int cout_pad, void* stream);
/* the same for `n` weights in ONE launch (ABI 12): `descs` is a DEVICE array of n descriptors (the training step re-packs every conv weight of a
 * U-Net in both directions after every optimiser step -- ~150 launches of a few microseconds each on a step that is bound by its launch count);
 * `blocks` workgroups per descriptor. */
typedef struct mi_pack_conv3_desc {
    const float* w; void* frag; float* generic;
    int Cout, Cin, adjoint, exp, cout_pad, reserved;
} mi_pack_conv3_desc;
int mi_pack_conv3_multi(const mi_pack_conv3_desc* descs, int n, int blocks, void* stream);

/* weight / bias gradients of CrossEmbedLayer (layers.py:254-305; the first layer: no data gradient is needed): one correlation over the
 * largest member's taps serves every member (a smaller member's gradient is the centre window of its channels), split-K on the fp32
 * matrix-core instruction, partials added in a fixed order.  Kernel sizes odd and <= 15, <= 16 output channels in all, Cin * kmax <= 96. */
typedef struct mi_crossembed_wgrad_params {
    int B, Cin, H, W;
    const float* x;                 /* [B][Cin][H][W]: the layer's input (image | low-res conditioning image) */
    const float* dy;                /* [B][sum cout][H][W] */
    int n_kernels; int ksize[3]; int cout[3];
    float* dw[3];                   /* [cout_i][Cin][k_i][k_i] */
    float* db[3];                   /* [cout_i] or NULL */
    float* partial; int nwg;        /* mi_crossembed_wgrad_workspace(Cin, kmax, nwg) floats; workgroups walking the pixel tiles (e.g. 256) */
} mi_crossembed_wgrad_params;
long long mi_crossembed_wgrad_workspace(int Cin, int kmax, int nwg);
int mi_crossembed_wgrad(const mi_crossembed_wgrad_params* p, void* stream);

/* the core of the folded cross-attention of the training graph (layers.py:220-251 with keys / values mapped into the token's channel space):
 *   out[b][i][:] = sum_h sum_j softmax_j(q[b][i] . kf[b][h][j]) vf[b][h][j]        (mask[b][j] == 0: row j takes no part)
 * forward (saves the per-(token, head) logsumexp) and backward (dq, and per-token-chunk partials of dkf / dvf that the caller adds) without the
 * [tokens x heads x context] score tensor.  fp32 VALU kernels; C in {8, 16, 32}, J * C <= 6144. */
typedef struct mi_folded_attn_params {
    int B, n, H, J, C, nchunk;
    const float* q;                 /* [B][n][C] */
    const float* kf; const float* vf;   /* [B][H][J][C] */
    const uint8_t* mask;            /* [B][J] or NULL */
    float* out;                     /* [B][n][C] (forward) */
    float* lse;                     /* [B][n][H]: written by the forward, read by the backward (log2 domain) */
    const float* dout;              /* [B][n][C] (backward) */
    float* dsum;                    /* [B][n][H] workspace of the backward */
    float* dq;                      /* [B][n][C] */
    float* dkf; float* dvf;         /* [nchunk][B][H][J][C] partials over token chunks */
    float* oh;                      /* [B][n][H][C] per-head outputs: written by the forward, read by the backward; or NULL (one more pass) */
} mi_folded_attn_params;
int mi_folded_attn_fwd(const mi_folded_attn_params* p, void* stream);
int mi_folded_attn_bwd(const mi_folded_attn_params* p, void* stream);

/* ---- optimiser step of the training path: torch.optim.Adam's update for every tensor of a model in ONE launch ---------------------------
 * (reference: train.py:99-100, training.py:375-377).  `tensors`, `chunk_tensor`, `chunk_off` are DEVICE arrays: one mi_adam_tensor per
 * parameter, and one (tensor index, chunk offset) pair per launched workgroup (chunk k of a tensor covers elements [k * chunk, (k + 1) * chunk)).
 * bias_correction1/2 = 1 - beta^step (computed by the host: the step count is host state, as in torch); grad_scale: optional device scalar every
 * gradient is multiplied by (a clipping coefficient computed on the device), or NULL. */
typedef struct { float* p; const float* g; float* m; float* v; long long n; } mi_adam_tensor;
typedef struct {
    const mi_adam_tensor* tensors;
    const int* chunk_tensor; const int* chunk_off;
    int nchunks, chunk;
    float lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2;
    float one_minus_beta1, one_minus_beta2;     /* formed in double by the host like torch's Python scalars (1 - 0.999f is off by 5e-5 relative) */
    const float* grad_scale;
} mi_adam_params;
int mi_adam_step(const mi_adam_params* p, void* stream);

/* ---- HIP graphs: capture a sequence of the calls above once, replay it per timestep ------- */
int mi_graph_begin(void* stream);
int mi_graph_end(void* stream, void** graph_exec);
int mi_graph_launch(void* graph_exec, void* stream);
int mi_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif
