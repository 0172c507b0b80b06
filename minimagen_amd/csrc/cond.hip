// K1 / K2 / K5 and the context fold for K9: everything that turns (timestep, text embeddings)
// into the vectors the image kernels consume.  All of it is tiny (kFLOPs..MFLOPs); one
// workgroup per (conditional|null, sample) row, plain fp32 dot products in reference order.
#include "common.hip.h"

#define MI_MAX_TCD 4096
#define MI_MAX_CD 1024

namespace {

// One output of a Linear: the row is walked in float4 steps with four independent accumulators (these kernels are latency chains
// of tiny mat-vecs; a single dependent FMA chain over scalar loads made cond_step 21 us per denoising step).
__device__ __forceinline__ float dot_row(const float* __restrict__ w, const float* __restrict__ x, int n) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int k = 0;
    if ((n & 3) == 0 && ((reinterpret_cast<size_t>(w) | reinterpret_cast<size_t>(x)) & 15) == 0) {
        for (; k + 4 <= n; k += 4) {
            const float4 wv = *reinterpret_cast<const float4*>(w + k), xv = *reinterpret_cast<const float4*>(x + k);
            a0 = fmaf(wv.x, xv.x, a0);
            a1 = fmaf(wv.y, xv.y, a1);
            a2 = fmaf(wv.z, xv.z, a2);
            a3 = fmaf(wv.w, xv.w, a3);
        }
    }
    for (; k < n; ++k) a0 = fmaf(w[k], x[k], a0);
    return (a0 + a1) + (a2 + a3);
}

// LayerNorm over n values held in LDS `v` (every work-item computes the same statistics).
__device__ __forceinline__ void ln_stats(const float* v, int n, float& mean, float& rstd) {
    float s = 0.0f;
    for (int k = 0; k < n; ++k) s += v[k];
    mean = s / (float)n;
    float q = 0.0f;
    for (int k = 0; k < n; ++k) { const float d = v[k] - mean; q = fmaf(d, d, q); }
    rstd = 1.0f / sqrtf(q / (float)n + 1e-5f);
}

__global__ __launch_bounds__(256) void text_cond_kernel(const mi_text_cond_params p) {
    __shared__ __attribute__((aligned(16))) float pooled[MI_MAX_CD];
    __shared__ __attribute__((aligned(16))) float hid[MI_MAX_TCD];
    const int tid = threadIdx.x, NT = 256;
    const int bb = blockIdx.x, b = bb % p.B;
    const bool keep = p.keep[bb] != 0;
    float* crow = p.c_text + (size_t)bb * p.max_len * p.cd;
    // 1. projected / null tokens (pre-norm) -> c_text
    for (int j = tid; j < p.max_len; j += NT) {
        // Unet.py:581-603: rows past the caption are zero-PADDED after the projection (no bias); the padded mask is False there, so
        // with a mask they become null embeddings -- but WITHOUT a mask every row of a kept sample stays, the padding as exact zeros
        const bool in_caption = j < p.L;
        const bool valid = keep && (p.text_mask == nullptr || (in_caption && p.text_mask[(size_t)b * p.L + j] != 0));
        const float* e = p.text_embeds + ((size_t)b * p.L + (in_caption ? j : 0)) * p.E;
        for (int o = 0; o < p.cd; ++o) {
            float v;
            if (valid) v = in_caption ? dot_row(p.text_to_cond.w + (size_t)o * p.E, e, p.E) + (p.text_to_cond.b ? p.text_to_cond.b[o] : 0.0f) : 0.0f;
            else v = p.null_text_embed[(size_t)j * p.cd + o];
            crow[(size_t)j * p.cd + o] = v;
        }
    }
    __syncthreads();
    // 2. mean over all max_len rows (Unet.py:610)
    for (int o = tid; o < p.cd; o += NT) {
        float s = 0.0f;
        for (int j = 0; j < p.max_len; ++j) s += crow[(size_t)j * p.cd + o];
        pooled[o] = s / (float)p.max_len;
    }
    __syncthreads();
    // 3. to_text_non_attn_cond
    float mean, rstd;
    ln_stats(pooled, p.cd, mean, rstd);
    __syncthreads();
    for (int o = tid; o < p.cd; o += NT) pooled[o] = (pooled[o] - mean) * rstd * p.ln_w[o] + p.ln_b[o];
    __syncthreads();
    for (int o = tid; o < p.tcd; o += NT) hid[o] = mi_silu(dot_row(p.h1.w + (size_t)o * p.cd, pooled, p.cd) + p.h1.b[o]);
    __syncthreads();
    for (int o = tid; o < p.tcd; o += NT) {
        const float v = dot_row(p.h2.w + (size_t)o * p.tcd, hid, p.tcd) + p.h2.b[o];
        p.text_hiddens[(size_t)bb * p.tcd + o] = keep ? v : p.null_text_hidden[o];
    }
    // 4. norm_cond on each text row (LayerNorm is per row, so the text rows of c never change during sampling)
    for (int j = tid; j < p.max_len; j += NT) {
        float* r = crow + (size_t)j * p.cd;
        float s = 0.0f;
        for (int k = 0; k < p.cd; ++k) s += r[k];
        const float m = s / (float)p.cd;
        float q = 0.0f;
        for (int k = 0; k < p.cd; ++k) { const float d = r[k] - m; q = fmaf(d, d, q); }
        const float rs = 1.0f / sqrtf(q / (float)p.cd + 1e-5f);
        for (int k = 0; k < p.cd; ++k) r[k] = (r[k] - m) * rs * p.norm_w[k] + p.norm_b[k];
    }
}

__device__ void time_trio(const mi_linear& th, const mi_linear& tc, const mi_linear& tt, const float* freq, long long tval, int dim,
                          float* emb, float* hid, float* t_acc, float* tok, bool accumulate, int tid, int NT) {
    const int half = dim / 2;
    for (int j = tid; j < half; j += NT) {
        const float arg = (float)tval * freq[j];
        emb[j] = sinf(arg);
        emb[half + j] = cosf(arg);
    }
    __syncthreads();
    for (int o = tid; o < th.out; o += NT) hid[o] = mi_silu(dot_row(th.w + (size_t)o * th.in, emb, th.in) + th.b[o]);
    __syncthreads();
    for (int o = tid; o < tc.out; o += NT) {
        const float v = dot_row(tc.w + (size_t)o * tc.in, hid, tc.in) + tc.b[o];
        t_acc[o] = accumulate ? t_acc[o] + v : v;
    }
    for (int o = tid; o < tt.out; o += NT) tok[o] = dot_row(tt.w + (size_t)o * tt.in, hid, tt.in) + tt.b[o];
    __syncthreads();
}

__global__ __launch_bounds__(256) void cond_step_kernel(const mi_cond_step_params p) {
    __shared__ __attribute__((aligned(16))) float emb[MI_MAX_CD];
    __shared__ __attribute__((aligned(16))) float hid[MI_MAX_TCD];
    __shared__ __attribute__((aligned(16))) float tvec[MI_MAX_TCD];
    __shared__ __attribute__((aligned(16))) float tok[4 * MI_MAX_CD];
    const int tid = threadIdx.x, NT = 256;
    const int bb = blockIdx.x, b = bb % p.B;
    const bool lowres = p.lth.w != nullptr;
    time_trio(p.th, p.tc, p.tt, p.freq, (long long)p.time[b], p.dim, emb, hid, tvec, tok, false, tid, NT);
    if (lowres)
        time_trio(p.lth, p.ltc, p.ltt, p.freq, (long long)p.lowres_time[b], p.dim, emb, hid, tvec, tok + p.ntok * p.cd, true, tid, NT);
    const int ntot = p.ntok * (lowres ? 2 : 1);
    for (int o = tid; o < p.tcd; o += NT) {
        float v = tvec[o];
        if (p.text_hiddens) v += p.text_hiddens[(size_t)bb * p.tcd + o];
        if (p.t_out) p.t_out[(size_t)bb * p.tcd + o] = v;
        hid[o] = mi_silu(v);          // SiLU(t), input of every time_mlp
        if (p.silu_out) p.silu_out[(size_t)bb * p.tcd + o] = hid[o];
    }
    // norm_cond over each time-token row
    for (int r = tid; r < ntot; r += NT) {
        const float* row = tok + r * p.cd;
        float s = 0.0f;
        for (int k = 0; k < p.cd; ++k) s += row[k];
        const float m = s / (float)p.cd;
        float q = 0.0f;
        for (int k = 0; k < p.cd; ++k) { const float d = row[k] - m; q = fmaf(d, d, q); }
        const float rs = 1.0f / sqrtf(q / (float)p.cd + 1e-5f);
        if (p.c_time)
            for (int k = 0; k < p.cd; ++k)
                p.c_time[((size_t)bb * ntot + r) * p.cd + k] = (row[k] - m) * rs * p.norm_w[k] + p.norm_b[k];
    }
    __syncthreads();
    if (p.ss)
        for (int r = tid; r < p.time_mlps.out; r += NT)
            p.ss[(size_t)bb * p.time_mlps.out + r] = dot_row(p.time_mlps.w + (size_t)r * p.tcd, hid, p.tcd) + p.time_mlps.b[r];
}

// context rows -> MFMA A-operand fragments of the folded cross-attention (see minimagen_hip.h).  Modes 1 / 2 split the work into
// "fold every step's rows once per sample()" (table of (g, vw) pairs) and "scatter one step's rows" (per denoising step).
__global__ __launch_bounds__(256) void attn_fold_rows_kernel(const mi_attn_fold_params p) {
    const int bb = blockIdx.x, blk = blockIdx.y;
    if (blk == p.n_blocks) {          // mode 2 only: this step's scale/shift rows
        const float* src = p.ss_all + ((size_t)(*p.t_state - p.t_off) * gridDim.x + bb) * p.ss_n;
        for (int i = threadIdx.x; i < p.ss_n; i += 256) p.ss[(size_t)bb * p.ss_n + i] = src[i];
        return;
    }
    const int C = p.C, NGP = (C / 4) < 4 ? 4 : (C / 4), MT = (C + 15) / 16, FR = NGP + 4 * MT;
    const float* mg = p.blk[blk].mg;
    const float* mv = p.blk[blk].mv;
    float* gv = p.blk[blk].gv + (size_t)bb * p.heads * p.JT * 64 * FR;
    const int first = p.write_null ? -1 : 0;
    const int total = (p.nrows - first) * p.heads * C;
    // row of the compact table: mode 2 -- one row per (timestep, batch row); mode 3 -- one row per timestep, shared by every batch row (the
    // time tokens depend on the timestep only); mode 1 -- the row being built
    const size_t trow = p.mode == 2 ? (size_t)(*p.t_state - p.t_off) * gridDim.x + bb : (p.mode == 3 ? (size_t)(*p.t_state - p.t_off) : (size_t)bb);
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int a = idx % C, h = (idx / C) % p.heads, r = idx / (C * p.heads) + first;
        float g, v;
        int j;
        if (r < 0) {
            j = 0;
            g = p.blk[blk].g0[h * C + a];
            v = p.blk[blk].v0[h * C + a];
        } else {
            j = p.row0 + r;
            float* tab = p.mode ? p.blk[blk].table + (((trow * p.nrows + r) * p.heads + h) * C + a) * 2 : nullptr;
            if (p.mode >= 2) {
                g = tab[0];
                v = tab[1];
            } else {
                const float* c = p.c_rows + (size_t)bb * p.c_stride_b + (size_t)r * p.cd;
                g = dot_row(mg + ((size_t)h * C + a) * p.cd, c, p.cd);
                v = dot_row(mv + ((size_t)h * C + a) * p.cd, c, p.cd);
                if (p.mode == 1) { tab[0] = g; tab[1] = v; continue; }
            }
        }
        const int jt = j >> 4, jm = j & 15;
        if (p.frag_f16) {
            // fp16x3 fragments, per lane and tile: KC x {4 halves G hi, 4 halves G lo}, then MT V chunks of 8 halves.  The V chunks are
            // arranged for the K = 32 PV instruction, which contracts a PAIR of context tiles (t0, t1 = t0 + 1): the chunk of the even
            // tile holds {4 VW hi(t0), 4 VW hi(t1)}, the chunk of the odd tile {4 VW lo(t0), 4 VW lo(t1)} -- each is one 16-byte A
            // operand as it lies.  The tile count is padded to even (the partner of an odd last tile stays zero: the buffer is
            // allocated zero-filled).
            const int KC = (C + 15) / 16, MTh = (C + 15) / 16, FRH = 8 * KC + 8 * MTh;          // halves per lane
            const int JTS = (p.JT + 1) & ~1;
            _Float16* th = reinterpret_cast<_Float16*>(p.blk[blk].gv) + (((size_t)bb * p.heads + h) * JTS + jt) * 64 * FRH;
            g = ldexpf(g, p.blk[blk].g_exp);            // exact power-of-two scalings: keep hi AND lo in the fp16 normal range
            v = ldexpf(v, p.blk[blk].v_exp);
            const _Float16 ghi = (_Float16)g, glo = (_Float16)(g - (float)ghi);
            const _Float16 vhi = (_Float16)v, vlo = (_Float16)(v - (float)vhi);
            _Float16* lg = th + (size_t)(jm + 16 * ((a & 15) >> 2)) * FRH + 8 * (a >> 4);      // A[m=j][k=a]: lane (j, a/4), element a%4
            lg[a & 3] = ghi;
            lg[4 + (a & 3)] = glo;
            _Float16* tv = th - (size_t)(jt & 1) * 64 * FRH;                                     // the even tile of the pair
            _Float16* lv = tv + (size_t)((a & 15) + 16 * (jm >> 2)) * FRH + 8 * KC + 8 * (a >> 4) + 4 * (jt & 1);   // A[m=a][k=j]: lane (a, j/4), element j%4
            lv[jm & 3] = vhi;
            lv[(size_t)64 * FRH + (jm & 3)] = vlo;                                               // same slot of the odd tile's chunk
            continue;
        }
        float* tilep = gv + ((size_t)h * p.JT + jt) * 64 * FR;
        tilep[(size_t)(jm + 16 * (a & 3)) * FR + (a >> 2)] = g;                           // A[m=j][k=a] of QK^T, k-step a/4
        tilep[(size_t)((a & 15) + 16 * (jm >> 2)) * FR + NGP + 4 * (a >> 4) + (jm & 3)] = v;   // A[m=a][k=j] of PV, step j%4
    }
}

}  // namespace

extern "C" int mi_attn_fragment_floats(int C) { return ((C / 4) < 4 ? 4 : (C / 4)) + 4 * ((C + 15) / 16); }

extern "C" int mi_text_cond_fwd(const mi_text_cond_params* p, void* stream) {
    if (p->cd > MI_MAX_CD || p->tcd > MI_MAX_TCD || p->B2 <= 0 || p->B <= 0) { mi_set_error("mi_text_cond_fwd: cd/tcd too large or empty batch"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(text_cond_kernel, dim3(p->B2), dim3(256), 0, (hipStream_t)stream, *p);
    return mi_check_launch("text_cond_kernel");
}

extern "C" int mi_cond_step_fwd(const mi_cond_step_params* p, void* stream) {
    if (p->cd > MI_MAX_CD || p->tcd > MI_MAX_TCD || p->dim > MI_MAX_CD || p->B2 <= 0 || p->ntok * p->cd > 2 * MI_MAX_CD) { mi_set_error("mi_cond_step_fwd: dims too large"); return MI_ERR_INVALID; }
    if ((p->lth.w != nullptr) && p->lowres_time == nullptr) { mi_set_error("mi_cond_step_fwd: lowres U-Net needs lowres_time"); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(cond_step_kernel, dim3(p->B2), dim3(256), 0, (hipStream_t)stream, *p);
    return mi_check_launch("cond_step_kernel");
}

extern "C" int mi_attn_fold_rows(const mi_attn_fold_params* p, void* stream) {
    const bool stage = p->mode == 2 || p->mode == 3;
    if (p->n_blocks < (stage ? 0 : 1) || p->n_blocks > MI_ATTN_MAX_BLOCKS || (p->n_blocks && (p->C % 4) != 0) || p->B2 <= 0) { mi_set_error("mi_attn_fold_rows: bad n_blocks / C"); return MI_ERR_INVALID; }
    if (p->n_blocks && p->mode != 1 && p->row0 + p->nrows > p->JT * 16) { mi_set_error("mi_attn_fold_rows: rows beyond the padded context"); return MI_ERR_INVALID; }
    if (p->mode < 0 || p->mode > 3 || (p->mode && p->write_null) || (stage && (!p->t_state || (p->ss_n > 0 && (!p->ss_all || !p->ss))))) { mi_set_error("mi_attn_fold_rows: bad mode / table arguments"); return MI_ERR_INVALID; }
    for (int k = 0; k < p->n_blocks; ++k) if (p->mode && !p->blk[k].table) { mi_set_error("mi_attn_fold_rows: mode %d needs blk[].table", p->mode); return MI_ERR_INVALID; }
    hipLaunchKernelGGL(attn_fold_rows_kernel, dim3(p->B2, p->n_blocks + ((stage && p->ss_n > 0) ? 1 : 0)), dim3(256), 0, (hipStream_t)stream, *p);
    return mi_check_launch("attn_fold_rows_kernel");
}
