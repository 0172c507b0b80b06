"""Host-side weight re-layout for the HIP kernels (done once per state-dict version).

Nothing here is on the per-step path; everything is derived from the reference's own parameter
tensors (state-dict layout of SURVEY.md Appendix B-9) so reference checkpoints load unchanged.
"""
from __future__ import annotations

import math

import torch

LOG2E = 1.4426950408889634


def pack_conv_weight(w: torch.Tensor, cout_tile: int) -> torch.Tensor:
    """[Cout][Cin][k][k] -> [Cin][k][k][Cout_pad] (all output channels of a tap contiguous)."""
    cout = w.shape[0]
    pad = (cout + cout_tile - 1) // cout_tile * cout_tile
    out = torch.zeros(w.shape[1], w.shape[2], w.shape[3], pad, dtype=torch.float32, device=w.device)
    out[..., :cout] = w.detach().permute(1, 2, 3, 0)
    return out.contiguous()


def fold_parallel_1x1(w3: torch.Tensor, b3, w1: torch.Tensor, b1):
    """Parallel(conv3x3, conv1x1) (layers.py:346-356, Unet.py:233-234) == one 3x3 conv whose centre tap
    carries the 1x1 weights (exact in real arithmetic)."""
    w = w3.detach().clone()
    w[:, :, 1, 1] += w1.detach()[:, :, 0, 0]
    return w, (b3.detach() + b1.detach())


def fold_cross_attention(to_q, to_kv, to_out, null_kv, heads: int, dim_head: int = 64):
    """Folded matrices of the bottleneck cross-attention (see include/minimagen_hip.h, K9).

    to_q [H*D][C], to_kv [2*H*D][cd] (k rows first, layers.py:226), to_out [C][H*D], null_kv [2][D].
    Returns mg, mv [H][C][cd] and g0, v0 [H][C] (fp32), computed in fp64.
    """
    dev = to_q.device
    H, D = heads, dim_head
    # host arithmetic (once per weight version): keeps library GEMMs off the GPU timeline
    to_q, to_kv, to_out, null_kv = (t.detach().to('cpu') for t in (to_q, to_kv, to_out, null_kv))
    q = to_q.detach().double().reshape(H, D, -1)               # [H][D][C]
    kv = to_kv.detach().double()
    k = kv[:H * D].reshape(H, D, -1)                            # [H][D][cd]
    v = kv[H * D:].reshape(H, D, -1)
    o = to_out.detach().double().reshape(-1, H, D).permute(1, 0, 2)   # [H][C][D]
    nk, nv = null_kv.detach().double().unbind(0)
    scale = D ** -0.5
    mg = torch.einsum('hda,hdb->hab', q, k) * (scale * LOG2E)
    mv = torch.einsum('had,hdb->hab', o, v)
    g0 = torch.einsum('hda,d->ha', q, nk) * (scale * LOG2E)
    v0 = torch.einsum('had,d->ha', o, nv)
    f = lambda t: t.float().contiguous().to(dev)
    return f(mg), f(mv), f(g0), f(v0)


def sinusoid_freq(dim: int, device) -> torch.Tensor:
    """exp(arange(dim/2) * -(ln 1e4 / (dim/2 - 1))) exactly as SinusoidalPosEmb builds it (layers.py:461-463)."""
    half = dim // 2
    k = math.log(10000) / (half - 1)
    return torch.exp(torch.arange(half) * -k).float().contiguous().to(device)


RP_PERM = (0, 2, 1, 3)      # lane group -> input row of the row-paired matrix-core conv (csrc/conv_rp.hip)


def rp_weight_exponent(mx: float) -> int:
    """power of two that brings a weight tensor's maximum magnitude into [128, 256)"""
    exp = 0 if (mx == 0.0 or not math.isfinite(mx)) else 7 - math.floor(math.log2(mx))
    return max(-100, min(100, exp))


def pack_conv_weight_rp(w: torch.Tensor, exp=None):
    """[Cout][Cin][3][3] (or [Cout][Cres][1][1]) fp32 -> (fragments, exponent) for csrc/conv_rp.hip.

    B operand of v_mfma_f32_16x16x32_f16 with N = (output-row parity dy, 8 output channels), K = (4 input rows, 8 input channels)
    per horizontal tap kx: [ceil(Cin/8)][steps][ceil(Cout/8)][64 lanes][8 hi | 8 lo] fp16, lane (lq, lg) holding
    W[co = 8jt + (lq & 7)][ci = 8k + e][ky = RP_PERM[lg] - (lq >> 3)][kx = step] (zero where ky is not a tap; a 1x1 residual conv
    is the centre tap).  The weights are pre-scaled by 2^exponent so that max|w| lands in [128, 256): the fp16 split hi = fp16(w'),
    lo = fp16(w' - hi) then keeps ~22 bits whatever the magnitude of the checkpoint's weights; the kernel undoes the scale."""
    cout, cin, kh, kw = w.shape
    assert (kh, kw) in ((3, 3), (1, 1), (4, 4))
    if exp is None:
        wd = w.detach().double().cpu()
        exp = rp_weight_exponent(float(wd.abs().max()))
    else:                       # the training path packs on the device, exponents of all layers fetched with one host round trip
        wd = w.detach().double()
    dev = wd.device
    ws = wd * (2.0 ** exp)
    if kh == 4:
        # k4 s2 (Downsample): N = 16 output channels of ONE output row, K = (4 vertical taps <-> lane group, 8 input channels), one
        # step per horizontal tap: lane (lq, lg) holds W[co = 16jt + lq][ci = 8k + e][ky = lg][kx = step]
        nj, ko = -(-cout // 16), -(-cin // 8)
        wp = torch.zeros(nj * 16, ko * 8, 4, 4, dtype=torch.float64, device=dev)
        wp[:cout, :cin] = ws
        lane = torch.arange(64, device=dev)
        lq, lg = lane & 15, lane >> 4
        co = 16 * torch.arange(nj, device=dev)[:, None] + lq[None, :]
        ci = 8 * torch.arange(ko, device=dev)[:, None] + torch.arange(8, device=dev)[None, :]
        out = wp[co[None, None, :, :, None], ci[:, None, None, None, :], lg[None, None, None, :, None], torch.arange(4, device=dev)[None, :, None, None, None]]
        hi = out.float().half()
        lo = (out - hi.double()).float().half()
        return torch.cat((hi, lo), dim=-1).contiguous().to(w.device), exp
    nj, ko, steps = -(-cout // 8), -(-cin // 8), kw
    wp = torch.zeros(nj * 8, ko * 8, 3, kw, dtype=torch.float64, device=dev)
    if kh == 3:
        wp[:cout, :cin] = ws
    else:
        wp[:cout, :cin, 1, :] = ws[:, :, 0, :]          # centre row; its single "step" is the centre column
    lane = torch.arange(64, device=dev)
    lq, lg = lane & 15, lane >> 4
    r = torch.tensor(RP_PERM, device=dev)[lg]
    ky = r - (lq >> 3)
    live = ((ky >= 0) & (ky <= 2)).double()
    kyc = ky.clamp(0, 2)
    co = (8 * torch.arange(nj, device=dev)[:, None] + (lq & 7)[None, :])                 # [nj][64]
    ci = (8 * torch.arange(ko, device=dev)[:, None] + torch.arange(8, device=dev)[None, :])         # [ko][8]
    # out[k, s, jt, lane, e] = wp[co[jt, lane], ci[k, e], ky[lane], s]  (one gather, no Python loop over the channel tiles)
    out = wp[co[None, None, :, :, None], ci[:, None, None, None, :], kyc[None, None, None, :, None], torch.arange(steps, device=dev)[None, :, None, None, None]]
    out = out * live[None, None, None, :, None]
    hi = out.float().half()
    lo = (out - hi.double()).float().half()
    return torch.cat((hi, lo), dim=-1).contiguous().to(w.device), exp


def pack_conv_weight_ig(w: torch.Tensor, exp=None):
    """[Cout][Cin][3][3] (or [Cout][Cres][1][1]) fp32 -> (fragments, exponent) for csrc/conv_wide.hip (the wide presets' GEMM kernel).

    B operand of v_mfma_f32_16x16x32_f16 with N = 16 output channels, K = 32 input channels of ONE tap:
    [Cin/32][taps][Cout/16][hi | lo][64 lanes][8 halves] fp16, lane (lq, lg) holding W[co = 16 nt + lq][ci = 32 g + 8 lg + e][tap]; Cin in
    multiples of 32, Cout of 16.  Pre-scaled by 2^exponent exactly like pack_conv_weight_rp (max|w| in [128, 256))."""
    cout, cin, kh, kw = w.shape
    assert (kh, kw) in ((3, 3), (1, 1)) and cin % 32 == 0 and cout % 16 == 0
    wd = w.detach().double()
    if exp is None:
        exp = rp_weight_exponent(float(wd.abs().max()))
    ws = (wd * (2.0 ** exp)).reshape(cout // 16, 16, cin // 32, 4, 8, kh * kw)          # [nt][lq][g][lg][e][tap]
    out = ws.permute(2, 5, 0, 3, 1, 4).reshape(cin // 32, kh * kw, cout // 16, 64, 8)     # [g][tap][nt][lane = 16 lg + lq][e]
    hi = out.float().half()
    lo = (out - hi.double()).float().half()
    return torch.stack((hi, lo), dim=3).contiguous().to(w.device), exp                   # [g][tap][nt][2][64][8]


def attn_f16_exponents(mg, mv, g0, v0, cmax: float, xmax: float):
    """Power-of-two operand scalings (x_exp, g_exp, v_exp) of the fp16x3 cross-attention from magnitude BOUNDS: |x^| <= xmax,
    |g_j| <= max_row sum_b |mg| * cmax (context rows bounded by cmax), likewise vw; each scaled bound lands at <= 2^12, which leaves a
    factor 16 to the fp16 maximum and ~2^22 of dynamic range below it in which hi and lo are both normal numbers."""
    def k(bound):
        bound = float(bound)
        if not math.isfinite(bound) or bound <= 0.0:
            return 0
        return max(-40, min(40, 12 - math.ceil(math.log2(bound))))
    gb = max(float(mg.abs().sum(-1).max()) * cmax, float(g0.abs().max()))
    vb = max(float(mv.abs().sum(-1).max()) * cmax, float(v0.abs().max()))
    return k(xmax), k(gb), k(vb)


def layernorm_bound(weight, bias, n: int) -> float:
    """max |LayerNorm(x)_i| over any input: a normalised element is at most sqrt(n - 1) in magnitude"""
    with torch.no_grad():           # ONE device -> host copy for both maxima
        if bias is not None:
            w, b = torch.stack((weight.detach().abs().max(), bias.detach().abs().max())).tolist()
        else:
            w, b = float(weight.detach().abs().max()), 0.0
    return w * math.sqrt(max(n - 1, 1)) + b


CE_TILES = ((0, 0), (0, 2), (1, 0), (2, 0))       # N tiles of the matrix-core CrossEmbed: (conv index, first output channel)
CE_TABLE_ROWS = 32


def _crossembed_toeplitz(scaled, cin: int):
    """the three (pre-scaled, channel-sliced) CrossEmbed weights laid out as pack_crossembed_mfma's table, before the hi / lo split:
    float64 [row][h][hi|lo][co2][lg][e = 4 dx + ci] with the values in the hi plane"""
    tab = torch.zeros(CE_TABLE_ROWS, 2, 2, 2, 4, 8, dtype=torch.float64)       # [row][h][hi|lo][co2][lg][e = 4 dx + ci]
    w3, w7, w15 = scaled
    for lg in range(4):
        for dx in range(2):
            for h in range(2):                                                   # k15: rows 16 + ky
                kx = 2 * lg + dx + 8 * h
                if kx < 15:
                    tab[16:31, h, 0, :, lg, 4 * dx:4 * dx + cin] = w15[:, :, :, kx].permute(2, 0, 1)      # [ky][co2][ci]
                kx = 2 * lg + dx + 8 * h - 4                                     # k7 on the same fragments: rows 8 + ky
                if 0 <= kx < 7:
                    tab[8:15, h, 0, :, lg, 4 * dx:4 * dx + cin] = w7[:, :, :, kx].permute(2, 0, 1)
            kx = 2 * (lg & 1) + dx                                               # k3: rows 4 t + q
            if kx < 3:
                for t in range(2):
                    for q in range(4):
                        ky = q - 1 + (lg >> 1)
                        if 0 <= ky < 3:
                            tab[4 * t + q, 0, 0, :, lg, 4 * dx:4 * dx + cin] = w3[2 * t:2 * t + 2, :, ky, kx]  # [co2][ci]
    return tab


def pack_crossembed_mfma(ws, chan0: int, cin: int):
    """CrossEmbedLayer weights (dim_scales (4, 2, 2), kernel sizes (3, 7, 15); layers.py:254-305) for crossembed_mfma_kernel.

    The three convs are Toeplitz GEMMs per 16-pixel x 8-row group: N = (output channel pair co2, output row dy); a lane supplies 2
    adjacent columns x 4 channels (8 fp16) of one window row, the 4 lane groups lg make up K = 32.  A B operand depends on (window row
    - dy) only, so the table holds one row per vertical tap and every lane picks its own row (row 31 is all zero: taps outside a kernel):
      k15 (rows 16 .. 30 = ky): two steps per window row, halves h = 0 / 1: kx = 2 lg + dx + 8 h (kx = 15: zero);
      k7  (rows 8 .. 14 = ky; 15 zero): on the k15 fragments of window rows 4 .. 17: kx = 2 lg + dx + 8 h - 4;
      k3  (rows 0 .. 3 for channels 0-1, 4 .. 7 for channels 2-3): one step per window row PAIR (r, r + 1): lane groups 0-1 take row r,
          2-3 row r + 1, columns kx = 2 (lg & 1) + dx (kx = 3: zero); table row q = (r - dy - 6) + 1 in 0 .. 3, ky = q - 1 + (lg >> 1).
    Layout [row 32][h 2][hi | lo][co2 2][lg 4][8 fp16 = 4 dx + ci]; k3 uses the h = 0 half.  ws: the three [cout][Cin_total][k][k]
    weights; channels chan0 .. chan0 + cin - 1 are packed (cin <= 4).  Returns (table [32][256] fp16, [3] power-of-two exponents: each
    conv is pre-scaled so that max|w| lands in [128, 256))."""
    assert cin <= 4 and len(ws) == 3 and [w.shape[-1] for w in ws] == [3, 7, 15] and [w.shape[0] for w in ws] == [4, 2, 2]
    exps, scaled = [], []
    for w in ws:
        wd = w.detach().double().cpu()[:, chan0:chan0 + cin]
        mx = float(wd.abs().max())
        e = 0 if (mx == 0.0 or not math.isfinite(mx)) else 7 - math.floor(math.log2(mx))
        e = max(-100, min(100, e))
        exps.append(e)
        scaled.append(wd * (2.0 ** e))
    tab = _crossembed_toeplitz(scaled, cin)
    hi = tab[:, :, 0].float().half()
    lo = (tab[:, :, 0] - hi.double()).float().half()
    out = torch.stack((hi, lo), dim=2)                                           # [row][h][hl][co2][lg][8]
    return out.reshape(CE_TABLE_ROWS, 256).contiguous().to(ws[0].device), exps


def crossembed_mfma_gather_index(shapes, chan0: int, cin: int):
    """pack_crossembed_mfma as a gather: int64 [32 * 2 * 2 * 4 * 8] positions into cat(w3.flatten(), w7.flatten(), w15.flatten(), [0]) for the
    table's [row][h][co2][lg][e] elements (the appended zero for taps outside a kernel) -- depends on the weights' SHAPES only, so a training
    step can rebuild the tables on the device from the updated weights without a host copy (pack_crossembed_mfma_device)"""
    idx_w, off = [], 0
    for sh in shapes:
        n = 1
        for d in sh:
            n *= d
        idx_w.append((torch.arange(n, dtype=torch.float64) + (off + 1)).reshape(sh)[:, chan0:chan0 + cin])
        off += n
    tab = _crossembed_toeplitz(idx_w, cin)[:, :, 0]                              # 0 where no tap, else flat position + 1
    idx = tab.round().long() - 1
    idx[idx < 0] = off
    return idx.reshape(-1).contiguous()


def pack_crossembed_mfma_device(ws, idx: torch.Tensor, exps):
    """the table of pack_crossembed_mfma built where the weights live: scale each member by 2^exps[i] (exact), gather into the table layout
    through crossembed_mfma_gather_index, split into fp16 hi | lo.  Same bits as the host packer for the same exponents."""
    flat = torch.cat([w.detach().reshape(-1) * (2.0 ** e) for w, e in zip(ws, exps)] + [ws[0].new_zeros(1)])
    t = flat[idx].reshape(CE_TABLE_ROWS, 2, 2, 4, 8)                             # [row][h][co2][lg][8] fp32
    hi = t.half()
    lo = (t - hi.float()).half()
    return torch.stack((hi, lo), dim=2).reshape(CE_TABLE_ROWS, 256).contiguous()
