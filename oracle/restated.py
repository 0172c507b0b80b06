"""CPU restatement (torch fp32, functional, state-dict driven) of the MinImagen
sampling hot path.  TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Every function cites the reference lines it follows (paths relative to
/root/reference).  The network structure is recovered from the state-dict keys
(SURVEY.md Appendix B-9), so one ``sd`` (a reference ``Unet.state_dict()``) is all
that is needed.  This file is validated against the reference itself by
tests/test_oracle.py (build container) and against tests/golden/*.pt everywhere.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import resize_restated

SD = Dict[str, torch.Tensor]
ATTN_DIM_HEAD = 64      # minimagen/Unet.py:86
MAX_TEXT_LEN = 256      # minimagen/Unet.py:150
SKIP_SCALE = 2 ** -0.5  # minimagen/Unet.py:194


# ----------------------------------------------------------------------------- small layers
# The reference computes in fp32 and so does this restatement.  tests/ may set COMPUTE_DTYPE = torch.float64 (with an fp64 state dict and
# fp64 inputs) to obtain a higher-precision value of the SAME algorithm -- used only to measure how far fp32 arithmetic itself (the
# reference's, the oracle's, the HIP path's) sits from the exact result on ill-conditioned weights; never for parity gates.
COMPUTE_DTYPE = torch.float32


def _linear(x, sd: SD, p: str):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _conv(x, sd: SD, p: str, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def sinusoidal_pos_emb(t: torch.Tensor, dim: int) -> torch.Tensor:
    """minimagen/layers.py:455-465."""
    half = dim // 2
    k = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half, dtype=COMPUTE_DTYPE) * -k)
    arg = t[:, None] * freq[None, :]          # int64 * fp32 -> fp32
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def layer_norm_custom(x, sd: SD, p: str):
    """minimagen/layers.py:333-343 (gamma parameter, beta zero buffer)."""
    return F.layer_norm(x, x.shape[-1:], sd[p + ".gamma"], sd[p + ".beta"])


def chan_layer_norm(x, g, eps=1e-5):
    """minimagen/layers.py:164-177."""
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) / (var + eps).sqrt() * g


def block(x, sd: SD, p: str, scale_shift=None, groups: int = 8):
    """minimagen/layers.py:131-145: GroupNorm -> [x*(scale+1)+shift] -> SiLU -> Conv3x3."""
    x = F.group_norm(x, groups, sd[p + ".groupnorm.weight"], sd[p + ".groupnorm.bias"], eps=1e-5)
    if scale_shift is not None:
        scale, shift = scale_shift
        x = x * (scale + 1) + shift
    x = F.silu(x)
    return _conv(x, sd, p + ".project", padding=1)


def cross_attention(x, context, sd: SD, p: str):
    """minimagen/layers.py:220-251.  x (b,n,dim), context (b,m,cdim); no mask on the hot path."""
    b = x.shape[0]
    x = layer_norm_custom(x, sd, p + ".norm")
    q = F.linear(x, sd[p + ".to_q.weight"])
    kv = F.linear(context, sd[p + ".to_kv.weight"])
    k, v = kv.chunk(2, dim=-1)
    heads = q.shape[-1] // ATTN_DIM_HEAD
    split = lambda t: t.reshape(b, t.shape[1], heads, ATTN_DIM_HEAD).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    nk, nv = sd[p + ".null_kv"].unbind(dim=-2)
    nk = nk.expand(b, heads, 1, ATTN_DIM_HEAD)
    nv = nv.expand(b, heads, 1, ATTN_DIM_HEAD)
    k = torch.cat((nk, k), dim=-2)
    v = torch.cat((nv, v), dim=-2)
    q = q * (ATTN_DIM_HEAD ** -0.5)
    sim = torch.einsum('bhid,bhjd->bhij', q, k)
    attn = sim.softmax(dim=-1, dtype=COMPUTE_DTYPE)
    out = torch.einsum('bhij,bhjd->bhid', attn, v)
    out = out.permute(0, 2, 1, 3).reshape(b, -1, heads * ATTN_DIM_HEAD)
    out = F.linear(out, sd[p + ".to_out.0.weight"])
    return layer_norm_custom(out, sd, p + ".to_out.1")


def self_attention(x, sd: SD, p: str):
    """minimagen/layers.py:52-104 (multi-query: one shared k/v head); no context/mask/bias on the hot path."""
    b = x.shape[0]
    x = layer_norm_custom(x, sd, p + ".norm")
    q = F.linear(x, sd[p + ".to_q.weight"])
    k, v = F.linear(x, sd[p + ".to_kv.weight"]).chunk(2, dim=-1)
    dh = k.shape[-1]
    heads = q.shape[-1] // dh
    q = q.reshape(b, -1, heads, dh).permute(0, 2, 1, 3) * (dh ** -0.5)
    nk, nv = sd[p + ".null_kv"].unbind(dim=-2)
    k = torch.cat((nk.expand(b, 1, dh), k), dim=-2)
    v = torch.cat((nv.expand(b, 1, dh), v), dim=-2)
    sim = torch.einsum('bhid,bjd->bhij', q, k)
    attn = sim.softmax(dim=-1, dtype=COMPUTE_DTYPE)
    out = torch.einsum('bhij,bjd->bhid', attn, v)
    out = out.permute(0, 2, 1, 3).reshape(b, -1, heads * dh)
    out = F.linear(out, sd[p + ".to_out.0.weight"])
    return layer_norm_custom(out, sd, p + ".to_out.1")


def _to_tokens(x):
    b, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(b, h * w, c)


def _from_tokens(t, shape):
    b, c, h, w = shape
    return t.reshape(b, h, w, c).permute(0, 3, 1, 2)


def transformer_block(x, sd: SD, p: str):
    """minimagen/layers.py:468-499 (+ ChanFeedForward :148-161)."""
    x = _from_tokens(self_attention(_to_tokens(x), sd, p + ".attn.fn"), x.shape) + x
    h = chan_layer_norm(x, sd[p + ".ff.0.g"])
    h = F.conv2d(h, sd[p + ".ff.1.weight"])
    h = F.gelu(h)
    h = chan_layer_norm(h, sd[p + ".ff.3.g"])
    h = F.conv2d(h, sd[p + ".ff.4.weight"])
    return h + x


def resnet_block(x, sd: SD, p: str, t=None, c=None):
    """minimagen/layers.py:417-439."""
    scale_shift = None
    if t is not None and (p + ".time_mlp.1.weight") in sd:
        te = _linear(F.silu(t), sd, p + ".time_mlp.1")[:, :, None, None]
        scale_shift = te.chunk(2, dim=1)
    h = block(x, sd, p + ".block1")
    if (p + ".cross_attn.fn.to_q.weight") in sd:
        assert c is not None
        h = _from_tokens(cross_attention(_to_tokens(h), c, sd, p + ".cross_attn.fn"), h.shape) + h
    h = block(h, sd, p + ".block2", scale_shift=scale_shift)
    res = _conv(x, sd, p + ".res_conv") if (p + ".res_conv.weight") in sd else x
    return h + res


def cross_embed(x, sd: SD, p: str):
    """minimagen/layers.py:298-305, stride 1, padding (k-1)//2."""
    outs = []
    i = 0
    while f"{p}.convs.{i}.weight" in sd:
        k = sd[f"{p}.convs.{i}.weight"].shape[-1]
        outs.append(_conv(x, sd, f"{p}.convs.{i}", padding=(k - 1) // 2))
        i += 1
    return torch.cat(outs, dim=1)


# ----------------------------------------------------------------------------- conditioning
def generate_t_tokens(sd: SD, time, lowres_noise_times=None):
    """minimagen/Unet.py:508-536."""
    dim = sd["to_time_hiddens.1.weight"].shape[1]
    cond_dim = sd["norm_cond.weight"].shape[0]

    def trio(prefix, times):
        hid = F.silu(_linear(sinusoidal_pos_emb(times, dim), sd, f"to_{prefix}time_hiddens.1"))
        t = _linear(hid, sd, f"to_{prefix}time_cond.0")
        tok = _linear(hid, sd, f"to_{prefix}time_tokens.0").reshape(times.shape[0], -1, cond_dim)
        return t, tok

    t, tokens = trio("", time)
    if "to_lowres_time_hiddens.1.weight" in sd:
        lt, ltok = trio("lowres_", lowres_noise_times)
        t = t + lt
        tokens = torch.cat((tokens, ltok), dim=-2)
    return t, tokens


def text_condition(sd: SD, text_embeds, text_mask, keep_mask, t, time_tokens):
    """minimagen/Unet.py:571-634.  keep_mask (b,) bool replaces prob_mask_like (all-True / all-False in sampling)."""
    text_tokens = None
    if text_embeds is not None:
        text_tokens = _linear(text_embeds, sd, "text_to_cond")[:, :MAX_TEXT_LEN]
        remainder = MAX_TEXT_LEN - text_tokens.shape[1]
        if remainder > 0:
            text_tokens = F.pad(text_tokens, (0, 0, 0, remainder))
        keep_embed = keep_mask[:, None, None]
        if text_mask is not None:
            tm = text_mask[:, :MAX_TEXT_LEN]
            if remainder > 0:
                tm = F.pad(tm, (0, remainder), value=False)
            keep_embed = tm[:, :, None] & keep_embed
        text_tokens = torch.where(keep_embed, text_tokens, sd["null_text_embed"])
        pooled = text_tokens.mean(dim=-2)
        h = F.layer_norm(pooled, pooled.shape[-1:], sd["to_text_non_attn_cond.0.weight"], sd["to_text_non_attn_cond.0.bias"])
        h = F.silu(_linear(h, sd, "to_text_non_attn_cond.1"))
        h = _linear(h, sd, "to_text_non_attn_cond.3")
        h = torch.where(keep_mask[:, None], h, sd["null_text_hidden"])
        t = t + h
    c = time_tokens if text_tokens is None else torch.cat((time_tokens, text_tokens), dim=-2)
    c = F.layer_norm(c, c.shape[-1:], sd["norm_cond.weight"], sd["norm_cond.bias"])
    return t, c


# ----------------------------------------------------------------------------- U-Net
def _count(sd: SD, fmt: str) -> int:
    n = 0
    while fmt.format(n) in sd:
        n += 1
    return n


def unet_forward(sd: SD, x, time, *, lowres_cond_img=None, lowres_noise_times=None,
                 text_embeds=None, text_mask=None, cond_drop_prob: float = 0., keep_mask=None):
    """minimagen/Unet.py:355-472."""
    b = x.shape[0]
    lowres = "to_lowres_time_hiddens.1.weight" in sd
    assert not (lowres and lowres_cond_img is None), 'low resolution conditioning image must be present'
    assert not (lowres and lowres_noise_times is None), 'low resolution conditioning noise time must be present'
    if keep_mask is None:
        assert cond_drop_prob in (0., 1.), "oracle needs an explicit keep_mask for 0<p<1"
        keep_mask = torch.full((b,), cond_drop_prob == 0., dtype=torch.bool)
    t, tokens = generate_t_tokens(sd, time, lowres_noise_times)
    t, c = text_condition(sd, text_embeds, text_mask, keep_mask, t, tokens)
    if lowres_cond_img is not None:
        x = torch.cat((x, lowres_cond_img), dim=1)
    x = cross_embed(x, sd, "init_conv")
    hiddens: List[torch.Tensor] = []
    n_levels = _count(sd, "downs.{}.1.block1.project.weight")
    for L in range(n_levels):
        if f"downs.{L}.0.weight" in sd:
            x = _conv(x, sd, f"downs.{L}.0", stride=2, padding=1)
        x = resnet_block(x, sd, f"downs.{L}.1", t, c)
        for k in range(_count(sd, f"downs.{L}.2." + "{}.block1.project.weight")):
            x = resnet_block(x, sd, f"downs.{L}.2.{k}", t)
            hiddens.append(x)
        if f"downs.{L}.3.attn.fn.to_q.weight" in sd:
            x = transformer_block(x, sd, f"downs.{L}.3")
        hiddens.append(x)
        if f"downs.{L}.4.weight" in sd:
            x = _conv(x, sd, f"downs.{L}.4", stride=2, padding=1)
        elif f"downs.{L}.4.fns.0.weight" in sd:
            x = _conv(x, sd, f"downs.{L}.4.fns.0", padding=1) + _conv(x, sd, f"downs.{L}.4.fns.1")
    x = resnet_block(x, sd, "mid_block1", t, c)
    if "mid_attn.fn.fn.to_q.weight" in sd:
        x = _from_tokens(self_attention(_to_tokens(x), sd, "mid_attn.fn.fn"), x.shape) + x
    x = resnet_block(x, sd, "mid_block2", t, c)
    skip = lambda x: torch.cat((x, hiddens.pop() * SKIP_SCALE), dim=1)
    for L in range(n_levels):
        x = skip(x)
        x = resnet_block(x, sd, f"ups.{L}.0", t, c)
        for k in range(_count(sd, f"ups.{L}.1." + "{}.block1.project.weight")):
            x = skip(x)
            x = resnet_block(x, sd, f"ups.{L}.1.{k}", t)
        if f"ups.{L}.2.attn.fn.to_q.weight" in sd:
            x = transformer_block(x, sd, f"ups.{L}.2")
        if f"ups.{L}.3.1.weight" in sd:
            x = F.interpolate(x, scale_factor=2, mode='nearest')
            x = _conv(x, sd, f"ups.{L}.3.1", padding=1)
    if "final_res_block.block1.project.weight" in sd:
        x = resnet_block(x, sd, "final_res_block", t)
    return _conv(x, sd, "final_conv", padding=1)


def unet_forward_with_cond_scale(sd: SD, x, time, cond_scale: float = 1., **kw):
    """minimagen/Unet.py:474-506."""
    logits = unet_forward(sd, x, time, cond_drop_prob=0., **kw)
    if cond_scale == 1:
        return logits
    null_logits = unet_forward(sd, x, time, cond_drop_prob=1., **kw)
    return null_logits + (logits - null_logits) * cond_scale


# ----------------------------------------------------------------------------- schedule
class Schedule:
    """minimagen/diffusion_model.py:13-66: linear beta schedule, all tables built in fp64, stored fp32."""

    def __init__(self, timesteps: int):
        assert not timesteps < 20
        self.num_timesteps = timesteps
        scale = 1000 / timesteps
        betas = torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)
        alphas = 1. - betas
        ac = torch.cumprod(alphas, dim=0)
        ac_prev = F.pad(ac[:-1], (1, 0), value=1.)
        f32 = lambda v: v.to(torch.float32)
        self.betas = f32(betas)
        self.alphas_cumprod = f32(ac)
        self.alphas_cumprod_prev = f32(ac_prev)
        self.sqrt_alphas_cumprod = f32(torch.sqrt(ac))
        self.sqrt_one_minus_alphas_cumprod = f32(torch.sqrt(1. - ac))
        self.log_one_minus_alphas_cumprod = f32(torch.log(1. - ac))
        self.sqrt_recip_alphas_cumprod = f32(torch.sqrt(1. / ac))
        self.sqrt_recipm1_alphas_cumprod = f32(torch.sqrt(1. / ac - 1))
        pv = betas * (1. - ac_prev) / (1. - ac)
        self.posterior_variance = f32(pv)
        self.posterior_log_variance_clipped = f32(torch.log(pv.clamp(min=1e-20)))
        self.posterior_mean_coef1 = f32(betas * torch.sqrt(ac_prev) / (1. - ac))
        self.posterior_mean_coef2 = f32((1. - ac_prev) * torch.sqrt(alphas) / (1. - ac))

    def get_times(self, batch, noise_level):
        """diffusion_model.py:68-69."""
        return torch.full((batch,), int(self.num_timesteps * noise_level), dtype=torch.long)

    def sampling_timesteps(self):
        """diffusion_model.py:81-87 (values only; every batch row is the same)."""
        return list(reversed(range(self.num_timesteps)))

    def q_sample(self, x_start, t: int, noise):
        """diffusion_model.py:142-147 (uniform t over the batch)."""
        return self.sqrt_alphas_cumprod[t] * x_start + self.sqrt_one_minus_alphas_cumprod[t] * noise

    def predict_start_from_noise(self, x_t, t: int, noise):
        """diffusion_model.py:159-162."""
        return self.sqrt_recip_alphas_cumprod[t] * x_t - self.sqrt_recipm1_alphas_cumprod[t] * noise

    def q_posterior(self, x_start, x_t, t: int):
        """diffusion_model.py:118-125."""
        mean = self.posterior_mean_coef1[t] * x_start + self.posterior_mean_coef2[t] * x_t
        return mean, self.posterior_variance[t], self.posterior_log_variance_clipped[t]


# ----------------------------------------------------------------------------- dynamic threshold
def quantile_rank(n: int, q: float):
    """torch.quantile's rank arithmetic, done in the INPUT dtype (fp32): SURVEY.md Appendix B-1."""
    rank = np.float32(q) * np.float32(n - 1)
    lo = int(np.floor(rank))
    w = np.float32(rank - np.float32(lo))
    return lo, w


def dynamic_threshold_quantile(absx: torch.Tensor, q: float):
    """Per-row q-quantile of non-negative fp32 rows (b, n), linear interpolation,
    as minimagen/Imagen.py:313-317 gets it from torch.quantile.
    Returns (s, v_lo, v_hi, lo, w): v_lo/v_hi are the two selected order statistics."""
    b, n = absx.shape
    lo, w = quantile_rank(n, q)
    srt = np.sort(absx.numpy(), axis=-1)
    v_lo = srt[:, lo]
    v_hi = srt[:, min(lo + 1, n - 1)]
    s = lerp_f32(v_lo, v_hi, w)
    return torch.from_numpy(s), torch.from_numpy(v_lo.copy()), torch.from_numpy(v_hi.copy()), lo, w


def lerp_f32(a: np.ndarray, b: np.ndarray, w: np.float32) -> np.ndarray:
    """ATen lerp (aten/src/ATen/native/Lerp.h): w<0.5 ? a + w*(b-a) : b - (b-a)*(1-w), with the
    multiply-add fused (vec::fmadd / compiler contraction).  Emulated with an exact fp64 product."""
    a32, b32 = a.astype(np.float32), b.astype(np.float32)
    d = (b32 - a32).astype(np.float32)
    if abs(float(w)) < 0.5:
        r = np.float64(w) * d.astype(np.float64) + a32.astype(np.float64)
    else:
        r = b32.astype(np.float64) - d.astype(np.float64) * np.float64(np.float32(1) - w)
    return r.astype(np.float32)


# ----------------------------------------------------------------------------- sampler
def p_mean_variance(sd: SD, sched: Schedule, x, t: int, *, pred=None, percentile=0.9, **unet_kw):
    """minimagen/Imagen.py:261-326 (uniform t).  Returns (mean, logvar, aux)."""
    b = x.shape[0]
    if pred is None:
        cond_scale = unet_kw.pop("cond_scale", 1.)
        pred = unet_forward_with_cond_scale(sd, x, torch.full((b,), t, dtype=torch.long), cond_scale=cond_scale, **unet_kw)
    x_start = sched.predict_start_from_noise(x, t, pred)
    s, v_lo, v_hi, lo, w = dynamic_threshold_quantile(x_start.reshape(b, -1).abs(), percentile)
    s = s.clamp(min=1.).reshape(b, 1, 1, 1)
    x_start_c = x_start.clamp(-s, s) / s
    mean, _, logvar = sched.q_posterior(x_start_c, x, t)
    return mean, logvar, dict(pred=pred, x_start=x_start, s=s.reshape(b), v_lo=v_lo, v_hi=v_hi, lo=lo, w=w)


def p_sample(sd: SD, sched: Schedule, x, t: int, noise, **kw):
    """minimagen/Imagen.py:329-370."""
    mean, logvar, aux = p_mean_variance(sd, sched, x, t, **kw)
    nonzero = torch.tensor(0. if t == 0 else 1.)
    return mean + nonzero * (0.5 * logvar).exp() * noise, aux


def p_sample_loop(sd: SD, sched: Schedule, shape, randn: Callable, *, lowres_cond_img=None, **kw):
    """minimagen/Imagen.py:373-420.  ``randn(shape)`` supplies noise in the reference's draw order."""
    if lowres_cond_img is not None:
        lowres_cond_img = lowres_cond_img * 2 - 1          # normalised AFTER noising (quirk B-6)
    img = randn(shape)
    for t in sched.sampling_timesteps():
        kw_t = dict(kw)
        # draw order: the U-Net runs first, then randn_like(x)
        mean, logvar, _ = p_mean_variance(sd, sched, img, t, lowres_cond_img=lowres_cond_img, **kw_t)
        noise = randn(shape)
        nonzero = torch.tensor(0. if t == 0 else 1.)
        img = mean + nonzero * (0.5 * logvar).exp() * noise
    img = img.clamp(-1., 1.)
    return (img + 1) * 0.5


def sample(unet_sds: Sequence[SD], image_sizes: Sequence[int], timesteps: int, *, text_embeds, text_masks=None,
           cond_scale: float = 1., lowres_sample_noise_level: float = 0.2, percentile: float = 0.9,
           channels: int = 3, randn: Callable):
    """minimagen/Imagen.py:424-510 (text_embeds path), one schedule per U-Net plus the low-res schedule (:75-78)."""
    b = text_embeds.shape[0]
    scheds = [Schedule(timesteps) for _ in unet_sds]
    lowres_sched = Schedule(timesteps)
    img = None
    for sd, size, sched in zip(unet_sds, image_sizes, scheds):
        lowres_cond_img = lowres_noise_times = None
        if "to_lowres_time_hiddens.1.weight" in sd:
            lowres_noise_times = lowres_sched.get_times(b, lowres_sample_noise_level)
            lowres_cond_img = resize_restated.resize(img, scale_factors=size / img.shape[-1], pad_mode='reflect') \
                if img.shape[-1] != size else img
            lowres_cond_img = lowres_sched.q_sample(lowres_cond_img, int(lowres_noise_times[0]), randn(lowres_cond_img.shape))
        img = p_sample_loop(sd, sched, (b, channels, size, size), randn,
                            text_embeds=text_embeds, text_mask=text_masks, cond_scale=cond_scale,
                            lowres_cond_img=lowres_cond_img, lowres_noise_times=lowres_noise_times,
                            percentile=percentile)
    return img


def make_randn(seed: int) -> Callable:
    gen = torch.Generator(device="cpu").manual_seed(seed)
    return lambda shape: torch.randn(tuple(shape), generator=gen, dtype=torch.float32)


# ----------------------------------------------------------------------------- synthetic inputs
def synthetic_text(batch: int, length: int = 64, dim: int = 512, seed: int = 7):
    """embeds = randn(B,L,dim, seed 7); row r keeps L - (r mod 24) leading tokens; masked rows zeroed (t5.py:82)."""
    gen = torch.Generator(device="cpu").manual_seed(seed)
    emb = torch.randn(batch, length, dim, generator=gen)
    keep = torch.tensor([max(1, length - (r % 24)) for r in range(batch)])
    mask = torch.arange(length)[None, :] < keep[:, None]
    emb = emb.masked_fill(~mask[:, :, None], 0.)
    return emb, mask
