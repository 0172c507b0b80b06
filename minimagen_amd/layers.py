"""The reference's layers (minimagen/layers.py) under the reference's names and state-dict keys.

Two execution paths share these modules' parameters:

* **sampling / inference** (the hot path): ``Unet.forward`` hands the whole module tree to ``minimagen_amd.engine``, which runs every
  layer as hand-written HIP kernels -- the ``forward`` methods below are NOT on that path;
* **training** (``Imagen.forward`` -> ``Unet.forward`` in train mode with autograd on): the ``forward`` methods below, differentiable
  torch ops on whatever device the parameters live on -- they state each layer's arithmetic once more in the reference's own order of
  operations, so the paths can be tested against each other.  On the GPU ``Block`` (and the plain 3x3 convs / CrossEmbedLayer, see
  ``Unet._forward_train``) switch to ``minimagen_amd.train_ops``: HIP kernels forward AND backward; ``CrossAttention`` to its folded form.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

from . import train_ops
from .helpers import default, exists


def _tokens(x: torch.Tensor) -> torch.Tensor:
    """'b c h w -> b (h w) c'"""
    return x.flatten(2).transpose(1, 2)


def _image(t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """'b (h w) c -> b c h w'"""
    return t.transpose(1, 2).reshape(like.shape[0], -1, *like.shape[2:])


class _Container(nn.Module):
    pass


class Identity(_Container):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x, *args, **kwargs):
        return x


class LayerNorm(_Container):
    """layers.py:333-343 -- gamma parameter, beta (zero) persistent buffer."""

    def __init__(self, dim: int):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer('beta', torch.zeros(dim))

    def forward(self, x):
        if torch.is_grad_enabled() and train_ops.layer_norm_supported(x, self.gamma):       # training on the device: train_ln.hip forward + backward
            return train_ops.layer_norm(x, self.gamma, self.beta)
        return F.layer_norm(x, x.shape[-1:], self.gamma, self.beta)


class AffineLayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters, same state-dict keys) whose training forward / backward runs on the HIP kernels (train_ops.layer_norm)"""

    def forward(self, x):
        if torch.is_grad_enabled() and self.elementwise_affine and len(self.normalized_shape) == 1 and train_ops.layer_norm_supported(x, self.weight):
            return train_ops.layer_norm(x, self.weight, self.bias, self.eps)
        return super().forward(x)


class ChanLayerNorm(_Container):
    """layers.py:164-177"""

    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))

    def forward(self, x):
        var = torch.var(x, dim=1, unbiased=False, keepdim=True)
        mean = torch.mean(x, dim=1, keepdim=True)
        return (x - mean) / (var + self.eps).sqrt() * self.g


class EinopsToAndFrom(_Container):
    """einops_exts.torch.EinopsToAndFrom: the wrapped module must be the attribute ``fn`` (state-dict keys)."""

    def __init__(self, from_einops: str, to_einops: str, fn: nn.Module):
        super().__init__()
        self.from_einops, self.to_einops = from_einops, to_einops
        self.fn = fn

    def forward(self, x, **kwargs):
        """only the 'b c h w' <-> 'b (h w) c' pair is used (layers.py:398, Unet.py:273)"""
        return _image(self.fn(_tokens(x), **kwargs), x)


class Residual(_Container):
    """layers.py:359-368"""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(x, **kwargs) + x


class Parallel(_Container):
    """layers.py:346-356"""

    def __init__(self, *fns):
        super().__init__()
        self.fns = nn.ModuleList(fns)

    def forward(self, x):
        if torch.is_grad_enabled() and train_ops.active(x):          # training on the device: every member on the HIP conv kernels where it fits
            one = lambda fn: (train_ops.conv3x3_forward(fn, x) if train_ops.is_plain_conv3x3(fn, x) else
                              (train_ops.conv1x1_forward(fn, x) if train_ops.is_conv1x1(fn, x) else fn(x)))
            return sum(one(fn) for fn in self.fns)
        return sum(fn(x) for fn in self.fns)


class SinusoidalPosEmb(_Container):
    """layers.py:442-465"""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        half = self.dim // 2
        freq = torch.exp(torch.arange(half, device=x.device) * -(math.log(10000) / (half - 1)))
        arg = x[:, None] * freq[None, :]
        return torch.cat((arg.sin(), arg.cos()), dim=-1)


class Attention(_Container):
    """layers.py:14-50 (multi-query self-attention)"""

    def __init__(self, dim: int, *, dim_head: int = 64, heads: int = 8, context_dim: int = None):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        inner_dim = dim_head * heads
        self.norm = LayerNorm(dim)
        self.null_kv = nn.Parameter(torch.randn(2, dim_head))
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim, dim_head * 2, bias=False)
        self.to_context = nn.Sequential(AffineLayerNorm(context_dim), nn.Linear(context_dim, dim_head * 2)) if exists(context_dim) else None
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim, bias=False), LayerNorm(dim))

    def forward(self, x, context=None, mask=None, attn_bias=None):
        """layers.py:52-104: one key / value head shared by all query heads, a learned null key / value in front"""
        b, n, _ = x.shape
        x = self.norm(x)
        q = self.to_q(x).reshape(b, n, self.heads, -1).transpose(1, 2) * self.scale
        k, v = self.to_kv(x).chunk(2, dim=-1)
        if exists(context) and exists(self.to_context):
            ck, cv = self.to_context(context).chunk(2, dim=-1)
            k, v = torch.cat((ck, k), dim=-2), torch.cat((cv, v), dim=-2)
        nk, nv = (t.expand(b, 1, -1) for t in self.null_kv.unbind(dim=-2))
        k, v = torch.cat((nk, k), dim=-2), torch.cat((nv, v), dim=-2)
        sim = torch.einsum('bhid,bjd->bhij', q, k)
        if exists(attn_bias):
            sim = sim + attn_bias
        if exists(mask):
            sim = sim.masked_fill(~F.pad(mask, (1, 0), value=True)[:, None, None, :], -torch.finfo(sim.dtype).max)
        attn = sim.softmax(dim=-1, dtype=torch.float32)
        out = torch.einsum('bhij,bjd->bhid', attn, v).transpose(1, 2).reshape(b, n, -1)
        return self.to_out(out)


class Block(_Container):
    """layers.py:107-129"""

    def __init__(self, dim: int, dim_out: int, groups: int = 8, norm: bool = True):
        super().__init__()
        self.groupnorm = nn.GroupNorm(groups, dim) if norm else Identity()
        self.activation = nn.SiLU()
        self.project = nn.Conv2d(dim, dim_out, 3, padding=1)

    def forward(self, x, scale_shift=None, residual=None):
        """``residual`` (not in the reference's signature): added to the output -- ResnetBlock hands its skip term here so that the device training
        path can add it in the conv's epilogue"""
        if torch.is_grad_enabled() and train_ops.active(x) and train_ops.block_supported(self, x):        # training on the device: fused HIP forward, HIP dgrad / wgrad
            return train_ops.block_forward(self, x, scale_shift, residual)
        x = self.groupnorm(x)
        if exists(scale_shift):
            scale, shift = scale_shift
            x = x * (scale + 1) + shift
        out = self.project(self.activation(x))
        return out if residual is None else out + residual


def ChanFeedForward(dim: int, mult: int = 2) -> nn.Sequential:
    """layers.py:148-161"""
    hidden_dim = int(dim * mult)
    return nn.Sequential(
        ChanLayerNorm(dim),
        nn.Conv2d(dim, hidden_dim, 1, bias=False),
        nn.GELU(),
        ChanLayerNorm(hidden_dim),
        nn.Conv2d(hidden_dim, dim, 1, bias=False)
    )


class CrossAttention(_Container):
    """layers.py:180-218"""

    def __init__(self, dim: int, *, context_dim: int = None, dim_head: int = 64, heads: int = 8, norm_context: bool = False):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        inner_dim = dim_head * heads
        context_dim = default(context_dim, dim)
        self.norm = LayerNorm(dim)
        self.norm_context = LayerNorm(context_dim) if norm_context else Identity()
        self.null_kv = nn.Parameter(torch.randn(2, dim_head))
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(context_dim, inner_dim * 2, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim, bias=False), LayerNorm(dim))

    def _forward_folded(self, x, context, mask=None):
        """The sampler's fold (packing.fold_cross_attention) as differentiable torch ops, for the training graph on the
        GPU when the token side is narrower than a head (C < dim_head: the BASELINE U-Nets, C = 16 against 64): the token side never
        leaves its C channels -- sim_h = x . (s Wq_h^T k_h^T), out = sum_h softmax(sim_h) . (v_h Wo_h^T) -- so the [tokens x heads*dim_head]
        query and pre-projection tensors (268 MB each at 64 x 64, B = 32) and three quarters of the multiply-adds disappear.  Equal to
        ``forward`` in real arithmetic; rounding differs at the 1e-6 level."""
        b, n, Cc = x.shape
        H, D = self.heads, self.dim_head
        x, context = self.norm(x), self.norm_context(context)
        k, v = (t.reshape(b, -1, H, D).transpose(1, 2) for t in self.to_kv(context).chunk(2, dim=-1))          # [b, H, J0, D]
        nk, nv = (t.expand(b, H, 1, -1) for t in self.null_kv.unbind(dim=-2))
        k, v = torch.cat((nk, k), dim=-2), torch.cat((nv, v), dim=-2)                                           # [b, H, J, D]
        J = k.shape[2]
        kf = torch.einsum('bhjd,hdc->bhjc', k, self.to_q.weight.reshape(H, D, Cc)) * self.scale                 # keys in token-channel space
        vf = torch.einsum('bhjd,chd->bhjc', v, self.to_out[0].weight.reshape(Cc, H, D))                         # values through to_out.0
        if train_ops.folded_attention_supported(x, kf):            # HIP forward + backward, no [tokens x heads x context] tensor at all
            full = F.pad(mask, (1, 0), value=True) if exists(mask) else None
            return self.to_out[1](train_ops.folded_attention(x, kf, vf, full))
        sim = torch.bmm(x, kf.reshape(b, H * J, Cc).transpose(1, 2)).reshape(b, n, H, J)
        if exists(mask):
            sim = sim.masked_fill(~F.pad(mask, (1, 0), value=True)[:, None, None, :], -torch.finfo(sim.dtype).max)
        attn = sim.softmax(dim=-1, dtype=torch.float32)
        out = torch.bmm(attn.reshape(b, n, H * J), vf.reshape(b, H * J, Cc))
        return self.to_out[1](out)

    def forward(self, x, context, mask=None):
        """layers.py:220-251"""
        b, n, _ = x.shape
        if torch.is_grad_enabled() and train_ops.ENABLED and (x.is_cuda or train_ops.FORCE) and x.shape[-1] < self.dim_head:
            return self._forward_folded(x, context, mask)
        x, context = self.norm(x), self.norm_context(context)
        heads = lambda t: t.reshape(b, t.shape[1], self.heads, -1).transpose(1, 2)
        q = heads(self.to_q(x)) * self.scale
        k, v = (heads(t) for t in self.to_kv(context).chunk(2, dim=-1))
        nk, nv = (t.expand(b, self.heads, 1, -1) for t in self.null_kv.unbind(dim=-2))
        k, v = torch.cat((nk, k), dim=-2), torch.cat((nv, v), dim=-2)
        sim = torch.einsum('bhid,bhjd->bhij', q, k)
        if exists(mask):
            sim = sim.masked_fill(~F.pad(mask, (1, 0), value=True)[:, None, None, :], -torch.finfo(sim.dtype).max)
        attn = sim.softmax(dim=-1, dtype=torch.float32)
        out = torch.einsum('bhij,bhjd->bhid', attn, v).transpose(1, 2).reshape(b, n, -1)
        return self.to_out(out)


class CrossEmbedLayer(_Container):
    """layers.py:254-300"""

    def __init__(self, dim_in: int, kernel_sizes, dim_out: int = None, stride: int = 2):
        super().__init__()
        assert all([*map(lambda t: (t % 2) == (stride % 2), kernel_sizes)])
        dim_out = default(dim_out, dim_in)
        kernel_sizes = sorted(kernel_sizes)
        num_scales = len(kernel_sizes)
        dim_scales = [int(dim_out / (2 ** i)) for i in range(1, num_scales)]
        dim_scales = [*dim_scales, dim_out - sum(dim_scales)]
        self.kernel_sizes, self.dim_scales, self.stride = kernel_sizes, dim_scales, stride
        self.convs = nn.ModuleList([])
        for kernel, dim_scale in zip(kernel_sizes, dim_scales):
            self.convs.append(nn.Conv2d(dim_in, dim_scale, kernel, stride=stride, padding=(kernel - stride) // 2))

    def forward(self, x):
        return torch.cat([conv(x) for conv in self.convs], dim=1)


def Downsample(dim: int, dim_out: int = None) -> nn.Conv2d:
    """layers.py:308-319"""
    dim_out = default(dim_out, dim)
    return nn.Conv2d(dim, dim_out, kernel_size=4, stride=2, padding=1)


class ResnetBlock(_Container):
    """layers.py:371-415"""

    def __init__(self, dim: int, dim_out: int, *, cond_dim: int = None, time_cond_dim: int = None, groups: int = 8):
        super().__init__()
        self.time_mlp = None
        if exists(time_cond_dim):
            self.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_cond_dim, dim_out * 2))
        self.cross_attn = None
        if exists(cond_dim):
            self.cross_attn = EinopsToAndFrom('b c h w', 'b (h w) c', CrossAttention(dim=dim_out, context_dim=cond_dim))
        self.block1 = Block(dim, dim_out, groups=groups)
        self.block2 = Block(dim_out, dim_out, groups=groups)
        self.res_conv = nn.Conv2d(dim, dim_out, 1) if dim != dim_out else Identity()
        self.groups = groups

    def forward(self, x, time_emb=None, cond=None):
        """layers.py:417-439"""
        scale_shift = None
        pre = self.__dict__.pop("_mi_scale_shift", None)       # Unet._forward_train on the device: every block's time MLP as ONE stacked linear layer
        if exists(self.time_mlp) and exists(time_emb):
            ss = pre if pre is not None else self.time_mlp(time_emb)
            scale_shift = ss[:, :, None, None].chunk(2, dim=1)
        h = self.block1(x)
        if exists(self.cross_attn):
            assert exists(cond)
            h = self.cross_attn(h, context=cond) + h
        if torch.is_grad_enabled() and train_ops.active(x):     # training on the device: the skip term goes into block2's conv epilogue where it can
            res = train_ops.conv1x1_forward(self.res_conv, x) if train_ops.is_conv1x1(self.res_conv, x) else self.res_conv(x)
            return self.block2(h, scale_shift=scale_shift, residual=res.contiguous())
        h = self.block2(h, scale_shift=scale_shift)
        return h + self.res_conv(x)


class TransformerBlock(_Container):
    """layers.py:468-494"""

    def __init__(self, dim: int, *, heads: int = 8, dim_head: int = 32, ff_mult: int = 2, context_dim: int = None):
        super().__init__()
        self.attn = EinopsToAndFrom('b c h w', 'b (h w) c', Attention(dim=dim, heads=heads, dim_head=dim_head, context_dim=context_dim))
        self.ff = ChanFeedForward(dim=dim, mult=ff_mult)

    def forward(self, x, context=None):
        x = self.attn(x, context=context) + x
        return self.ff(x) + x


def Upsample(dim: int, dim_out: int = None) -> nn.Sequential:
    """layers.py:502-515"""
    dim_out = default(dim_out, dim)
    return nn.Sequential(nn.Upsample(scale_factor=2, mode='nearest'), nn.Conv2d(dim, dim_out, 3, padding=1))
