"""Parameter containers with the reference's layer names (minimagen/layers.py).

These modules only HOLD parameters under the reference's state-dict keys and
carry the default initialisation of the torch modules the reference uses.  They have no compute of
their own: the arithmetic of each layer is a HIP kernel scheduled by ``minimagen_amd.engine`` from
``Unet.forward``.  Calling one directly is an error.
"""
from __future__ import annotations

import torch
from torch import nn

from .helpers import default, exists


class _Container(nn.Module):
    def forward(self, *args, **kwargs):
        raise RuntimeError(f"{type(self).__name__} is a parameter container; its arithmetic runs in the HIP engine via Unet.forward")


class Identity(_Container):
    def __init__(self, *args, **kwargs):
        super().__init__()


class LayerNorm(_Container):
    """layers.py:333-343 -- gamma parameter, beta (zero) persistent buffer."""

    def __init__(self, dim: int):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer('beta', torch.zeros(dim))


class ChanLayerNorm(_Container):
    """layers.py:164-177"""

    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))


class EinopsToAndFrom(_Container):
    """einops_exts.torch.EinopsToAndFrom: the wrapped module must be the attribute ``fn`` (state-dict keys)."""

    def __init__(self, from_einops: str, to_einops: str, fn: nn.Module):
        super().__init__()
        self.from_einops, self.to_einops = from_einops, to_einops
        self.fn = fn


class Residual(_Container):
    """layers.py:359-368"""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class Parallel(_Container):
    """layers.py:346-356"""

    def __init__(self, *fns):
        super().__init__()
        self.fns = nn.ModuleList(fns)


class SinusoidalPosEmb(_Container):
    """layers.py:442-465"""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim


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
        self.to_context = nn.Sequential(nn.LayerNorm(context_dim), nn.Linear(context_dim, dim_head * 2)) if exists(context_dim) else None
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim, bias=False), LayerNorm(dim))


class Block(_Container):
    """layers.py:107-129"""

    def __init__(self, dim: int, dim_out: int, groups: int = 8, norm: bool = True):
        super().__init__()
        self.groupnorm = nn.GroupNorm(groups, dim) if norm else Identity()
        self.activation = nn.SiLU()
        self.project = nn.Conv2d(dim, dim_out, 3, padding=1)


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


class TransformerBlock(_Container):
    """layers.py:468-494"""

    def __init__(self, dim: int, *, heads: int = 8, dim_head: int = 32, ff_mult: int = 2, context_dim: int = None):
        super().__init__()
        self.attn = EinopsToAndFrom('b c h w', 'b (h w) c', Attention(dim=dim, heads=heads, dim_head=dim_head, context_dim=context_dim))
        self.ff = ChanFeedForward(dim=dim, mult=ff_mult)


def Upsample(dim: int, dim_out: int = None) -> nn.Sequential:
    """layers.py:502-515"""
    dim_out = default(dim_out, dim)
    return nn.Sequential(nn.Upsample(scale_factor=2, mode='nearest'), nn.Conv2d(dim, dim_out, 3, padding=1))
