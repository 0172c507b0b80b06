"""``Unet`` with the reference's constructor, ``forward`` / ``forward_with_cond_scale`` signatures and
state-dict layout (minimagen/Unet.py), executed by hand-written HIP kernels on MI355X."""
from __future__ import annotations

from typing import Union

import torch
from torch import nn

from .helpers import default, exists, cast_tuple, prob_mask_like
from .layers import (Attention, CrossEmbedLayer, Downsample, EinopsToAndFrom, Identity, Parallel, Residual, ResnetBlock,
                     SinusoidalPosEmb, TransformerBlock, Upsample)
from .t5 import get_encoded_dim


class _Rearrange(nn.Module):
    """placeholder for einops' Rearrange('b (r d) -> b r d') (no parameters; keeps the Sequential indices)"""

    def __init__(self, r):
        super().__init__()
        self.r = r


class Unet(nn.Module):
    """Denoising U-Net (minimagen/Unet.py:25-472).  Parameters live in torch modules named exactly as in
    the reference so that ``load_state_dict`` of a reference checkpoint works; the forward pass is a
    sequence of HIP kernel launches built by :class:`minimagen_amd.engine.UnetEngine`."""

    def __init__(
            self,
            *,
            dim: int = 128,
            dim_mults: tuple = (1, 2, 4),
            channels: int = 3,
            channels_out: int = None,
            cond_dim: int = None,
            text_embed_dim=get_encoded_dim('t5_small'),
            num_resnet_blocks: Union[int, tuple] = 1,
            layer_attns: Union[bool, tuple] = True,
            layer_cross_attns: Union[bool, tuple] = True,
            attn_heads: int = 8,
            lowres_cond: bool = False,
            memory_efficient: bool = False,
            attend_at_middle: bool = False
    ):
        super().__init__()
        # constructor kwargs, for _cast_model_parameters (Unet.py:81-83)
        self._locals = dict(dim=dim, dim_mults=dim_mults, channels=channels, channels_out=channels_out, cond_dim=cond_dim,
                            text_embed_dim=text_embed_dim, num_resnet_blocks=num_resnet_blocks, layer_attns=layer_attns,
                            layer_cross_attns=layer_cross_attns, attn_heads=attn_heads, lowres_cond=lowres_cond,
                            memory_efficient=memory_efficient, attend_at_middle=attend_at_middle)

        ATTN_DIM_HEAD = 64      # Unet.py:86
        NUM_TIME_TOKENS = 2     # Unet.py:87
        RESNET_GROUPS = 8       # Unet.py:88
        self.num_time_tokens = NUM_TIME_TOKENS
        self.attn_dim_head = ATTN_DIM_HEAD
        self.attn_heads = attn_heads
        self.dim = dim

        cond_dim = default(cond_dim, dim)
        self.cond_dim = cond_dim
        time_cond_dim = dim * 4 * (2 if lowres_cond else 1)
        self.time_cond_dim = time_cond_dim

        self.to_time_hiddens = nn.Sequential(SinusoidalPosEmb(dim), nn.Linear(dim, time_cond_dim), nn.SiLU())
        self.to_time_cond = nn.Sequential(nn.Linear(time_cond_dim, time_cond_dim))
        self.to_time_tokens = nn.Sequential(nn.Linear(time_cond_dim, cond_dim * NUM_TIME_TOKENS), _Rearrange(NUM_TIME_TOKENS))

        self.lowres_cond = lowres_cond
        if lowres_cond:
            self.to_lowres_time_hiddens = nn.Sequential(SinusoidalPosEmb(dim), nn.Linear(dim, time_cond_dim), nn.SiLU())
            self.to_lowres_time_cond = nn.Sequential(nn.Linear(time_cond_dim, time_cond_dim))
            self.to_lowres_time_tokens = nn.Sequential(nn.Linear(time_cond_dim, cond_dim * NUM_TIME_TOKENS), _Rearrange(NUM_TIME_TOKENS))

        self.norm_cond = nn.LayerNorm(cond_dim)
        self.text_embed_dim = text_embed_dim
        self.text_to_cond = nn.Linear(self.text_embed_dim, cond_dim)
        max_text_len = 256
        self.max_text_len = max_text_len
        self.null_text_embed = nn.Parameter(torch.randn(1, max_text_len, cond_dim))
        self.null_text_hidden = nn.Parameter(torch.randn(1, time_cond_dim))
        self.to_text_non_attn_cond = nn.Sequential(
            nn.LayerNorm(cond_dim),
            nn.Linear(cond_dim, time_cond_dim),
            nn.SiLU(),
            nn.Linear(time_cond_dim, time_cond_dim)
        )

        self.channels = channels
        self.channels_out = default(channels_out, channels)
        self.init_conv = CrossEmbedLayer(channels if not lowres_cond else channels * 2, dim_out=dim, kernel_sizes=(3, 7, 15), stride=1)

        dims = [dim, *map(lambda m: dim * m, dim_mults)]
        in_out = list(zip(dims[:-1], dims[1:]))
        num_resolutions = len(in_out)
        num_resnet_blocks = cast_tuple(num_resnet_blocks, num_resolutions)
        resnet_groups = cast_tuple(RESNET_GROUPS, num_resolutions)
        layer_attns = cast_tuple(layer_attns, num_resolutions)
        layer_cross_attns = cast_tuple(layer_cross_attns, num_resolutions)
        assert all([layers == num_resolutions for layers in list(map(len, (resnet_groups, layer_attns, layer_cross_attns)))])

        self.skip_connect_scale = 2 ** -0.5
        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        layer_params = [num_resnet_blocks, resnet_groups, layer_attns, layer_cross_attns]
        reversed_layer_params = list(map(reversed, layer_params))
        skip_connect_dims = []

        for ind, ((dim_in, dim_out), layer_num_resnet_blocks, groups, layer_attn, layer_cross_attn) in enumerate(zip(in_out, *layer_params)):
            is_last = ind == (num_resolutions - 1)
            layer_cond_dim = cond_dim if layer_cross_attn else None
            transformer_block_klass = TransformerBlock if layer_attn else Identity
            current_dim = dim_in
            pre_downsample = None
            if memory_efficient:
                pre_downsample = Downsample(dim_in, dim_out)
                current_dim = dim_out
            skip_connect_dims.append(current_dim)
            post_downsample = None
            if not memory_efficient:
                post_downsample = Downsample(current_dim, dim_out) if not is_last else Parallel(
                    nn.Conv2d(dim_in, dim_out, 3, padding=1), nn.Conv2d(dim_in, dim_out, 1))
            self.downs.append(nn.ModuleList([
                pre_downsample,
                ResnetBlock(current_dim, current_dim, cond_dim=layer_cond_dim, time_cond_dim=time_cond_dim, groups=groups),
                nn.ModuleList([ResnetBlock(current_dim, current_dim, time_cond_dim=time_cond_dim, groups=groups)
                               for _ in range(layer_num_resnet_blocks)]),
                transformer_block_klass(dim=current_dim, heads=attn_heads, dim_head=ATTN_DIM_HEAD),
                post_downsample,
            ]))

        mid_dim = dims[-1]
        self.mid_block1 = ResnetBlock(mid_dim, mid_dim, cond_dim=cond_dim, time_cond_dim=time_cond_dim, groups=resnet_groups[-1])
        self.mid_attn = EinopsToAndFrom('b c h w', 'b (h w) c',
                                        Residual(Attention(mid_dim, heads=attn_heads, dim_head=ATTN_DIM_HEAD))) if attend_at_middle else None
        self.mid_block2 = ResnetBlock(mid_dim, mid_dim, cond_dim=cond_dim, time_cond_dim=time_cond_dim, groups=resnet_groups[-1])

        for ind, ((dim_in, dim_out), layer_num_resnet_blocks, groups, layer_attn, layer_cross_attn) in enumerate(
                zip(reversed(in_out), *reversed_layer_params)):
            is_last = ind == (num_resolutions - 1)
            layer_cond_dim = cond_dim if layer_cross_attn else None
            transformer_block_klass = TransformerBlock if layer_attn else Identity
            skip_connect_dim = skip_connect_dims.pop()
            self.ups.append(nn.ModuleList([
                ResnetBlock(dim_out + skip_connect_dim, dim_out, cond_dim=layer_cond_dim, time_cond_dim=time_cond_dim, groups=groups),
                nn.ModuleList([ResnetBlock(dim_out + skip_connect_dim, dim_out, time_cond_dim=time_cond_dim, groups=groups)
                               for _ in range(layer_num_resnet_blocks)]),
                transformer_block_klass(dim=dim_out, heads=attn_heads, dim_head=ATTN_DIM_HEAD),
                Upsample(dim_out, dim_in) if not is_last or memory_efficient else Identity()
            ]))

        self.init_conv_to_final_conv_residual = False       # Unet.py:91
        self.final_res_block = ResnetBlock(dim, dim, time_cond_dim=time_cond_dim, groups=resnet_groups[0])
        self.final_conv = nn.Conv2d(dim, self.channels_out, 3, padding=3 // 2)

        self._engine = None

    # ------------------------------------------------------------------ reference API
    def _cast_model_parameters(self, *, lowres_cond, text_embed_dim, channels, channels_out):
        """Unet.py:332-353: re-instantiate (fresh weights) when any of the four settings differs."""
        if lowres_cond == self.lowres_cond and channels == self.channels and \
                text_embed_dim == self.text_embed_dim and channels_out == self.channels_out:
            return self
        updated_kwargs = dict(lowres_cond=lowres_cond, text_embed_dim=text_embed_dim, channels=channels, channels_out=channels_out)
        return self.__class__(**{**self._locals, **updated_kwargs})

    def engine(self):
        from .engine import UnetEngine
        if self._engine is None:
            object.__setattr__(self, '_engine', UnetEngine(self))
        return self._engine

    def forward(self, x: torch.Tensor, time: torch.Tensor, *, lowres_cond_img: torch.Tensor = None,
                lowres_noise_times: torch.Tensor = None, text_embeds: torch.Tensor = None, text_mask: torch.Tensor = None,
                cond_drop_prob: float = 0.) -> torch.Tensor:
        """Unet.py:355-472 (noise prediction)."""
        assert not (self.lowres_cond and not exists(lowres_cond_img)), 'low resolution conditioning image must be present'
        assert not (self.lowres_cond and not exists(lowres_noise_times)), 'low resolution conditioning noise time must be present'
        keep = prob_mask_like((x.shape[0],), 1 - cond_drop_prob, device='cpu')      # Unet.py:587
        with torch.no_grad():
            return self.engine().forward_once(x, time, lowres_cond_img=lowres_cond_img if self.lowres_cond else None,
                                              lowres_noise_times=lowres_noise_times if self.lowres_cond else None,
                                              text_embeds=text_embeds, text_mask=text_mask, keep=keep)

    def forward_with_cond_scale(self, *args, cond_scale: float = 1., **kwargs) -> torch.Tensor:
        """Unet.py:474-506: both guidance halves run as ONE batch of 2B rows through the engine."""
        if cond_scale == 1:
            return self.forward(*args, **kwargs)
        x, time = args
        assert not (self.lowres_cond and not exists(kwargs.get('lowres_cond_img'))), 'low resolution conditioning image must be present'
        assert not (self.lowres_cond and not exists(kwargs.get('lowres_noise_times'))), 'low resolution conditioning noise time must be present'
        with torch.no_grad():
            return self.engine().forward_once(x, time, lowres_cond_img=kwargs.get('lowres_cond_img') if self.lowres_cond else None,
                                              lowres_noise_times=kwargs.get('lowres_noise_times') if self.lowres_cond else None,
                                              text_embeds=kwargs.get('text_embeds'), text_mask=kwargs.get('text_mask'),
                                              keep=None, cond_scale=cond_scale)


class Base(Unet):
    """Unet.py:637-664"""
    defaults = dict(dim=512, dim_mults=(1, 2, 3, 4), num_resnet_blocks=3, layer_attns=(False, True, True, True),
                    layer_cross_attns=(False, True, True, True), memory_efficient=False)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **{**Base.defaults, **kwargs})


class Super(Unet):
    """Unet.py:667-692"""
    defaults = dict(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=(2, 4, 8, 8), layer_attns=(False, False, False, True),
                    layer_cross_attns=(False, False, False, True), memory_efficient=True)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **{**Super.defaults, **kwargs})


class BaseTest(Unet):
    """Unet.py:695-722.  NOTE: like the reference, the constructor merges ``Base.defaults`` (reference quirk,
    SURVEY.md section 4); the small configuration is ``BaseTest.defaults`` (what training.get_default_args reads)."""
    defaults = dict(dim=8, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=False, layer_cross_attns=False, memory_efficient=False)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **{**Base.defaults, **kwargs})


class SuperTest(Unet):
    """Unet.py:725-750 (same quirk with ``Super.defaults``)."""
    defaults = dict(dim=8, dim_mults=(1, 2), num_resnet_blocks=(1, 2), layer_attns=False, layer_cross_attns=False, memory_efficient=True)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **{**Super.defaults, **kwargs})
