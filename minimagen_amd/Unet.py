"""``Unet`` with the reference's constructor, ``forward`` / ``forward_with_cond_scale`` signatures and
state-dict layout (minimagen/Unet.py), executed by hand-written HIP kernels on MI355X."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Union

import warnings

import torch
import torch.nn.functional as F
from torch import nn

from . import train_ops
from .helpers import default, exists, cast_tuple, prob_mask_like
from .layers import (AffineLayerNorm, Attention, CrossEmbedLayer, Downsample, EinopsToAndFrom, Identity, Parallel, Residual, ResnetBlock,
                     SinusoidalPosEmb, TransformerBlock, Upsample)
from .t5 import get_encoded_dim


class _Rearrange(nn.Module):
    """placeholder for einops' Rearrange('b (r d) -> b r d') (no parameters; keeps the Sequential indices)"""

    def __init__(self, r):
        super().__init__()
        self.r = r


ATTN_DIM_HEAD = 64      # Unet.py:86
NUM_TIME_TOKENS = 2     # Unet.py:87
RESNET_GROUPS = 8       # Unet.py:88
MAX_TEXT_LEN = 256      # Unet.py:145
CROSS_EMBED_KERNELS = (3, 7, 15)


@dataclass(frozen=True)
class LevelSpec:
    """One resolution level of the U-Net, the unit both halves of the trunk are generated from."""
    index: int
    width_in: int           # channels entering the level on the way down (= leaving it on the way up)
    width_out: int          # channels handed to the next deeper level
    res_blocks: int         # ResnetBlocks after the level's first one
    groups: int
    self_attention: bool    # TransformerBlock after the ResnetBlocks
    cross_attention: bool   # the level's first ResnetBlock attends to the conditioning tokens
    deepest: bool

    def trunk_width(self, memory_efficient: bool) -> int:
        """channels the level's ResnetBlocks run at on the way down: the down-sampling conv comes first when memory_efficient"""
        return self.width_out if memory_efficient else self.width_in


def level_plan(dim: int, dim_mults, num_resnet_blocks, layer_attns, layer_cross_attns) -> List[LevelSpec]:
    """The per-level table behind Unet.py:170-330: widths dim * (1, *dim_mults) chained level to level, the per-level knobs broadcast
    from scalars."""
    widths = [dim] + [dim * m for m in dim_mults]
    n = len(dim_mults)
    per_level = [cast_tuple(v, n) for v in (num_resnet_blocks, RESNET_GROUPS, layer_attns, layer_cross_attns)]
    assert all(len(v) == n for v in per_level), "per-level settings must have one entry per resolution"
    return [LevelSpec(i, widths[i], widths[i + 1], per_level[0][i], per_level[1][i], bool(per_level[2][i]), bool(per_level[3][i]), i == n - 1)
            for i in range(n)]


class Unet(nn.Module):
    """Denoising U-Net (minimagen/Unet.py:25-472).  Parameters live in torch modules named exactly as in
    the reference so that ``load_state_dict`` of a reference checkpoint works; the forward pass is a
    sequence of HIP kernel launches built by :class:`minimagen_amd.engine.UnetEngine`."""
    _warned_torch_path = False

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
        self._locals = {k: v for k, v in locals().items() if k not in ('self', '__class__')}
        self._engine = None

        # ---- scalars the engine reads
        self.dim, self.channels, self.channels_out = dim, channels, default(channels_out, channels)
        self.cond_dim = default(cond_dim, dim)
        self.time_cond_dim = dim * 4 * (2 if lowres_cond else 1)
        self.text_embed_dim, self.max_text_len = text_embed_dim, MAX_TEXT_LEN
        self.num_time_tokens, self.attn_dim_head, self.attn_heads = NUM_TIME_TOKENS, ATTN_DIM_HEAD, attn_heads
        self.lowres_cond, self.memory_efficient = lowres_cond, memory_efficient
        self.skip_connect_scale = 2 ** -0.5
        self.init_conv_to_final_conv_residual = False       # Unet.py:91
        self.levels = level_plan(dim, dim_mults, num_resnet_blocks, layer_attns, layer_cross_attns)

        # ---- conditioning path (Unet.py:104-168): one (hiddens, cond, tokens) trio per timestep input, then the text branch
        for prefix in ("",) + (("lowres_",) if lowres_cond else ()):
            for name, module in self._time_trio().items():
                setattr(self, f"to_{prefix}{name}", module)
        self.norm_cond = AffineLayerNorm(self.cond_dim)
        self.text_to_cond = nn.Linear(text_embed_dim, self.cond_dim)
        self.null_text_embed = nn.Parameter(torch.randn(1, MAX_TEXT_LEN, self.cond_dim))
        self.null_text_hidden = nn.Parameter(torch.randn(1, self.time_cond_dim))
        self.to_text_non_attn_cond = nn.Sequential(AffineLayerNorm(self.cond_dim), nn.Linear(self.cond_dim, self.time_cond_dim), nn.SiLU(),
                                                   nn.Linear(self.time_cond_dim, self.time_cond_dim))

        # ---- trunk, generated from the level table: stem, the way down, the middle, the way up (mirror of the table), head
        self.init_conv = CrossEmbedLayer(channels * (2 if lowres_cond else 1), dim_out=dim, kernel_sizes=CROSS_EMBED_KERNELS, stride=1)
        self.downs = nn.ModuleList(self._down_level(lv) for lv in self.levels)
        self.ups = nn.ModuleList(self._up_level(lv) for lv in reversed(self.levels))
        bottom = self.levels[-1]
        self.mid_block1 = self._resblock(bottom.width_out, bottom.width_out, bottom.groups, cross_attention=True)
        self.mid_attn = EinopsToAndFrom('b c h w', 'b (h w) c', Residual(Attention(bottom.width_out, heads=attn_heads, dim_head=ATTN_DIM_HEAD))) \
            if attend_at_middle else None
        self.mid_block2 = self._resblock(bottom.width_out, bottom.width_out, bottom.groups, cross_attention=True)
        self.final_res_block = self._resblock(dim, dim, self.levels[0].groups, cross_attention=False)
        self.final_conv = nn.Conv2d(dim, self.channels_out, 3, padding=1)

    # ------------------------------------------------------------------ builders
    def _time_trio(self):
        tcd = self.time_cond_dim
        return dict(time_hiddens=nn.Sequential(SinusoidalPosEmb(self.dim), nn.Linear(self.dim, tcd), nn.SiLU()),
                    time_cond=nn.Sequential(nn.Linear(tcd, tcd)),
                    time_tokens=nn.Sequential(nn.Linear(tcd, self.cond_dim * NUM_TIME_TOKENS), _Rearrange(NUM_TIME_TOKENS)))

    def _resblock(self, width_in: int, width_out: int, groups: int, cross_attention: bool) -> ResnetBlock:
        return ResnetBlock(width_in, width_out, cond_dim=self.cond_dim if cross_attention else None, time_cond_dim=self.time_cond_dim, groups=groups)

    def _attention_slot(self, lv: LevelSpec, width: int) -> nn.Module:
        return TransformerBlock(dim=width, heads=self.attn_heads, dim_head=ATTN_DIM_HEAD) if lv.self_attention else Identity()

    def _down_level(self, lv: LevelSpec) -> nn.ModuleList:
        """[pre-downsample | None, first ResnetBlock, ResnetBlocks, attention slot, post-downsample | None] (Unet.py:196-262): the
        resolution changes before the blocks when memory_efficient, after them otherwise (the deepest level then keeps its resolution
        and widens through Parallel(3x3, 1x1))"""
        w = lv.trunk_width(self.memory_efficient)
        before = Downsample(lv.width_in, lv.width_out) if self.memory_efficient else None
        after = None
        if not self.memory_efficient:
            after = Parallel(nn.Conv2d(lv.width_in, lv.width_out, 3, padding=1), nn.Conv2d(lv.width_in, lv.width_out, 1)) if lv.deepest \
                else Downsample(w, lv.width_out)
        return nn.ModuleList([before, self._resblock(w, w, lv.groups, lv.cross_attention),
                              nn.ModuleList(self._resblock(w, w, lv.groups, False) for _ in range(lv.res_blocks)),
                              self._attention_slot(lv, w), after])

    def _up_level(self, lv: LevelSpec) -> nn.ModuleList:
        """[first ResnetBlock, ResnetBlocks, attention slot, upsample | Identity] (Unet.py:283-323); every ResnetBlock consumes the
        running tensor concatenated with one skip tensor of the mirrored down level"""
        skip = lv.trunk_width(self.memory_efficient)
        w = lv.width_out
        keeps_resolution = lv.index == 0 and not self.memory_efficient          # the outermost level of a non-memory-efficient net
        return nn.ModuleList([self._resblock(w + skip, w, lv.groups, lv.cross_attention),
                              nn.ModuleList(self._resblock(w + skip, w, lv.groups, False) for _ in range(lv.res_blocks)),
                              self._attention_slot(lv, w),
                              Identity() if keeps_resolution else Upsample(w, lv.width_in)])

    # ------------------------------------------------------------------ packed-weight lifetime
    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        if self._engine is not None:
            self._engine.invalidate()
        train_ops.invalidate(self)
        return out

    def _apply(self, fn, *args, **kwargs):
        before = [t.data_ptr() for t in self.parameters()]
        out = super()._apply(fn, *args, **kwargs)           # .to() / .cuda() / .float(): the packed copies point at the old storage ...
        if before != [t.data_ptr() for t in self.parameters()]:
            if getattr(self, '_engine', None) is not None:
                self._engine.invalidate()                   # ... but a no-op move (Imagen.sample re-homes the U-Nets on every call) keeps them
            train_ops.invalidate(self)
        return out

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
        if self.training and torch.is_grad_enabled():
            # training (Imagen.forward): the differentiable training graph of the same module tree (HIP convolutions forward + backward on the GPU,
            # torch ops for the rest); sampling / evaluation takes the HIP inference engine
            if x.is_cuda and not train_ops.active(x) and not Unet._warned_torch_path:          # once per process
                Unet._warned_torch_path = True
                warnings.warn("minimagen_amd.Unet.forward: module in train() mode with autograd enabled and MINIMAGEN_TRAIN_HIP=0 -> torch ops "
                              "only; call .eval() or wrap the call in torch.no_grad() to run inference on the HIP engine", stacklevel=2)
            return self._forward_train(x, time, lowres_cond_img=lowres_cond_img, lowres_noise_times=lowres_noise_times,
                                       text_embeds=text_embeds, text_mask=text_mask, cond_drop_prob=cond_drop_prob)
        keep = prob_mask_like((x.shape[0],), 1 - cond_drop_prob, device='cpu')      # Unet.py:587
        with torch.no_grad():
            return self.engine().forward_once(x, time, lowres_cond_img=lowres_cond_img if self.lowres_cond else None,
                                              lowres_noise_times=lowres_noise_times if self.lowres_cond else None,
                                              text_embeds=text_embeds, text_mask=text_mask, keep=keep)

    def _forward_train(self, x, time, *, lowres_cond_img=None, lowres_noise_times=None, text_embeds=None, text_mask=None,
                       cond_drop_prob: float = 0.) -> torch.Tensor:
        """Unet.py:355-472 over the layers' own differentiable ``forward`` methods (minimagen_amd/layers.py): the training path -- on the GPU
        with the 3x3 convolutions, the Blocks and CrossEmbed on the HIP kernels in both directions (minimagen_amd/train_ops.py).  Same order of operations as the reference; per-sample conditioning dropout with probability ``cond_drop_prob``."""
        b = x.shape[0]
        dev_path = train_ops.active(x)      # the 3x3 convolutions (Blocks, Upsample, final_conv) forward and backward on the HIP kernels
        if dev_path:
            train_ops.begin_step(self)
        conv3 = lambda m, v: train_ops.conv3x3_forward(m, v) if (dev_path and train_ops.is_plain_conv3x3(m, v)) else m(v)
        resample = lambda m, v: train_ops.conv4x4s2_forward(m, v) if (dev_path and train_ops.is_conv4x4s2(m, v)) else m(v)      # Downsample k4 s2
        # ---- conditioning (Unet.py:508-634)
        hid = self.to_time_hiddens(time)
        t, tokens = self.to_time_cond(hid), self.to_time_tokens[0](hid).reshape(b, self.num_time_tokens, self.cond_dim)
        if self.lowres_cond:
            lhid = self.to_lowres_time_hiddens(lowres_noise_times)
            t = t + self.to_lowres_time_cond(lhid)
            tokens = torch.cat((tokens, self.to_lowres_time_tokens[0](lhid).reshape(b, self.num_time_tokens, self.cond_dim)), dim=-2)
        c = tokens
        if exists(text_embeds):
            text_tokens = self.text_to_cond(text_embeds)[:, :self.max_text_len]
            pad = self.max_text_len - text_tokens.shape[1]
            if pad > 0:
                text_tokens = F.pad(text_tokens, (0, 0, 0, pad))
            keep = prob_mask_like((b,), 1 - cond_drop_prob, device=x.device)
            keep_embed = keep[:, None, None]
            if exists(text_mask):
                tm = text_mask[:, :self.max_text_len]
                if pad > 0:
                    tm = F.pad(tm, (0, pad), value=False)
                keep_embed = tm[:, :, None] & keep_embed
            text_tokens = torch.where(keep_embed, text_tokens, self.null_text_embed.to(text_tokens.dtype))
            text_hiddens = self.to_text_non_attn_cond(text_tokens.mean(dim=-2))
            t = t + torch.where(keep[:, None], text_hiddens, self.null_text_hidden.to(t.dtype))
            c = torch.cat((tokens, text_tokens), dim=-2)
        c = self.norm_cond(c)
        if dev_path:
            # the step is bound by its launch count: the ResnetBlocks' time MLPs (SiLU -> Linear(time_cond_dim, 2 dim_out), layers.py:386-391, one
            # per block, all of the same t) run as ONE linear layer over the stacked weights -- 4 launches instead of 2 per block, and one
            # matrix product per direction in the backward instead of two per block; each block reads its slice (ResnetBlock.forward)
            rbs = self.__dict__.get("_mi_time_blocks")
            if rbs is None:
                rbs = self.__dict__["_mi_time_blocks"] = [m for m in self.modules() if isinstance(m, ResnetBlock) and m.time_mlp is not None]
            if len(rbs) > 1:
                lins = [m.time_mlp[1] for m in rbs]
                allss = F.linear(F.silu(t), torch.cat([l.weight for l in lins], 0), torch.cat([l.bias for l in lins], 0))
                for m, part in zip(rbs, allss.split([l.out_features for l in lins], dim=1)):
                    m.__dict__["_mi_scale_shift"] = part
        # ---- trunk (Unet.py:396-472)
        lowres = lowres_cond_img if self.lowres_cond else None
        if dev_path and train_ops.crossembed_supported(self.init_conv, x, lowres):
            x = train_ops.crossembed_forward(self.init_conv, x, lowres)       # matrix-core forward, shared-correlation weight gradient
        else:
            x = self.init_conv(torch.cat((x, lowres), dim=1) if self.lowres_cond else x)
        hiddens = []
        for pre, first, blocks, attn, post in self.downs:
            if exists(pre):
                x = resample(pre, x)
            x = first(x, t, c)
            for blk in blocks:
                x = blk(x, t)
                hiddens.append(x)
            x = attn(x)
            hiddens.append(x)
            if exists(post):
                x = resample(post, x)
        x = self.mid_block1(x, t, c)
        if exists(self.mid_attn):
            x = self.mid_attn(x)
        x = self.mid_block2(x, t, c)
        with_skip = lambda v: torch.cat((v, hiddens.pop() * self.skip_connect_scale), dim=1)
        for first, blocks, attn, up in self.ups:
            x = first(with_skip(x), t, c)
            for blk in blocks:
                x = blk(with_skip(x), t)
            x = attn(x)
            x = conv3(up[1], up[0](x)) if isinstance(up, nn.Sequential) else up(x)
        x = self.final_res_block(x, t)
        if dev_path:
            for m in self.__dict__.get("_mi_time_blocks", ()):          # (a block that did not run must not keep a slice of this step's graph)
                m.__dict__.pop("_mi_scale_shift", None)
        return conv3(self.final_conv, x)

    def forward_with_cond_scale(self, *args, cond_scale: float = 1., **kwargs) -> torch.Tensor:
        """Unet.py:474-506: both guidance halves run as ONE batch of 2B rows through the engine."""
        x, time = args
        if cond_scale == 1:         # the sampling API: always the HIP engine, whatever the module's train flag
            # Unet.py:482-485: the single forward still honours a caller's cond_drop_prob (default 0: keep every row's conditioning)
            keep1 = prob_mask_like((x.shape[0],), 1 - kwargs.get('cond_drop_prob', 0.), device='cpu')
            with torch.no_grad():
                return self.engine().forward_once(x, time, lowres_cond_img=kwargs.get('lowres_cond_img') if self.lowres_cond else None,
                                                  lowres_noise_times=kwargs.get('lowres_noise_times') if self.lowres_cond else None,
                                                  text_embeds=kwargs.get('text_embeds'), text_mask=kwargs.get('text_mask'),
                                                  keep=keep1)
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
