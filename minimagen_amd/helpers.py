"""Small utilities with the reference's names (minimagen/helpers.py) that the hot path needs."""
from __future__ import annotations

import math
from contextlib import contextmanager
from functools import wraps

import torch


def exists(val) -> bool:
    return val is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def cast_tuple(val, length: int = None) -> tuple:
    """helpers.py:9-22"""
    if isinstance(val, list):
        val = tuple(val)
    output = val if isinstance(val, tuple) else ((val,) * default(length, 1))
    if exists(length):
        assert len(output) == length
    return output


def eval_decorator(fn):
    """helpers.py:35-46"""
    @wraps(fn)
    def inner(model, *args, **kwargs):
        was_training = model.training
        model.eval()
        out = fn(model, *args, **kwargs)
        model.train(was_training)
        return out
    return inner


def identity(t, *args, **kwargs):
    return t


def maybe(fn):
    @wraps(fn)
    def inner(x):
        if not exists(x):
            return x
        return fn(x)
    return inner


def module_device(module: torch.nn.Module) -> torch.device:
    return next(module.parameters()).device


@contextmanager
def null_context(*args, **kwargs):
    yield


def prob_mask_like(shape: tuple, prob: float, device) -> torch.Tensor:
    """helpers.py:121-135 (host-side mask; deterministic for prob in {0,1}, the only values sampling uses)."""
    if prob == 1:
        return torch.ones(shape, device=device, dtype=torch.bool)
    elif prob == 0:
        return torch.zeros(shape, device=device, dtype=torch.bool)
    return torch.zeros(shape, device=device).float().uniform_(0, 1) < prob


# ---- cubic resize tap tables (helpers.py:138-164 -> resize_right.resize(image, scale_factors, pad_mode='reflect'))
def _cubic(x: torch.Tensor) -> torch.Tensor:
    ax = x.abs()
    ax2, ax3 = ax * ax, ax * ax * ax
    return ((1.5 * ax3 - 2.5 * ax2 + 1.) * (ax <= 1.).to(x.dtype) +
            (-0.5 * ax3 + 2.5 * ax2 - 4. * ax + 2.) * ((1. < ax) & (ax <= 2.)).to(x.dtype))


def cubic_taps(in_sz: int, out_sz_target: int, pad_mode: str = 'reflect'):
    """Tap tables of resize_right (v0.0.2) for one axis: returns (out_sz, idx[out,K] int32, w[out,K] fp32).

    The pad (pad_mode) is folded into ``idx`` so the kernel gathers straight from the unpadded axis.
    Follows the package's published algorithm (projected grid with pixel-centre alignment, support 4 for
    up-scaling, 4/s with kernel s*cubic(s x) for antialiased down-scaling, weights normalised per output).
    """
    scale = out_sz_target / in_sz
    eps = torch.finfo(torch.float32).eps
    out_sz = int(math.ceil(scale * in_sz))
    grid = torch.arange(out_sz) / scale + (in_sz - 1) / 2 - (out_sz - 1) / (2 * scale)
    if scale < 1.:
        support = 4. / scale
        kern = lambda t: scale * _cubic(scale * t)
    else:
        support = 4.
        kern = _cubic
    left = (grid - support / 2 - eps).ceil().long()
    fov = left[:, None] + torch.arange(int(math.ceil(support - eps)))[None, :]
    pad_l = int(-fov[0, 0].item())
    w = kern((grid + pad_l)[:, None] - (fov + pad_l).to(grid.dtype))
    sw = w.sum(1, keepdim=True)
    sw[sw == 0] = 1
    w = (w / sw).float().contiguous()
    idx = fov.clone()
    if pad_mode == 'reflect':
        idx = torch.where(idx < 0, -idx, idx)
        idx = torch.where(idx >= in_sz, 2 * (in_sz - 1) - idx, idx)
    elif pad_mode == 'edge':
        idx = idx.clamp(0, in_sz - 1)
    else:
        raise NotImplementedError(f"pad_mode {pad_mode}")
    assert int(idx.min()) >= 0 and int(idx.max()) < in_sz, "pad wider than the image"
    return out_sz, idx.int().contiguous(), w


def quantile_rank(n: int, q: float):
    """torch.quantile's rank arithmetic, carried out in the input dtype fp32:
    rank = fp32(q) * (n-1); k_lo = floor(rank); w = rank - k_lo."""
    rank = torch.tensor(q, dtype=torch.float32) * torch.tensor(n - 1, dtype=torch.float32)
    lo = int(torch.floor(rank).item())
    w = float((rank - torch.tensor(float(lo), dtype=torch.float32)).item())
    return lo, min(lo + 1, n - 1), w


def normalize_neg_one_to_one(img):
    """[0, 1] -> [-1, 1] (helpers.py:99-107).  The sampling path has this fused into mi_lowres_augment; this is the tensor form."""
    return img * 2 - 1


def unnormalize_zero_to_one(img):
    """[-1, 1] -> [0, 1] (helpers.py:110-118; fused into mi_finalize_images on the sampling path)"""
    return (img + 1) * 0.5


_TAPS_DEV = {}


def _taps_on(in_sz: int, out_sz: int, pad_mode: str, device, dtype):
    """the tap tables of one (size pair, padding) resident on ``device`` (no host -> device copy per call: the training step stays capturable
    in a HIP graph and the eager loop loses two copies per resize)"""
    key = (in_sz, out_sz, pad_mode, str(device), dtype)
    if key not in _TAPS_DEV:
        _, idx, w = cubic_taps(in_sz, out_sz, pad_mode)
        _TAPS_DEV[key] = (idx.long().to(device), w.to(device, dtype))
    return _TAPS_DEV[key]


def resize_image_to(image: torch.Tensor, target_image_size: int, clamp_range: tuple = None, pad_mode: str = 'reflect') -> torch.Tensor:
    """helpers.py:138-164 as tensor ops (training path; the sampling path runs mi_resize_fwd on the same tap tables): resize_right's
    cubic resampling, H pass then W pass, antialiased when shrinking; identity when the size already matches."""
    orig = image.shape[-1]
    if orig == target_image_size:
        return image
    out = image
    for dim, in_sz in ((-2, image.shape[-2]), (-1, image.shape[-1])):
        idx, w = _taps_on(in_sz, int(round(target_image_size * in_sz / orig)), pad_mode, out.device, out.dtype)
        g = out.index_select(dim, idx.reshape(-1))
        if dim == -2:
            g = g.reshape(*out.shape[:-2], idx.shape[0], idx.shape[1], out.shape[-1])
            out = (g * w[:, :, None]).sum(-2)
        else:
            g = g.reshape(*out.shape[:-1], idx.shape[0], idx.shape[1])
            out = (g * w).sum(-1)
    if exists(clamp_range):
        out = out.clamp(*clamp_range)
    return out
