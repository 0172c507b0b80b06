"""Restatement of ``resize_right.resize`` (Shocher's ResizeRight, pinned 0.0.2 in the
reference's requirements.txt line 31) for the only way MinImagen calls it:

    resize(image, scale_factors=s, pad_mode='reflect')      # minimagen/helpers.py:159

TEST INFRASTRUCTURE ONLY.  The package is neither installed nor vendored under
/root/reference, so this follows its published algorithm.
PARITY UNPINNED on the package itself: there is no resize-right implementation or golden
vector in the container.  Independent evidence: away from the (reflect-padded) border the
result agrees with Pillow's BICUBIC -- the same Keys a = -1/2 kernel and pixel-centre
convention -- to fp32 rounding (tests/test_oracle.py::test_cubic_resize_interior_matches_pillow_bicubic);
at the border the x4 case of the cascade is checked against hand-derived closed-form taps,
exact in binary (tests/test_kernels.py::BORDER_TAPS_X4 / border_closed_form_check).

Algorithm (per resized dim, dims processed in order of increasing scale factor,
ties in dim order => H then W for a (B,C,H,W) tensor and a scalar factor):
  1. out = ceil(s*in); grid[o] = o/s + (in-1)/2 - (out-1)/(2s)
  2. s<1 and antialiasing: support = 4/s, kernel(x) = s*cubic(s*x); else support 4, cubic
  3. left[o] = ceil(grid[o] - support/2 - eps), fov[o,k] = left[o]+k, k < ceil(support-eps)
  4. pad = (-fov[0,0], fov[-1,-1]-in+1); F.pad(mode); shift fov, grid by pad[0]
  5. w[o,k] = kernel(grid[o]-fov[o,k]) normalised to sum 1 (zero rows -> divide by 1)
  6. out[o] = sum_k w[o,k] * padded[fov[o,k]]
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def cubic(x: torch.Tensor) -> torch.Tensor:
    """Keys cubic, a = -1/2, support 4."""
    absx = x.abs()
    absx2 = absx ** 2
    absx3 = absx ** 3
    return ((1.5 * absx3 - 2.5 * absx2 + 1.) * (absx <= 1.).to(x.dtype) +
            (-0.5 * absx3 + 2.5 * absx2 - 4. * absx + 2.) * ((1. < absx) & (absx <= 2.)).to(x.dtype))


def taps_for_dim(in_sz: int, scale: float, antialiasing: bool = True):
    """Returns (out_sz, pad(left,right), fov[out,K] int64 (indices into the padded axis), w[out,K] fp32)."""
    eps = torch.finfo(torch.float32).eps
    out_sz = int(math.ceil(scale * in_sz))
    out_coords = torch.arange(out_sz)
    grid = out_coords / scale + (in_sz - 1) / 2 - (out_sz - 1) / (2 * scale)   # float32
    if scale < 1. and antialiasing:
        support = 4. / scale
        kern = lambda t: scale * cubic(scale * t)
    else:
        support = 4.
        kern = cubic
    left = (grid - support / 2 - eps).ceil().long()
    ordinal = torch.arange(int(math.ceil(support - eps)))
    fov = left[:, None] + ordinal[None, :]
    pad = (int(-fov[0, 0].item()), int(fov[-1, -1].item() - in_sz + 1))
    fov = fov + pad[0]
    grid = grid + pad[0]
    w = kern(grid[:, None] - fov.to(grid.dtype))
    sw = w.sum(1, keepdim=True)
    sw[sw == 0] = 1
    w = w / sw
    return out_sz, pad, fov, w


def _resize_dim(x: torch.Tensor, dim: int, scale: float, pad_mode: str) -> torch.Tensor:
    in_sz = x.shape[dim]
    out_sz, pad, fov, w = taps_for_dim(in_sz, scale)
    t = x.transpose(dim, -1)
    if pad != (0, 0):
        lead = t.shape[:-1]
        t3 = t.reshape(1, -1, t.shape[-1])
        if pad_mode == 'constant':
            t3 = F.pad(t3, pad, mode='constant', value=0.)
        else:
            mode = {'edge': 'replicate', 'reflect': 'reflect', 'symmetric': 'reflect'}[pad_mode]
            t3 = F.pad(t3, pad, mode=mode)
        t = t3.reshape(*lead, t3.shape[-1])
    neigh = t[..., fov]                      # (..., out, K)
    w = w.to(x.dtype)
    acc = neigh[..., 0] * w[:, 0]
    for k in range(1, w.shape[1]):           # sequential sum over taps
        acc = acc + neigh[..., k] * w[:, k]
    return acc.transpose(dim, -1)


def resize(image: torch.Tensor, scale_factors=None, pad_mode: str = 'constant', **_ignored) -> torch.Tensor:
    assert scale_factors is not None and image.ndim == 4
    s = float(scale_factors)
    if s == 1.:
        return image
    out = image
    for dim in (2, 3):                       # equal factors -> stable order H, W
        out = _resize_dim(out, dim, s, pad_mode)
    return out.contiguous()
