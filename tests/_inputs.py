"""Regenerate golden inputs from the metadata stored by tests/golden/make_golden.py."""
import json
import os

import torch

from oracle import restated as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=False)


def unet_params():
    return json.load(open(os.path.join(GOLDEN, "unet_params.json")))


def seeded(shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def text(meta):
    emb, mask = R.synthetic_text(2, length=meta["L"], seed=meta["text_seed"])
    r, c = 1, 6
    mask[r, c:] = False
    emb = emb.masked_fill(~mask[:, :, None], 0.)
    b = meta.get("B", 2)
    return emb[:b].contiguous(), mask[:b].contiguous()


def hostile_state_dict(sd, seed):
    """Trained-like / adversarial ranges on a random-init state dict ("what's weak" #2): every weight matrix / filter bank
    scaled by a power-of-two-ish factor drawn log-uniformly from 2^-10 .. 2^6, ONE element of every output filter another 2^8 larger (an
    outlier weight), normalisation gains spread over 2^-4 .. 2^4 per channel (outlier channels), biases over 2^-6 .. 2^4.  The fp16-split
    matrix-core kernels derive their power-of-two operand scalings from bounds on exactly these quantities."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        v = v.clone()
        if v.is_floating_point() and v.numel() > 1:
            if v.dim() >= 2:
                v *= 2.0 ** float(torch.rand((), generator=g) * 16 - 10)
                flat = v.view(v.shape[0], -1)
                j = torch.randint(0, flat.shape[1], (flat.shape[0],), generator=g)
                flat[torch.arange(flat.shape[0]), j] *= 256.0
            elif any(t in k for t in ("norm", "gamma")) or k.endswith(".g"):
                v *= 2.0 ** (torch.rand(v.shape, generator=g) * 8 - 4)
            else:
                v *= 2.0 ** float(torch.rand((), generator=g) * 10 - 6)
        out[k] = v
    return out
