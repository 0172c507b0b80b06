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
