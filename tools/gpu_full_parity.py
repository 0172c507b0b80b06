"""One-off full-length parity measurement on the GPU box: the BASELINE cascade (unet_0 @64x64 + unet_1 @256x256, T=100 per stage,
cond_scale 3, dynamic thresholding) at B=4 through the HIP path and through the CPU oracle with the same injected noise.
~2 minutes (the oracle's 800 U-Net forwards dominate).  Prints max / mean |d| on the [0,1] images."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from minimagen_amd import _lib as L
from minimagen_amd.Imagen import Imagen
from minimagen_amd.Unet import Unet
from oracle import restated as R
from tests import _inputs as I

L.use_library(L.DEFAULT_LIB)
dev = torch.device("cuda:0")
T, B = int(os.environ.get("PARITY_T", "100")), 4
p = I.unet_params()
im = Imagen([Unet(**p["unet0"]), Unet(**p["unet1"])], text_encoder_name="t5_small", image_sizes=(64, 256), timesteps=T, cond_drop_prob=0.15)
sds = [I.load("unet0_sd.pt"), I.load("unet1_sd.pt")]
for u, sd in zip(im.unets, sds):
    u.load_state_dict(sd)
im = im.to(dev)
emb, mask = R.synthetic_text(B, length=64, seed=7)
for prec in ("fp32", "half"):
    t0 = time.time()
    out = im.sample(text_embeds=emb.to(dev), text_masks=mask.to(dev), cond_scale=3., _noise=R.make_randn(1234), _precision=prec).cpu()
    st16 = [u.engine().workspace(B, 2 * B, S, S, precision=prec).store16 for u, S in zip(im.unets, (64, 256))]
    print(f"HIP {prec}: {time.time() - t0:.1f}s; bf16 activation storage per stage: {st16}", flush=True)
    if prec == "fp32":
        t0 = time.time()
        torch.set_num_threads(min(os.cpu_count(), 32))
        ref = R.sample(sds, [64, 256], T, text_embeds=emb, text_masks=mask, cond_scale=3., randn=R.make_randn(1234))
        print(f"oracle: {time.time() - t0:.1f}s", flush=True)
    d = (out - ref).abs()
    print(f"cascade 64->256, B={B}, T={T}/stage, cond_scale 3, {prec}: max|d| = {d.max().item():.3e}, mean|d| = {d.mean().item():.3e}, "
          f"p99.9 = {d.flatten().kthvalue(int(0.999 * d.numel())).values.item():.3e}", flush=True)
