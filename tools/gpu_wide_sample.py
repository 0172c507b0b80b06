"""Sampling with the reference's default Unet() (dim 128, channels 128 / 256 / 512) at 64x64 through Imagen.sample (captured step graphs):
per-step time, next to tools/wide_unet_forward.py's eager per-evaluation time.  usage: python tools/gpu_wide_sample.py [B] [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minimagen_amd.Imagen import Imagen
from minimagen_amd.Unet import Unet
from oracle import restated as R          # synthetic_text only (development tool)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 25          # (T <= 20 is outside the reference's linear schedule: beta_end >= 1, diffusion_model.py:23)
dev = torch.device("cuda:0")
torch.manual_seed(6)
im = Imagen((Unet(),), text_encoder_name="t5_small", image_sizes=(64,), timesteps=T, cond_drop_prob=0.1).to(dev).eval()
emb, mask = R.synthetic_text(B, length=20, seed=8)
emb, mask = emb.to(dev), mask.to(dev)
for k in range(2):
    im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=k)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for k in range(n):
    out = im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=10 + k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"Unet() default @64x64, B={B}, T={T}, cond_scale 3: {dt * 1e3:.1f} ms per sample() = {dt / T * 1e3:.2f} ms per denoising step (captured graphs), finite={bool(torch.isfinite(out).all())}")
