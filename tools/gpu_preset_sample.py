"""Sampling with the reference's Base / Super presets at full width through Imagen.sample (captured step graphs): per-step time of ONE stage.
usage: python tools/gpu_preset_sample.py base|super [B] [T]     (base: 64 x 64; super: 64 -> 256 with the low-res conditioning image)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minimagen_amd.Imagen import Imagen
from minimagen_amd.Unet import Base, Super
from oracle import restated as R          # synthetic_text only (development tool)
which = sys.argv[1] if len(sys.argv) > 1 else "base"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
T = int(sys.argv[3]) if len(sys.argv) > 3 else 25
dev = torch.device("cuda:0")
torch.manual_seed(6)
if which == "base":
    im = Imagen((Base(),), text_encoder_name="t5_small", image_sizes=(64,), timesteps=T, cond_drop_prob=0.1).to(dev).eval()
    kw = {}
else:
    # the SR stage alone: a two-stage Imagen whose first stage is skipped by handing in its output (start_image_or_video is not in the reference's
    # API; the stage is timed through the U-Net's own guided forward instead)
    im = None
emb, mask = R.synthetic_text(B, length=20, seed=8)
emb, mask = emb.to(dev), mask.to(dev)
if im is not None:
    for k in range(2):
        im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 2
    for k in range(n):
        out = im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=10 + k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"Base() @64x64, B={B}, T={T}, cond_scale 3: {dt / T * 1e3:.2f} ms per denoising step (captured graphs), {sum(p.numel() for p in im.parameters()) / 1e6:.0f} M parameters, finite={bool(torch.isfinite(out).all())}")
else:
    u = Super(lowres_cond=True).to(dev).eval()
    x = torch.randn(B, 3, 256, 256, device=dev)
    lr = torch.randn(B, 3, 256, 256, device=dev)
    tm = torch.randint(0, 100, (B,), device=dev)
    lt = torch.full((B,), 20, device=dev)
    for _ in range(2):
        o = u.forward_with_cond_scale(x, tm, text_embeds=emb, text_mask=mask, cond_scale=3., lowres_cond_img=lr, lowres_noise_times=lt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        o = u.forward_with_cond_scale(x, tm, text_embeds=emb, text_mask=mask, cond_scale=3., lowres_cond_img=lr, lowres_noise_times=lt)
    torch.cuda.synchronize()
    print(f"Super() @256x256, B={B} (2B rows with guidance): {(time.perf_counter() - t0) / n * 1e3:.2f} ms per forward_with_cond_scale (eager launches, host included), "
          f"{sum(p.numel() for p in u.parameters()) / 1e6:.0f} M parameters, finite={bool(torch.isfinite(o).all())}")
