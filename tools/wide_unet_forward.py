"""Unet() with the reference's default arguments (dim 128, channels 128 / 256 / 512, Unet.py:31-48) through the wide regime of the
kernels: B image-forwards with guidance at 64 x 64, timed per launch; run under rocprofv3 --pmc for the MFMA utilisation of the
C = 128-512 convs (tools/gpu_wide_pmc.sh).  usage: python tools/wide_unet_forward.py [B]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minimagen_amd import _lib as L
from minimagen_amd.Unet import Unet
from oracle import restated as R          # synthetic_text only (development tool, not a product path)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
L.use_library(L.DEFAULT_LIB)
dev = torch.device("cuda:0")
torch.manual_seed(6)
u = Unet().to(dev).eval()
emb, mask = R.synthetic_text(B, length=20, seed=8)
x = torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(41)).to(dev)
tm = torch.randint(0, 100, (B,), generator=torch.Generator().manual_seed(1)).to(dev)
emb, mask = emb.to(dev), mask.to(dev)
for _ in range(2):
    o = u.forward_with_cond_scale(x, tm, text_embeds=emb, text_mask=mask, cond_scale=3.)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    o = u.forward_with_cond_scale(x, tm, text_embeds=emb, text_mask=mask, cond_scale=3.)
torch.cuda.synchronize()
print(f"Unet() default, 64x64, B={B} (2B rows with guidance): {(time.perf_counter() - t0) / n * 1e3:.2f} ms per forward_with_cond_scale (eager launches, host included)")
