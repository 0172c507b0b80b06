"""Diagnose: the second of two back-to-back pipelined sample() calls differs from the same call made synchronously (bench.py's check, B=32)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from minimagen_amd.Imagen import Imagen

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "32"))
T = int(os.environ.get("T", "100"))
im, sizes = bench.build_imagen("cascade64_256", T, dev)
emb, mask = bench.synthetic_text(B)
emb, mask = emb.to(dev), mask.to(dev)
stash = {}
orig = Imagen._lowres_conditioning
def spy(self, img, *a, **k):
    stash["img0"] = img.clone()
    return orig(self, img, *a, **k)
Imagen._lowres_conditioning = spy

def call(seed, pipelined):
    out = im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=seed, _async=pipelined)
    return out, stash["img0"]

for k in range(2):
    call(k, True)
torch.cuda.synchronize()
ref, ref0 = call(7, False)
ref, ref0 = ref.clone(), ref0.clone()
torch.cuda.synchronize()
ref_b, ref0_b = call(7, False)
torch.cuda.synchronize()
print("sync vs sync: final equal", torch.equal(ref, ref_b), "stage0 equal", torch.equal(ref0, ref0_b), flush=True)
for trial in range(4):
    a, a0 = call(6, True)
    b, b0 = call(7, True)
    torch.cuda.synchronize()
    d0 = (b0 - ref0).abs()
    d1 = (b - ref).abs()
    rows0 = [int(r) for r in torch.nonzero(d0.flatten(1).amax(1) > 0).flatten()]
    rows1 = [int(r) for r in torch.nonzero(d1.flatten(1).amax(1) > 0).flatten()]
    print(f"trial {trial}: stage0 max|d| {d0.max().item():.3e} rows {rows0[:40]}  final max|d| {d1.max().item():.3e} rows {rows1[:40]}", flush=True)
# single pipelined call after an idle GPU (nothing overlaps its base stage)
torch.cuda.synchronize()
c, c0 = call(7, True)
torch.cuda.synchronize()
print("lone pipelined call: stage0 equal", torch.equal(c0, ref0), "final equal", torch.equal(c, ref), flush=True)
