"""Experiment: the training step's forward + backward (Imagen.forward -> loss.backward(), HIP conv / attention kernels + torch ops) captured in ONE
HIP graph (torch.cuda.CUDAGraph) and replayed -- what the step costs on the GPU when the host no longer paces it (the reference's batch sizes are
host-launch-bound in eager mode).  usage: python tools/gpu_train_graph.py [unet_number] [B]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
im, sizes = bench.build_imagen("cascade64_256", 1000, dev)
im.train()
S = sizes[-1]
imgs = torch.rand(B, 3, S, S, device=dev)
emb, mask = bench.synthetic_text(B)
emb, mask = emb.to(dev), mask.to(dev)
params = list(im.unets[n - 1].parameters())

def step():
    loss = im(imgs, text_embeds=emb, text_masks=mask, unet_number=n)
    loss.backward()
    return loss

for _ in range(3):                                  # eager warm-up (packs, library auto-tuning)
    im.zero_grad(set_to_none=True); step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    im.zero_grad(set_to_none=True); step()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 10
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        im.zero_grad(set_to_none=True); step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
im.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
ref = [p.grad.clone() for p in params]
l0 = float(loss)
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / 20
print(f"unet {n - 1}, B={B}, {S}x{S}: forward + backward eager {eager * 1e3:.2f} ms, one captured HIP graph {graph * 1e3:.2f} ms per replay "
      f"(loss {l0:.4f}, gradients finite: {all(bool(torch.isfinite(r).all()) for r in ref)})")
