"""torch.profiler view of the SR training step's HOST time (both the calling thread and the autograd thread): CPU time by operator / autograd node,
so that the Python cost of every custom Function's forward and backward is visible (cProfile sees the calling thread only)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from minimagen_amd import optim as mi_optim
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
im, sizes = bench.build_imagen("cascade64_256", 1000, dev)
im.train()
B, S = 32, sizes[-1]
imgs = torch.rand(B, 3, S, S, device=dev)
emb, mask = bench.synthetic_text(B)
emb, mask = emb.to(dev), mask.to(dev)
params = list(im.unets[1].parameters())
opt = mi_optim.Adam(params, lr=1e-6)


def step(k):
    torch.manual_seed(k)
    loss = im(imgs, text_embeds=emb, text_masks=mask, unet_number=2)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 50)
    opt.step()


for k in range(4):
    step(k)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for k in range(5):
        step(10 + k)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=int(os.environ.get("TOP", "40")), max_name_column_width=60))
