"""Host-side profile (cProfile) of the SR training step at B = 32 on the device path: where the Python time of a step goes -- the step is bound by
its ~1 150 launches, so per-call host cost is what is left to remove.  Prints the step time with and without the profiler and the top functions."""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from minimagen_amd import optim as mi_optim

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
t0 = time.perf_counter()
for k in range(10):
    step(10 + k)
t_host = time.perf_counter() - t0          # the host's own time to ISSUE ten steps
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"10 steps: host issue {t_host * 100:.2f} ms per step, with the GPU drained {t_all * 100:.2f} ms per step")
pr = cProfile.Profile()
pr.enable()
for k in range(10):
    step(30 + k)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(os.environ.get("TOP", "45")))
