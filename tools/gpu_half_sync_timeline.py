"""Where does a synchronous sample() call spend its time (host enqueue vs GPU, per stage), fp32 vs reduced precision, B=32?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from minimagen_amd.Imagen import Imagen

dev = torch.device("cuda:0")
B, T = int(os.environ.get("B", "32")), 100
rec = []
orig_loop = Imagen._p_sample_loop
def loop(self, unet, shape, **kw):
    st = torch.cuda.current_stream()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    h0 = time.perf_counter(); e0.record(st)
    out = orig_loop(self, unet, shape, **kw)
    e1.record(st); h1 = time.perf_counter()
    rec.append((shape[-1], h0, h1, e0, e1))
    return out
Imagen._p_sample_loop = loop
for precision in os.environ.get("ORDER", "fp32,half").split(","):
    im, sizes = bench.build_imagen("cascade64_256", T, dev)
    emb, mask = bench.synthetic_text(B)
    emb, mask = emb.to(dev), mask.to(dev)
    for k in range(3):
        im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=k, _precision=precision)
    torch.cuda.synchronize()
    for k in range(3):
        rec.clear()
        t0 = time.perf_counter()
        im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=10 + k, _precision=precision)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        parts = ", ".join(f"stage {s}: host enqueue {1e3*(h1-h0):.1f} ms (starts at +{1e3*(h0-t0):.1f}), GPU {e0.elapsed_time(e1):.1f} ms" for s, h0, h1, e0, e1 in rec)
        print(f"{precision} call {k}: sample() returned after {1e3*(t1-t0):.1f} ms, GPU done after {1e3*(t2-t0):.1f} ms; {parts}", flush=True)
    del im
    torch.cuda.empty_cache()
