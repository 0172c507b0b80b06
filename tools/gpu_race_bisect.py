"""Bisect the concurrency corruption: the base-stage sampler (single-stage Imagen, B=32) while ANOTHER stream keeps the GPU busy.
argv: load kind (torch | sr), graph (1|0).  Env knobs of the engine are read at import, so each configuration is its own process."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

load_kind, use_graph = sys.argv[1], sys.argv[2] == "1"
dev = torch.device("cuda:0")
B, T = 32, 50
im, _ = bench.build_imagen("base64", T, dev)
emb, mask = bench.synthetic_text(B)
emb, mask = emb.to(dev), mask.to(dev)
ref = im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=7, _use_graph=use_graph).clone()
ref2 = im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=7, _use_graph=use_graph).clone()
torch.cuda.synchronize()
assert torch.equal(ref, ref2)
side = torch.cuda.Stream(device=dev)
if load_kind == "sr":
    im2, _ = bench.build_imagen("cascade64_256", T, dev)
    eng = im2.unets[1].engine()
    eng.pack()
    ws = eng.workspace(B, 2 * B, 256, 256)
    keep = torch.cat((torch.ones(B, dtype=torch.bool), torch.zeros(B, dtype=torch.bool)))
    with torch.cuda.stream(side):
        eng.set_text(ws, emb, mask, keep)
        eng.prepare_lowres(ws)
    def load():
        with torch.cuda.stream(side):
            for _ in range(60):
                eng.run(ws)
else:
    a = torch.randn(64, 3, 1024, 1024, device=dev)
    def load():
        with torch.cuda.stream(side):
            x = a
            for _ in range(200):
                x = x * 1.0001 + 0.1
bad = 0
for trial in range(4):
    torch.cuda.synchronize()
    load()
    out = im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=7, _use_graph=use_graph)
    torch.cuda.synchronize()
    d = (out - ref).abs().flatten(1).amax(1)
    rows = [int(r) for r in torch.nonzero(d > 0).flatten()]
    bad += len(rows)
    print(f"  trial {trial}: max|d| {d.max().item():.2e} rows {rows}", flush=True)
print(f"RESULT load={load_kind} graph={use_graph} env={ {k: v for k, v in os.environ.items() if k.startswith('MINIMAGEN_')} }: corrupted rows total {bad}", flush=True)
