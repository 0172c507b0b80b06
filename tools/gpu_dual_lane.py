"""Experiment: is the per-launch floor of the small-image levels idle GPU time that a second, independent sampling lane can fill?
Times one sample() of B rows against K concurrent lanes of B/K rows each (separate Imagen instances = separate stage streams,
workspaces and graphs; same weights; Philox keyed by the global row, so the union of the lanes' outputs is bit-identical)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

dev = torch.device("cuda:0")
T = int(os.environ.get("T", "100"))
B = int(os.environ.get("B", "32"))
workload = os.environ.get("WORKLOAD", "cascade64_256")
calls = 6
res = {}
ims = []
for k in range(4):
    im, sizes = bench.build_imagen(workload, T, dev)
    if ims:
        im.load_state_dict(ims[0].state_dict())
    ims.append(im)
emb, mask = bench.synthetic_text(B)
emb, mask = emb.to(dev), mask.to(dev)
ref = ims[0].sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=5)
for lanes in (1, 2, 4):
    n = B // lanes
    parts = [(emb[i * n:(i + 1) * n].contiguous(), mask[i * n:(i + 1) * n].contiguous()) for i in range(lanes)]
    def run(seed, pipelined):
        outs = []
        for i in range(lanes):
            outs.append(ims[i].sample(text_embeds=parts[i][0], text_masks=parts[i][1], cond_scale=3., _seed=seed, _sample_offset=i * n, _async=True))
        if not pipelined:
            torch.cuda.synchronize()
        return outs
    outs = run(5, False)
    torch.cuda.synchronize()
    same = bool(torch.equal(torch.cat(outs, 0), ref))
    for pipelined in (False, True):
        run(6, pipelined); run(7, pipelined)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(calls):
            run(10 + c, pipelined)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / calls
        key = f"lanes{lanes}_{'pipelined' if pipelined else 'sync'}"
        res[key] = dict(ms_per_batch=round(dt * 1e3, 2), steps_per_s=round(B * T * len(sizes) / dt), bit_identical_to_one_lane=same)
        print(key, res[key], flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"dual_lane_{workload}_B{B}.json"), "w"), indent=1)
