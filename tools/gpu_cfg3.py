"""Why was BASELINE config 3 (cascade 64->256, B=16, reduced precision) slow in the round-2 driver line?  Times sample() the way bench.py's
main loop does (warm-up 3, 8 timed calls) for {fp32, half} x {B=16, B=32} x {synchronous, pipelined}, and each call of a cold start."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

dev = torch.device("cuda:0")
T = 100
res = {}
for precision in ("half", "fp32"):
    for B in (16, 32):
        im, sizes = bench.build_imagen("cascade64_256", T, dev)
        emb, mask = bench.synthetic_text(B)
        emb, mask = emb.to(dev), mask.to(dev)
        cold = []
        for k in range(4):          # the first calls, one at a time: set-up cost (workspaces, graph capture + instantiation, allocator growth)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=k, _precision=precision)
            torch.cuda.synchronize(); cold.append((time.perf_counter() - t0) * 1e3)
        sync = bench.timed_calls(im, emb, mask, 3., precision, 8, 0, False) * 1e3
        pipe = bench.timed_calls(im, emb, mask, 3., precision, 8, 1, True) * 1e3
        key = f"{precision}_B{B}"
        res[key] = dict(cold_call_ms=[round(c, 1) for c in cold], sync_ms=round(sync, 1), pipelined_ms=round(pipe, 1),
                        steps_per_s_sync=round(B * 2 * T / sync * 1e3), steps_per_s_pipelined=round(B * 2 * T / pipe * 1e3))
        print(key, res[key], flush=True)
        del im
        torch.cuda.empty_cache()
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "cfg3_timing.json"), "w"), indent=1)
