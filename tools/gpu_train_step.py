"""Training step (Imagen.forward -> loss.backward() -> Adam) of the BASELINE U-Nets on the MI355X: the device path (3x3 convolutions forward +
backward on the HIP kernels, minimagen_amd/train_ops.py) against the torch-op path (MINIMAGEN_TRAIN_HIP=0 semantics), same weights, same
RNG.  Prints step time, peak memory, loss and the largest relative gradient difference; writes gpurun_out/train_step.json."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from minimagen_amd import train_ops

dev = torch.device("cuda:0")
im, sizes = bench.build_imagen("cascade64_256", 1000, dev)
im.train()
res = {}
for unet_number, B in ((1, int(os.environ.get("B1", "32"))), (2, int(os.environ.get("B2", "8")))):
    S = sizes[-1]
    imgs = torch.rand(B, 3, S, S, device=dev)
    emb, mask = bench.synthetic_text(B)
    emb, mask = emb.to(dev), mask.to(dev)
    opt = torch.optim.Adam(im.unets[unet_number - 1].parameters(), lr=1e-4)
    grads = {}
    only = os.environ.get("PROFILE_ONLY")          # "1" / "0": run one path only (for a kernel profile of that path)
    for hip in ((False, True) if only is None else (only == "1",)):
        train_ops.ENABLED = hip
        def step(seed, do_opt):
            torch.manual_seed(seed)
            loss = im(imgs, text_embeds=emb, text_masks=mask, unet_number=unet_number)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            if do_opt:
                opt.step()
            return loss
        loss = step(5, False)
        grads[hip] = (float(loss), {n: p.grad.clone() for n, p in im.unets[unet_number - 1].named_parameters()})
        for k in range(3):
            step(6 + k, False)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        n = 10
        for k in range(n):
            step(20 + k, False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        key = f"unet{unet_number}_B{B}_{'hip_convs' if hip else 'torch_ops'}"
        res[key] = dict(ms_per_fwd_bwd=round(dt * 1e3, 2), peak_mem_MB=round(torch.cuda.max_memory_allocated() / 2 ** 20), loss=grads[hip][0])
        print(key, res[key], flush=True)
    if only is None:
      worst = max(((grads[True][1][n] - g).abs().max().item() / max(1e-3, g.abs().max().item()), n) for n, g in grads[False][1].items())
      res[f"unet{unet_number}_max_rel_grad_diff"] = worst
      print("largest relative gradient difference:", worst, "loss", grads[False][0], grads[True][0], flush=True)
    # with the optimiser in the loop (weights re-packed every step)
    train_ops.ENABLED = True if only is None else (only == "1")
    for with_opt in (True,):
        for k in range(2):
            torch.manual_seed(40 + k); l = im(imgs, text_embeds=emb, text_masks=mask, unet_number=unet_number); opt.zero_grad(set_to_none=True); l.backward(); opt.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(10):
            torch.manual_seed(50 + k); l = im(imgs, text_embeds=emb, text_masks=mask, unet_number=unet_number); opt.zero_grad(set_to_none=True); l.backward(); opt.step()
        torch.cuda.synchronize()
        res[f"unet{unet_number}_B{B}_hip_convs_with_adam_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 2)
        print("with Adam + re-pack:", res[f"unet{unet_number}_B{B}_hip_convs_with_adam_ms"], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"train_step_B{os.environ.get('B1', '32')}_{os.environ.get('B2', '8')}.json"), "w"), indent=1)
