"""Kernels of two streams side by side (what the stage-pipelined sampler does): every kernel of a U-Net evaluation as the VICTIM on one
stream, another U-Net's kernels as the LOAD on a second stream; the victim's outputs must stay bit-identical to an idle-GPU run.
(Round 3 found packed-fp32 VALU instructions with SGPR operands returning wrong products next to another kernel's matrix-core waves.)"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from minimagen_amd import _lib as L

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "32"))
REPS = int(os.environ.get("REPS", "40"))
im, _ = bench.build_imagen("cascade64_256", 100, dev)
emb, mask = bench.synthetic_text(B)
emb, mask = emb.to(dev), mask.to(dev)
keep = torch.cat((torch.ones(B, dtype=torch.bool), torch.zeros(B, dtype=torch.bool)))
engs, wss = [], []
for stage, S in enumerate((64, 256)):
    eng = im.unets[stage].engine(); eng.pack()
    ws = eng.workspace(B, 2 * B, S, S, precision=os.environ.get("PRECISION", "fp32"))
    g = torch.Generator().manual_seed(stage)
    ws.x.copy_(torch.randn(ws.x.shape, generator=g))
    ws.times.fill_(37)
    if ws.lowres is not None:
        ws.lowres.copy_(torch.randn(ws.lowres.shape, generator=g)); ws.lowres_times.fill_(20)
        eng.prepare_lowres(ws)
    eng.set_text(ws, emb, mask, keep)
    engs.append(eng); wss.append(ws)
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

def outputs(ws):
    return [t for t in ws.tensors if t is not None and t.is_floating_point()] + [ws.pred]

total_bad = 0
for victim in (0, 1):
    load = 1 - victim
    with torch.cuda.stream(streams[victim]):
        engs[victim].run(wss[victim])
        ref = [t.clone() for t in outputs(wss[victim])]
    torch.cuda.synchronize()
    with torch.cuda.stream(streams[victim]):
        engs[victim].run(wss[victim])
        again = [t.clone() for t in outputs(wss[victim])]
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(ref, again)), "not deterministic on an idle GPU"
    bad = 0
    nrep = REPS if victim == 0 else max(4, REPS // 8)
    for rep in range(nrep):
        with torch.cuda.stream(streams[load]):
            for _ in range(12 if load == 1 else 120):
                engs[load].run(wss[load])
        with torch.cuda.stream(streams[victim]):
            for _ in range(6 if victim == 0 else 1):
                engs[victim].run(wss[victim])
                got = [t.clone() for t in outputs(wss[victim])]
        torch.cuda.synchronize()
        nb = sum(0 if torch.equal(a, b) else 1 for a, b in zip(ref, got))
        if nb and bad < 3:
            first = next(i for i, (a, b) in enumerate(zip(ref, got)) if not torch.equal(a, b))
            d = (ref[first].float() - got[first].float()).abs()
            print(f"  victim stage {victim} rep {rep}: {nb} tensors differ; first tensor #{first} shape {tuple(ref[first].shape)} max|d| {d.max().item():.3e} count {(d > 0).sum().item()}", flush=True)
        bad += 1 if nb else 0
    total_bad += bad
    print(f"victim = stage {victim} U-Net evaluation ({'64' if victim == 0 else '256'}^2), load = stage {load}: {bad} of {nrep} repetitions corrupted", flush=True)
print("RESULT total corrupted repetitions:", total_bad, flush=True)
sys.exit(1 if total_bad else 0)
