"""Round-3 open item: the SECOND Imagen built in one process ran its SR stage at 261-272 ms instead of 147-150 ms, the third at full speed again
(DESIGN.md section 6).  Build the cascade several times in one process; per instance: SR-stage time of synchronous sample() calls, the per-launch
times of its heaviest launches (program order, HIP events), and where its big tensors sit in the address space."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from minimagen_amd import _lib as L

dev = torch.device("cuda:0")
# per-phase device time of sample(): wrap the engine / sampler entry points with HIP events on the stream they run on
from minimagen_amd.engine import UnetEngine
from minimagen_amd.Imagen import Imagen
PH = {}
def wrap(cls, name):
    orig = getattr(cls, name)
    def f(self, *a, **k):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0 = time.perf_counter(); e0.record(st)
        out = orig(self, *a, **k)
        e1.record(st)
        PH.setdefault(name, []).append((e0, e1, time.perf_counter() - h0))
        return out
    setattr(cls, name, f)
for cls, name in ((UnetEngine, "set_text"), (UnetEngine, "prepare_step_tables"), (UnetEngine, "pack"), (Imagen, "_lowres_conditioning"), (Imagen, "_p_sample_loop")):
    wrap(cls, name)
B, T = int(os.environ.get("B", "32")), int(os.environ.get("T", "25"))
MODE = os.environ.get("MODE", "empty_cache")        # empty_cache | keep_cache | no_del
keep = []
for inst in range(int(os.environ.get("N", "4"))):
    precision = os.environ.get("PRECISION", "fp32")
    im, sizes = bench.build_imagen("cascade64_256", T, dev)
    emb, mask = bench.synthetic_text(B)
    emb, mask = emb.to(dev), mask.to(dev)
    for k in range(2):
        im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=k, _precision=precision)
    torch.cuda.synchronize()
    PH.clear()
    t0 = time.perf_counter()
    for k in range(3):
        im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=10 + k, _precision=precision)
    torch.cuda.synchronize()
    call_ms = (time.perf_counter() - t0) / 3 * 1e3
    print("   phases (device ms / host ms, summed over 3 calls): " + "; ".join(f"{n} {sum(e0.elapsed_time(e1) for e0, e1, _ in v):.1f} / {sum(h for _, _, h in v) * 1e3:.1f}" for n, v in PH.items()), flush=True)
    g = bench.graph_step_ms(im, 1, B, 3.0, T, precision=precision)
    # the same captured graph replayed on each of this instance's stage streams (what sample() does) and on the caller's stream
    lib = L.lib()
    wsr = im.unets[1].engine().workspace(B, 2 * B, 256, 256, precision=precision)
    stt_, ent = [(stt, e) for stt in wsr.sampler_state.values() for e in stt.graphs.values() if e.get("graph")][-1]
    line = []
    for nm, sobj in [("caller", torch.cuda.current_stream())] + [(f"lane{l}.stage{k}(prio {sx.priority})", sx) for l, ln in enumerate(im._stage_streams) for k, sx in enumerate(ln)]:
        torch.cuda.synchronize()
        with torch.cuda.stream(sobj):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            L.check(lib.mi_step_set(L.ptr(stt_.t_state), L.ptr(wsr.times), B, T - 1, sobj.cuda_stream), "mi_step_set")      # the graph walks the step tables downwards
            lib.mi_graph_launch(ent["graph"], sobj.cuda_stream)
            e0.record(sobj)
            nrep = max(1, (T - 2 * ent.get("per", 1)) // ent.get("per", 1))
            for _ in range(nrep):
                lib.mi_graph_launch(ent["graph"], sobj.cuda_stream)
            e1.record(sobj)
        torch.cuda.synchronize()
        line.append(f"{nm} {e0.elapsed_time(e1) / nrep / ent.get('per', 1):.3f}")
    print("   ms per step of the SR graph by launch stream: " + "; ".join(line), flush=True)
    rows = bench.op_breakdown(im, 1, B, 3.0, reps=10, precision=precision)
    heavy = sorted([r for r in rows if r["ms"] > 0.04], key=lambda r: -r["ms"])[:6]
    eng = im.unets[1].engine()
    ws = eng.workspace(B, 2 * B, 256, 256, precision=precision)
    big = sorted({t.data_ptr(): t for t in ws.tensors if t is not None and t.numel() * t.element_size() >= (32 << 20)}.values(), key=lambda t: t.data_ptr())
    print(f"instance {inst}: sample() {call_ms:.1f} ms per call (T={T}), SR graph step {g:.3f} ms, sum of SR launches {sum(r['ms'] for r in rows):.3f} ms", flush=True)
    print("   heaviest launches: " + "; ".join(f"{r['op']} {r['ms']*1e3:.0f} us" for r in heavy), flush=True)
    print(f"   {len(big)} tensors >= 32 MB; address mod 2 MiB (KiB): {sorted({(t.data_ptr() % (2 << 20)) >> 10 for t in big})}; mod 1 GiB (MiB): {[ (t.data_ptr() % (1 << 30)) >> 20 for t in big][:12]}", flush=True)
    st = torch.cuda.memory_stats(dev)
    print(f"   allocator: reserved {st['reserved_bytes.all.current'] >> 20} MiB, active {st['active_bytes.all.current'] >> 20} MiB, segments {st['segment.all.current']}, large-pool segments {st['segment.large_pool.current']}", flush=True)
    if MODE == "no_del":
        keep.append(im)
    else:
        for u in im.unets:
            u.engine().invalidate()
        del im, ws, eng, big, rows
        if MODE == "empty_cache":
            torch.cuda.empty_cache()
