#!/usr/bin/env python
"""bench.py -- MinImagen cascaded-diffusion sampling on MI355X (the BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full ``Imagen.sample()`` over one batch of synthetic text embeddings: the hot path named by
BASELINE.json (``Imagen.sample -> _p_sample_loop -> _p_sample -> Unet.forward_with_cond_scale``), i.e.
T denoising steps per stage, classifier-free guidance (2 U-Net evaluations per step), dynamic thresholding
and the posterior draw, all stages.  value = denoising-steps/s = (images x sum_stages T) / wall time, summed
over all ranks (weak scaling: the per-GPU batch is fixed, every rank samples its own rows, one RCCL
all_gather of the finished images inside the timed region).

Also printed in the same JSON line:
  roofline      the dominant kernel launch of the dominant stage, timed live with HIP events on the launch stream
  cpu_baseline  the oracle (a CPU port of the reference's algorithm) timed on this host's cores on a bounded sample
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the first HIP call (minimagen_amd sets the same default at import)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32, dense
MFMA_F16_PEAK_TFLOPS = 2500.0 # dense f16/bf16 MFMA peak (MI355X_MICROARCH.md; the sparsity figure is not used)


def synthetic_text(batch, length=64, dim=512, seed=7, row0=0):
    """randn(B,L,512) seed 7, row r keeps L-(r mod 24) leading tokens, masked rows zeroed (t5.py:82)."""
    gen = torch.Generator(device="cpu").manual_seed(seed)
    emb = torch.randn(row0 + batch, length, dim, generator=gen)[row0:]
    keep = torch.tensor([max(1, length - ((row0 + r) % 24)) for r in range(batch)])
    mask = torch.arange(length)[None, :] < keep[:, None]
    return emb.masked_fill(~mask[:, :, None], 0.).contiguous(), mask.contiguous()


def build_imagen(workload, timesteps, dev):
    from minimagen_amd.Imagen import Imagen
    from minimagen_amd.Unet import Unet
    p = json.load(open(os.path.join(ROOT, "tests", "golden", "unet_params.json")))   # = the reference's parameters/*.json
    torch.manual_seed(0)
    if workload == "base64":
        unets, sizes = [Unet(**p["unet0"])], (64,)
    elif workload == "cascade64_256_1024":      # BASELINE config 5's shape: the third stage reuses the unet_1 parameters (no 1024^2 config ships)
        unets, sizes = [Unet(**p["unet0"]), Unet(**p["unet1"]), Unet(**p["unet1"])], (64, 256, 1024)
    else:
        unets, sizes = [Unet(**p["unet0"]), Unet(**p["unet1"])], (64, 256)
    im = Imagen(unets, text_encoder_name="t5_small", image_sizes=sizes, timesteps=timesteps, cond_drop_prob=0.15)
    return im.to(dev), sizes


# ---- algorithmic bytes / flops of one program entry (SURVEY.md 8(d) definition: every Conv2d / Linear call counts
# input + output + weights&bias elements, an attention core counts q + k + v + out; norms/activations/adds count zero)
def entry_cost(name, p, cond_dim=8):
    """(algorithmic bytes, algorithmic flops, bound) of one launch.  Bytes follow the STORED element size of every tensor: 4 for fp32,
    2 where the reduced-precision configuration keeps the activation as bf16 (mi_act.st / out_st; SURVEY 8(d): "bf16 = half"); weights
    and biases are fp32 in both configurations."""
    if name == "conv":
        c0, c1 = p.in0.C, (p.in1.C if p.in1.data else 0)
        cin = c0 + c1
        hin, win = (p.H // 2, p.W // 2) if p.up2 else (p.H * p.stride, p.W * p.stride)
        e_out = 2 if p.out_st else 4
        by = p.B * hin * win * (c0 * (2 if p.in0.st else 4) + c1 * (2 if p.in1.st else 4)) + p.B * p.Cout * p.H * p.W * e_out \
            + (p.Cout * cin * p.ksize * p.ksize + p.Cout) * 4
        flops = 2.0 * p.B * p.H * p.W * p.Cout * cin * p.ksize * p.ksize
        if p.res0.data and p.res_w:
            r0, r1 = p.res0.C, (p.res1.C if p.res1.data else 0)
            by += p.B * p.H * p.W * (r0 * (2 if p.res0.st else 4) + r1 * (2 if p.res1.st else 4)) + p.B * p.Cout * p.H * p.W * e_out + (p.Cout * (r0 + r1) + p.Cout) * 4
            flops += 2.0 * p.B * p.H * p.W * p.Cout * (r0 + r1)
        return float(by), flops, "hbm"
    if name == "crossembed":
        cin = p.C0 + (p.C1 if p.in1 else 0)
        by, flops = 0, 0.0
        for i in range(p.n_kernels):
            k, co = p.ksize[i], p.cout[i]
            by += p.B * cin * p.H * p.W * 4 + p.B * co * p.H * p.W * (2 if p.out_st else 4) + (co * cin * k * k + co) * 4     # (the image itself is always fp32)
            flops += 2.0 * p.B * p.H * p.W * co * cin * k * k
        return float(by), flops, "hbm"
    if name == "cross_attn":
        inner, j, i = p.heads * 64, p.J, p.HW
        cd = cond_dim
        e = 2 if p.x.st else 4              # q / k / v / out and the projections' activations follow the activation storage type
        lin_act = (p.B2 * i * p.C + p.B2 * i * inner) + (p.B2 * (j - 1) * cd + p.B2 * (j - 1) * 2 * inner) + (p.B2 * i * inner + p.B2 * i * p.C)
        lin_w = inner * p.C + 2 * inner * cd + inner * p.C
        core = 2 * p.B2 * i * inner + 2 * p.B2 * j * inner
        flops = p.B2 * (4.0 * p.heads * i * j * 64 + 2 * 2.0 * i * p.C * inner)
        return float((lin_act + core) * e + lin_w * 4), flops, "mfma"
    return 0.0, 0.0, "hbm"


def op_breakdown(im, stage, B, cond_scale, reps=20, precision="fp32"):
    """Time every kernel launch of one U-Net evaluation of `stage` with HIP events on the launch stream, IN PROGRAM ORDER (each
    launch sees the cache state its predecessor leaves behind, as inside the captured graph), averaged over `reps` evaluations."""
    from minimagen_amd import _lib as L
    unet = im.unets[stage]
    S = im.image_sizes[stage]
    eng = unet.engine()
    ws = eng.workspace(B, 2 * B if cond_scale != 1 else B, S, S, precision=precision)
    stream = L.current_stream()
    prog = [(fn, p, name) for fn, p, name in ws.prog if p is not None]
    for _ in range(2):
        eng.run(ws, stream)
    acc = [0.0] * len(prog)
    for _ in range(reps):
        evs = []
        for fn, p, name in prog:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(C.byref(p), stream)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        for k, (e0, e1) in enumerate(evs):
            acc[k] += e0.elapsed_time(e1)
    rows = []
    for k, (fn, p, name) in enumerate(prog):
        ms = acc[k] / reps
        by, fl, bound = entry_cost(name, p, unet.cond_dim)
        desc = name
        if name == "conv":
            cin = p.in0.C + (p.in1.C if p.in1.data else 0)
            desc = f"conv k{p.ksize}s{p.stride}{'u' if p.up2 else ''} {cin}->{p.Cout} @{p.H}x{p.W} B{p.B}{' gn' if p.gn_groups else ''}{' res' if p.res0.data else ''}"
        elif name == "cross_attn":
            desc = f"cross_attn C{p.C} tokens{p.HW} ctx{p.J} B{p.B2}"
        elif name == "crossembed":
            desc = f"crossembed {p.C0 + (p.C1 if p.in1 else 0)}->{sum(p.cout[i] for i in range(p.n_kernels))} @{p.H}x{p.W} B{p.B}"
        rows.append(dict(op=desc, kernel=name, ms=ms, alg_bytes=by, alg_flops=fl, bound=bound,
                         rows=int(getattr(p, "B", 0) or getattr(p, "B2", 0))))
    # work the engine hoists out of the step (the low-res half of CrossEmbed, once per sample()) is still part of every forward of the
    # reference: its algorithmic bytes count, its time does not appear here
    for fn, p, name in getattr(ws, "prog_pre", []):
        by, fl, bound = entry_cost(name.replace("_lowres", ""), p, unet.cond_dim)
        rows.append(dict(op=name + " (hoisted: once per sample())", kernel=name, ms=0.0, alg_bytes=by, alg_flops=fl, bound=bound, rows=int(p.B), hoisted=True))
    return rows


def unet_prog(im, stage, B, args):
    S = im.image_sizes[stage]
    ws = im.unets[stage].engine().workspace(B, 2 * B if args.cond_scale != 1 else B, S, S, precision=args.precision)
    return [e for e in ws.prog if e[1] is not None]


def graph_step_ms(im, stage, B, cond_scale, T, reps=40, precision="fp32"):
    """Per-step time of the stage's captured HIP graph (U-Net evaluation of both guidance halves + CFG/x0 + quantile + posterior draw),
    replayed back to back exactly as sample() does -- the figure the end-to-end throughput is made of (no per-launch event overhead)."""
    from minimagen_amd import _lib as L
    lib = L.lib()
    unet = im.unets[stage]
    S = im.image_sizes[stage]
    ws = unet.engine().workspace(B, 2 * B if cond_scale != 1 else B, S, S, precision=precision)
    states = getattr(ws, "sampler_state", {})
    entries = [(st, e) for st in states.values() for e in getattr(st, "graphs", {}).values() if e.get("graph")]
    if not entries:
        return None
    st, entry = entries[-1]
    stream = L.current_stream()
    per = entry.get("per", 1)                      # denoising steps captured per graph
    reps = max(1, min(reps, T - 2 * per) // per)
    L.check(lib.mi_step_set(L.ptr(st.t_state), L.ptr(ws.times), B, T - 1, stream), "mi_step_set")
    L.check(lib.mi_graph_launch(entry["graph"], stream), "mi_graph_launch")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.mi_graph_launch(entry["graph"], stream), "mi_graph_launch")
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (reps * per)


def t5_leg(dev, B, Lq=64, reps=10):
    """SURVEY.md 8(d) T5 leg: T5EncoderModel(T5Config()) (= t5-small shape) with random-init weights seed 0, random ids, ragged masks,
    encoded by the HIP kernels of csrc/t5.hip (once per sample() when captions are given as text)."""
    from transformers import T5Config, T5EncoderModel
    from minimagen_amd.t5 import T5EncoderHIP
    torch.manual_seed(0)
    enc = T5EncoderHIP.from_hf(T5EncoderModel(T5Config()).eval(), device=dev)
    ids = torch.randint(0, 32128, (B, Lq), generator=torch.Generator().manual_seed(1))
    _, mask = synthetic_text(B, length=Lq)
    enc.encode(ids.to(dev), mask.to(dev))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        enc.encode(ids.to(dev), mask.to(dev))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    flops = B * Lq * 37.7e6 + 6 * B * 4 * 8 * Lq * Lq * 64
    return {"ms": ms, "tokens_per_s": B * Lq / (ms * 1e-3), "tflops_algorithmic": flops / (ms * 1e-3) / 1e12,
            "config": f"t5-small shape (d_model 512, 6 layers, 8 heads, d_ff 2048), B={B}, L={Lq}; fp32 in / out, GEMMs as 3-term fp16-split products on "
                      "v_mfma_f32_16x16x32_f16 with per-K-slice block scaling (fp32 accumulate), attention on fp32 MFMA; once per sample() when captions are text"}


def pmc_traffic(dom, rows, S):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of THIS command (profiles/rNN_*_pmc_by_launch_shape.csv:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH doubled per the gfx950 correction, see tools/summarize_profiles.py).
    null when no committed profile matches the launch shape."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cascade_T25_pmc_by_launch_shape.csv")))
    if not files or dom["kernel"] != "cross_attn":
        return {"traffic": None}
    grid = rows * (-(-(S // 4) * (S // 4) // 128)) * 512         # bottleneck level = S/4; 128 tokens per 512-thread workgroup (attention.hip launcher)
    for r in csv.DictReader(open(files[-1])):
        if "cross_attn" in r["kernel"] and int(r["grid"]) == grid and r["fetch_MB_x2"] not in ("", "None") and r["write_MB"] not in ("", "None"):
            return {"traffic": (float(r["fetch_MB_x2"]) + float(r["write_MB"])) * 1e6,
                    "traffic_source": f"committed-profile (not measured in this run): profiles/{os.path.basename(files[-1])}, "
                                      "rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per launch of this launch shape (bytes)"}
    return {"traffic": None}


def pmc_traffic_live(args, dom, rows, S):
    """HBM bytes per launch of the dominant kernel, MEASURED IN THIS RUN when rocprofv3 is on the box: two passes of a child process that
    runs the dominant stage's U-Net evaluation three times (`--pmc-child`), one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE -- each
    counter in its own pass, kernel trace only, FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request on wide coalesced reads), KiB ->
    bytes: MI355X_MICROARCH.md, HBM / rocprofv3 section.  None when rocprofv3 is missing or a pass fails (the caller then falls back to the
    committed profile, labelled as such)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if dom["kernel"] != "cross_attn" or not shutil.which("rocprofv3") or not args.live_pmc:
        return None
    grid = rows * (-(-(S // 4) * (S // 4) // 128)) * 512
    tmp = tempfile.mkdtemp(prefix="mi_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", args.workload, "--batch", str(args.batch), "--precision", args.precision,
             "--cond-scale", str(args.cond_scale), "--timesteps", str(args.timesteps)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["TMPDIR"] = "/tmp"
    got = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "--pmc", ctr, "-d", d, "-o", "pmc", "--"] + child,
                               capture_output=True, text=True, env=env, cwd="/tmp", timeout=300)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "cross_attn" in row["Kernel_Name"] and int(row["Grid_Size"]) == grid and row["Counter_Name"] == ctr:
                        vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or not vals:
                return None
            got[ctr] = sum(vals) / len(vals)
        return {"traffic": (2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024.0,
                "traffic_source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes over a child process that runs "
                                  "this stage's U-Net evaluation), per launch of this launch shape, FETCH_SIZE x 2 (gfx950 correction), KiB -> bytes",
                "traffic_fetch_bytes_x2": 2.0 * got["FETCH_SIZE"] * 1024.0, "traffic_write_bytes": got["WRITE_SIZE"] * 1024.0}
    except Exception as exc:
        print(f"bench: live PMC pass failed ({type(exc).__name__}: {exc}); using the committed profile", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_child(args):
    """what the live PMC passes profile: the last stage's U-Net evaluation (both guidance halves), three times"""
    from minimagen_amd import _lib as L
    dev = torch.device("cuda:0")
    L.use_library(L.DEFAULT_LIB)
    im, sizes = build_imagen(args.workload, args.timesteps, dev)
    stage, B, S = len(sizes) - 1, args.batch, sizes[-1]
    unet = im.unets[stage]
    eng = unet.engine()
    eng.pack()
    two = args.cond_scale != 1
    ws = eng.workspace(B, 2 * B if two else B, S, S, precision=args.precision)
    emb, mask = synthetic_text(B)
    keep = torch.cat((torch.ones(B, dtype=torch.bool), torch.zeros(B if two else 0, dtype=torch.bool)))
    ws.x.normal_()
    if ws.lowres is not None:
        ws.lowres.normal_()
        eng.prepare_lowres(ws)
    eng.set_text(ws, emb.to(dev), mask.to(dev), keep)
    for _ in range(3):
        eng.run(ws)
    torch.cuda.synchronize()


def cpu_baseline_reference(ncores):
    """Config 1 of BASELINE.json through the UNMODIFIED reference (oracle/ref_loader.py imports it from /root/reference; only where
    that tree exists, i.e. never on the GPU box): Imagen.sample, base U-Net 64x64, B=4, cond_scale 3, a bounded number of timesteps."""
    from oracle import ref_loader
    ref = ref_loader.load_reference()
    p = json.load(open(os.path.join(ROOT, "tests", "golden", "unet_params.json")))
    T = 25                                     # bounded sample: 25 of the reference's timesteps (its loop has no early exit)
    unet = ref.Unet(**p["unet0"])
    im = ref.Imagen(unets=[unet], text_encoder_name="t5_small", image_sizes=(64,), timesteps=T, cond_drop_prob=0.15)
    im.unets[0].load_state_dict(torch.load(os.path.join(ROOT, "tests", "golden", "unet0_sd.pt"), weights_only=False))
    emb, mask = synthetic_text(4)
    with torch.no_grad():
        im.sample(text_embeds=emb, text_masks=mask, cond_scale=3.)          # warm-up (thread pools, oneDNN primitives)
        t0 = time.time()
        im.sample(text_embeds=emb, text_masks=mask, cond_scale=3.)
        dt = time.time() - t0
    return dict(value=4 * T / dt, unit="denoising-steps/s", cores=ncores, kind="reference",
                sample=f"the reference's own Imagen.sample (/root/reference, CPU, torch {torch.__version__}): base U-Net (unet_0 params) 64x64, B=4, "
                       f"cond_scale=3 (BASELINE config 1), {T} timesteps in {dt:.1f}s on {ncores} host threads")


def cpu_baseline():
    """Config 1 of BASELINE.json on this host: base U-Net 64x64, B=4, T=100, cond_scale 3 -- the reference itself where its tree exists
    (the build container), the oracle (a CPU port of its algorithm) on the GPU box."""
    from oracle import restated as R
    ncores = min(os.cpu_count(), 16)     # tiny-channel convs do not scale further; more threads only add sync cost
    torch.set_num_threads(ncores)
    try:
        from oracle import ref_loader
        if ref_loader.reference_available():
            return cpu_baseline_reference(ncores)
    except Exception as e:                 # the port below is always available
        print(f"bench: reference CPU baseline unavailable ({type(e).__name__}: {e}); timing the oracle port", file=sys.stderr)
    sd = torch.load(os.path.join(ROOT, "tests", "golden", "unet0_sd.pt"), weights_only=False)
    emb, mask = synthetic_text(4)
    B, T = 4, 100
    sched = R.Schedule(T)
    randn = R.make_randn(1234)
    img = randn((B, 3, 64, 64))
    done, t0 = 0, time.time()
    for t in sched.sampling_timesteps():          # the reference's own loop order; stop after ~20 s of CPU work
        img, _ = R.p_sample(sd, sched, img, t, randn((B, 3, 64, 64)), text_embeds=emb, text_mask=mask, cond_scale=3.)
        done += 1
        if time.time() - t0 > 20.0:
            break
    dt = time.time() - t0
    # plus one SR U-Net forward (unet_1 params, lowres conditioning, 256x256) at B=4 on the same host threads
    sd1 = torch.load(os.path.join(ROOT, "tests", "golden", "unet1_sd.pt"), weights_only=False)
    x1, lr1 = randn((4, 3, 256, 256)), randn((4, 3, 256, 256))
    tm, lt = torch.full((4,), 50), torch.full((4,), 20)
    R.unet_forward(sd1, x1, tm, lowres_cond_img=lr1, lowres_noise_times=lt, text_embeds=emb, text_mask=mask)          # warm-up
    t1 = time.time()
    R.unet_forward(sd1, x1, tm, lowres_cond_img=lr1, lowres_noise_times=lt, text_embeds=emb, text_mask=mask)
    sr_ms = (time.time() - t1) * 1e3
    return dict(value=B * done / dt, unit="denoising-steps/s", cores=ncores, kind="port",
                sample=f"oracle/restated.py p_sample loop: base U-Net (unet_0 params) 64x64, B=4, cond_scale=3 (BASELINE config 1), "
                       f"{done} of {T} timesteps in {dt:.1f}s on {ncores} host threads",
                sr_unet_forward_B4_ms=sr_ms)


def timed_calls(im, emb, mask, cond_scale, precision, calls, warmup, pipelined, seed0=2):
    """seconds per sample() call: `warmup` untimed calls, then `calls` timed ones (the main loop's method)"""
    for k in range(warmup):
        im.sample(text_embeds=emb, text_masks=mask, cond_scale=cond_scale, _seed=seed0 + k, _precision=precision, _async=pipelined)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(calls):
        im.sample(text_embeds=emb, text_masks=mask, cond_scale=cond_scale, _seed=seed0 + warmup + k, _precision=precision, _async=pipelined)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / calls


def secondary_lines(timesteps, cond_scale):
    """The other single-GPU BASELINE.json configurations as extra keys of the driver line, each timed by the main loop of THIS script in a
    fresh process (3 warm-up calls, then 8 / 8 / 3 timed sample() calls, pipelined; plus the same calls one at a time): config 2 (base 64^2,
    B=32, fp32), config 3's shape (cascade 64->256, B=16, reduced precision) and config 5's per-GPU shape (cascade 64->256->1024, B=8,
    reduced precision, noise augmentation on both SR stages).  Never the headline.
    (A fresh process per configuration keeps each measurement independent of what ran before.  The slowdown of a second Imagen built in
    one process that motivated it in round 3 -- 261 instead of 150 ms per call -- is root-caused and fixed: HIP streams sharing a hardware
    queue, profiles/r04_second_instance_slowdown.txt.)  Config 5's entry also carries the per-launch breakdown summary of its 1024^2 stage
    (`unet_eval`, `roofline`) measured in that process."""
    import subprocess
    out = {}
    for key, workload, B, precision, calls in (("config2_base64_B32_fp32", "base64", 32, "fp32", 8),
                                               ("config3_cascade64_256_B16_half", "cascade64_256", 16, "half", 8),
                                               ("config5_cascade64_256_1024_B8_half", "cascade64_256_1024", 8, "half", 3)):
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--batch", str(B), "--precision", precision, "--steps", str(calls),
               "--warmup", "3" if calls > 3 else "1", "--timesteps", str(timesteps), "--cond-scale", str(cond_scale),
               "--no-secondary", "--no-cpu-baseline", "--no-t5", "--no-live-pmc"] + ([] if workload == "cascade64_256_1024" else ["--no-breakdown"])      # (the secondary legs keep the committed-profile look-up: bounded run time)
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            out[key] = {"error": (r.stderr or r.stdout)[-400:]}
            continue
        j = json.loads(lines[-1])
        n_img = B
        out[key] = {"denoising_steps_per_s": j["value"], "images_per_s": j["images_per_s"], "ms_per_sample_call": j["ms_per_step"],
                    "denoising_steps_per_s_no_pipeline": j.get("value_no_pipeline"), "ms_per_sample_call_no_pipeline": j.get("ms_per_step_no_pipeline"),
                    "pipelined_equals_synchronous": j.get("pipelined_equals_synchronous"), "timed_calls": calls, "per_gpu_batch": n_img,
                    "precision": precision, "image_sizes": {"base64": [64], "cascade64_256": [64, 256], "cascade64_256_1024": [64, 256, 1024]}[workload]}
        for extra in ("unet_eval", "roofline"):          # (config 5: the last stage's per-launch summary, measured in the same process)
            if extra in j:
                out[key][extra] = j[extra]
    return out


def train_step_leg():
    """SURVEY 8(f) rank 3 (not the headline): one training step (Imagen.forward -> loss.backward(), no optimiser) of the SR U-Net (unet_1
    parameters, lowres_cond, 256^2) at B = 32 on the device path (conv stack, CrossEmbed and the folded cross-attention core forward and
    backward on the HIP kernels, minimagen_amd/train_ops.py) with the same step on torch ops / MIOpen beside it: ms per step and peak
    allocator memory.  Runs in a fresh process (`--train-step-only`)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--train-step-only"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if (r.returncode == 0 and lines) else {"error": (r.stderr or r.stdout)[-400:]}
    except Exception as exc:                      # never let the secondary leg take the driver line down
        return {"error": repr(exc)[:400]}


def wide_step_leg():
    """SURVEY 8(f) rank 2 (not the headline): sampling with the reference's default ``Unet()`` (dim 128, channels 128 / 256 / 512, self- and
    cross-attention at every level: the wide regime of the kernels) at 64 x 64, B = 16 with guidance, T = 25: ms per denoising step through
    Imagen.sample's captured graphs.  Runs in a fresh process (`--wide-step-only`)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--wide-step-only"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if (r.returncode == 0 and lines) else {"error": (r.stderr or r.stdout)[-400:]}
    except Exception as exc:                      # never let the secondary leg take the driver line down
        return {"error": repr(exc)[:400]}


def wide_step_only():
    from minimagen_amd.Imagen import Imagen
    from minimagen_amd.Unet import Unet
    dev = torch.device("cuda:0")
    B, T = 16, 25
    torch.manual_seed(6)
    im = Imagen((Unet(),), text_encoder_name="t5_small", image_sizes=(64,), timesteps=T, cond_drop_prob=0.1).to(dev).eval()
    emb, mask = synthetic_text(B)
    emb, mask = emb.to(dev), mask.to(dev)
    for k in range(2):
        im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for k in range(n):
        out = im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=10 + k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    im.check_device_status()
    # roofline of this leg: the reference graph's algorithmic flops (every Conv2d / Linear / attention product of one image-forward of Unet(),
    # counted by torch's FlopCounterMode over the package's torch-op graph of the same modules at B = 1) x the 2B image-forwards of a guided
    # step / step time / the dense f16 matrix-core peak; each product is issued as three f16 MFMAs (hi*hi + hi*lo + lo*hi)
    roof = None
    try:
        from torch.utils.flop_counter import FlopCounterMode
        from minimagen_amd import train_ops
        hip_was, train_ops.ENABLED = train_ops.ENABLED, False
        with torch.no_grad(), FlopCounterMode(display=False) as fc:
            im.unets[0]._forward_train(torch.randn(1, 3, 64, 64, device=dev), torch.tensor([7], device=dev), text_embeds=emb[:1], text_mask=mask[:1])
        train_ops.ENABLED = hip_was
        gflop = fc.get_total_flops() / 1e9
        ach = gflop * 2 * B / (dt / T) / 1e3
        roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_F16_PEAK_TFLOPS,
                "issued_frac": 3 * ach / MFMA_F16_PEAK_TFLOPS, "alg_GFLOP_per_image_forward": gflop,
                "note": "whole guided denoising step (not one launch): algorithmic flops of 2B image-forwards of the reference graph / step time; issued = 3 f16 MFMA terms per product"}
    except Exception as exc:                      # the count is bookkeeping: never lose the timing over it
        roof = {"error": repr(exc)[:300]}
    print(json.dumps({"roofline": roof, "config": f"Unet() default (dim 128, dim_mults (1, 2, 4), attention at every level) @64x64, B={B}, cond_scale 3 (2 U-Net evaluations per step in one "
                                f"{2 * B}-row batch), T={T}, fp32 (3-term fp16-split MFMA products); 2 warm-up + {n} timed sample() calls",
                      "ms_per_denoising_step": dt / T * 1e3, "denoising_steps_per_s": B * T / dt, "finite": bool(torch.isfinite(out).all()),
                      "parameters_M": sum(p.numel() for p in im.parameters()) / 1e6}))


def train_step_only():
    from minimagen_amd import train_ops
    dev = torch.device("cuda:0")
    im, sizes = build_imagen("cascade64_256", 1000, dev)
    im.train()
    B, S, unet_number = 32, sizes[-1], 2
    imgs = torch.rand(B, 3, S, S, device=dev)
    emb, mask = synthetic_text(B)
    emb, mask = emb.to(dev), mask.to(dev)
    out = {"config": f"unet_1 params (lowres_cond) @{S}x{S}, B={B}, fp32: Imagen.forward (random timestep, q_sample, low-res augmentation, MSE on the noise) + loss.backward(); "
                     "3 warm-up + 10 timed steps per path; ms_per_fwd_bwd without, ms_per_step_with_clip_and_adam with the optimiser"}
    for hip in (False, True):
        train_ops.ENABLED = hip
        def step(seed):
            torch.manual_seed(seed)
            loss = im(imgs, text_embeds=emb, text_masks=mask, unet_number=unet_number)
            for prm in im.unets[unet_number - 1].parameters():
                prm.grad = None
            loss.backward()
            return loss
        for k in range(3):
            step(6 + k)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        for k in range(10):
            loss = step(20 + k)
        torch.cuda.synchronize()
        key = "hip_kernels" if hip else "torch_ops_miopen"
        out[key] = {"ms_per_fwd_bwd": (time.perf_counter() - t0) / 10 * 1e3, "peak_mem_MB": torch.cuda.max_memory_allocated() / 2 ** 20, "loss": float(loss)}
        # the whole optimisation step as the reference's loop runs it (training.py:363-377): + gradient-norm clip at 50 + Adam -- the one-launch
        # multi-tensor kernel (minimagen_amd.optim.Adam) on the device path, torch.optim.Adam on the torch-op path
        from minimagen_amd import optim as mi_optim
        params = list(im.unets[unet_number - 1].parameters())
        opt = mi_optim.Adam(params, lr=1e-6) if hip else torch.optim.Adam(params, lr=1e-6)

        def full_step(seed):
            loss = step(seed)
            torch.nn.utils.clip_grad_norm_(params, 50)
            opt.step()
            return loss
        for k in range(3):
            full_step(40 + k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(10):
            full_step(50 + k)
        torch.cuda.synchronize()
        out[key]["ms_per_step_with_clip_and_adam"] = (time.perf_counter() - t0) / 10 * 1e3
    out["speedup_vs_torch_ops"] = out["torch_ops_miopen"]["ms_per_fwd_bwd"] / out["hip_kernels"]["ms_per_fwd_bwd"]
    print(json.dumps(out))


def main():
    if "--train-step-only" in sys.argv:
        return train_step_only()
    if "--wide-step-only" in sys.argv:
        return wide_step_only()
    pmc_child_mode = "--pmc-child" in sys.argv
    if pmc_child_mode:
        sys.argv.remove("--pmc-child")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cascade64_256", choices=["cascade64_256", "base64", "cascade64_256_1024"])
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--timesteps", type=int, default=100)
    ap.add_argument("--cond-scale", type=float, default=3.0)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "half"],
                    help="fp32 (default, the headline): every contraction fp32-grade; half: single-fp16-term matrix-core contractions "
                         "(BASELINE's reduced-precision configurations; parity gate 3e-2) -- reported as a secondary line, never the headline")
    ap.add_argument("--no-t5", action="store_true", help="skip the T5 text-embedding pass (K16) for the batch (timed by default, outside the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-live-pmc", dest="live_pmc", action="store_false", help="roofline.traffic from the committed PMC profile instead of two rocprofv3 passes in this run")
    ap.add_argument("--no-pipeline", action="store_true", help="make every sample() call wait for the previous one (no cross-call stage pipelining)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of the other single-GPU BASELINE configurations")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling (BASELINE config 4: fixed total batch, e.g. 128, sharded over the ranks; overrides --batch). "
                         "Default 0 = weak scaling with --batch rows per GPU")
    ap.add_argument("--breakdown-out", default="")
    ap.add_argument("--lanes", type=int, default=0, help="independent call lanes of the pipelined mode (sets of stage streams / workspaces / graphs that "
                    "sample(_async=True) alternates between; 0 = the library default MINIMAGEN_SAMPLE_LANES = 2, 1 = one lane: only the stages of successive calls overlap)")
    args = ap.parse_args()
    if pmc_child_mode:
        return pmc_child(args)

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU path)"
    # MINIMAGEN_BENCH_ONE_GPU=1 is a control-flow smoke test only (no valid number): every rank on cuda:0, collectives through gloo on
    # host copies -- lets the N > 1 path run on a single-GPU box (RCCL refuses two ranks on one device)
    one_gpu = os.environ.get("MINIMAGEN_BENCH_ONE_GPU", "0") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    # MINIMAGEN_DIST_SINGLE=1 (tests): take the N > 1 control flow -- RCCL process group, barriers, the all_gather on its own stream, the
    # max-over-ranks all_reduce -- with a group of ONE rank, so that the single-GPU test tier runs every RCCL call of this script
    dist_on = world > 1 or os.environ.get("MINIMAGEN_DIST_SINGLE", "0") == "1"
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from minimagen_amd import _lib as L
    from minimagen_amd.distributed import gather_samples, shard_bounds
    L.use_library(L.DEFAULT_LIB)
    import minimagen_amd.Imagen as MI
    if args.lanes > 0:
        MI.SAMPLE_LANES = args.lanes
    im, sizes = build_imagen(args.workload, args.timesteps, dev)
    strong = args.global_batch > 0
    if strong:
        gB = args.global_batch
        row0, row1 = shard_bounds(gB, world, rank)          # the partitioning of minimagen_amd.distributed.sample_distributed
        B = row1 - row0
        assert B > 0, "more ranks than samples"
    else:
        B = args.batch
        gB, row0 = B * world, rank * B
    emb, mask = synthetic_text(B, row0=row0)
    emb, mask = emb.to(dev), mask.to(dev)

    gstream = torch.cuda.Stream(device=dev) if (dist_on and not one_gpu) else None

    def one_step(k, pipelined=True, gather=True):
        # successive sample() calls are pipelined across the per-stage HIP streams (_async: the caller's stream is not made to wait; the
        # timed region ends with a device-wide synchronize, so every image is finished inside it)
        out = im.sample(text_embeds=emb, text_masks=mask, cond_scale=args.cond_scale, _seed=1234 + k, _sample_offset=row0, _precision=args.precision,
                        _async=pipelined)
        if dist_on and gather and one_gpu:
            if pipelined:
                torch.cuda.current_stream().wait_event(im.last_sample_done)
            host = out.cpu()
            pad = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(pad, host)
            out = torch.cat(pad, 0).to(dev)
        elif dist_on and gather:
            # RCCL over xGMI, the only collective of the path: one all_gather_into_tensor of the finished images, on its own stream behind
            # the last stage's event -- the caller's stream stays free, so the next call's base stage still starts under this call's SR stage
            if pipelined:
                gstream.wait_event(im.last_sample_done)
            else:
                gstream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(gstream):
                out.record_stream(gstream)
                out = gather_samples(out, gB)
        return out

    def timed(steps, pipelined):
        """(max-over-ranks seconds, this rank's own seconds, last output): barrier + synchronize on both sides of exactly `steps` steps"""
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            out = one_step(100 + k, pipelined)
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist_on:
            tt = torch.tensor([dt], device="cpu" if one_gpu else dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, mine, out

    pipelined = not args.no_pipeline
    for k in range(args.warmup):
        one_step(k, pipelined)
    # the benched mode is value-checked outside the timed region: the second of two back-to-back pipelined calls (its base stage ran
    # under the first call's super-resolution stage) against the same call made synchronously
    pipe_ok = None
    if pipelined:
        torch.cuda.synchronize()
        ref_out = one_step(7, False, gather=False).clone()
        torch.cuda.synchronize()
        one_step(6, True, gather=False)
        chk = one_step(7, True, gather=False)
        torch.cuda.synchronize()
        pipe_ok = bool(torch.equal(chk, ref_out))
        if not pipe_ok:
            ne = chk != ref_out
            nan_both = torch.isnan(chk) & torch.isnan(ref_out)
            where = ne.nonzero()[:4].tolist()
            raise AssertionError(f"pipelined sample() differs from the synchronous call: {int(ne.sum())} of {ne.numel()} elements ({int((ne & ~nan_both).sum())} not NaN in both), "
                                 f"per image {ne.flatten(1).sum(1).tolist()}, max |d| {float((chk - ref_out).abs().nan_to_num().max()):.3e}, first at {where}")
        del ref_out, chk
    dt, mine, out = timed(args.steps, pipelined)
    assert torch.isfinite(out).all() and out.shape[0] == gB
    im.check_device_status()            # no kernel with inter-workgroup waits (the grouped sampler tail) gave up waiting
    dt_sync = dt_one = None
    if pipelined:                       # the reference's sample() is synchronous: report that mode beside the pipelined headline
        dt_sync, _, _ = timed(max(2, min(args.steps, 8)), False)
        dt_sync /= max(2, min(args.steps, 8))
        if MI.SAMPLE_LANES > 1:         # ... and the pipelined mode with ONE lane (stage overlap of successive calls only)
            lanes = MI.SAMPLE_LANES
            MI.SAMPLE_LANES = 1
            one_step(90, True); one_step(91, True)
            dt_one, _, _ = timed(max(2, min(args.steps, 8)), True)
            dt_one /= max(2, min(args.steps, 8))
            MI.SAMPLE_LANES = lanes
    per_rank = None
    gather_ms = None
    if dist_on:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "rows": B, "seconds": mine})
        if not one_gpu:                 # the collective alone, on its stream, with HIP events
            loc = torch.zeros(B, 3, sizes[-1], sizes[-1], device=dev)
            with torch.cuda.stream(gstream):
                gather_samples(loc, gB)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    gather_samples(loc, gB)
                e1.record()
            e1.synchronize()
            gather_ms = e0.elapsed_time(e1) / 5

    n_stages = len(sizes)
    steps_per_sample = args.timesteps * n_stages
    value = gB * steps_per_sample * args.steps / dt
    res = {
        "metric": {1: "denoising-steps/sec (images/sec x T), base 64^2", 2: "denoising-steps/sec (images/sec x T), base 64^2 + SR 64->256 cascade",
                   3: "denoising-steps/sec (images/sec x T), base 64^2 + SR 64->256 + SR 256->1024 cascade"}[n_stages],
        "value": value, "unit": "denoising-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "f16 operands on the matrix cores, f32 accumulate/softmax/statistics/storage", "data": "synthetic (random-init weights seed 0, randn text embeddings seed 7 with ragged masks, Philox noise)",
        "config": {"workload": f"{args.workload}: unet_0 params @64x64" + (" + unet_1 params (lowres_cond) @256x256" if n_stages >= 2 else "") + (" + unet_1 params (lowres_cond) @1024x1024" if n_stages == 3 else "")
                   + f", T={args.timesteps}/stage, cond_scale={args.cond_scale} (2 U-Net evals/step), dynamic thresholding 0.9, " + ("fp32 storage / accumulate / softmax / statistics, contractions as 3-term fp16-split (hi*hi + hi*lo + lo*hi) MFMA products"
                      if args.precision == "fp32" else "half-precision matrix-core contractions (single fp16 term)"),
                   "per_gpu_batch": B if not strong else f"{gB}/{world} (fixed global batch, contiguous shards)", "global_batch": gB,
                   "timesteps": args.timesteps, "parallelism": f"dp{world}",
                   "call_mode": (f"successive sample() calls PIPELINED (_async=True) over {MI.SAMPLE_LANES} call lane(s) of per-stage HIP streams: "
                                 "two calls are in flight side by side and the base stage of a call runs under the super-resolution stage of the one "
                                 "before it; outputs bit-identical to synchronous calls, checked in this run; value_one_lane = stage overlap of "
                                 "successive calls only (round 2's mode), value_no_pipeline = the same calls one at a time, the reference's synchronous semantics") if pipelined
                                else "synchronous sample() calls, one at a time (the reference's semantics)"},
        "images_per_s": gB * args.steps / dt,
        "pipelined": pipelined,
    }
    if pipelined:
        res["pipelined_equals_synchronous"] = pipe_ok
        res["value_no_pipeline"] = gB * steps_per_sample / dt_sync
        res["ms_per_step_no_pipeline"] = dt_sync * 1e3
        res["lanes"] = MI.SAMPLE_LANES
        if dt_one is not None:
            res["value_one_lane"] = gB * steps_per_sample / dt_one
            res["ms_per_step_one_lane"] = dt_one * 1e3
    if dist_on:
        res["per_rank"] = [dict(r, denoising_steps_per_s=r["rows"] * steps_per_sample * args.steps / r["seconds"]) for r in per_rank]
        res["all_gather_ms"] = gather_ms

    if rank == 0 and not args.no_breakdown:
        stage = n_stages - 1
        rows = op_breakdown(im, stage, B, args.cond_scale, precision=args.precision)
        total_ms = sum(r["ms"] for r in rows)
        dom = max(rows, key=lambda r: r["ms"])
        attn_f16 = True                 # the folded attention runs on the fp16x3-split matrix-core kernel (the fp32-MFMA variants are gone)
        if dom["bound"] == "hbm":
            ach, peak, unit = dom["alg_bytes"] / (dom["ms"] * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            peak = MFMA_F16_PEAK_TFLOPS if (dom["kernel"] == "cross_attn" and attn_f16) else MFMA_F32_PEAK_TFLOPS
            ach, unit = dom["alg_flops"] / (dom["ms"] * 1e-3) / 1e12, "TFLOP/s"
        res["roofline"] = {"bound": dom["bound"], "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "traffic": None,
                           "kernel": dom["op"], "kernel_ms": dom["ms"], "stage": f"stage {stage} U-Net evaluation ({sizes[stage]}x{sizes[stage]})"}
        if dom["kernel"] == "cross_attn":
            # `achieved` prices the launch at the reference's ALGORITHMIC flops (q/k/v/out at 512 wide).  The folded
            # kernel needs 4x fewer multiply-adds (K = C = 16 instead of dim_head = 64); the fp16x3 variant issues each of them as
            # three f16 MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulate) to keep fp32-level accuracy -- report the issued rate too
            k = 0.25 * ((3.0 if args.precision == "fp32" else 1.0) if attn_f16 else 1.0)
            ex = dom["alg_flops"] * k / (dom["ms"] * 1e-3) / 1e12
            res["roofline"]["executed"] = {"achieved": ex, "frac": ex / peak,
                                           "note": f"MFMA flops actually issued = {k:g} x algorithmic (folded attention" + ((", fp16x3 split" if args.precision == "fp32" else ", single fp16 term") + "; v_mfma_f32_16x16x32_f16 for both contractions: QK^T per tile as {G hi|G lo}.{x hi|x hi} + {G hi|G lo}.{x lo|0} -- two pipe slots instead of three K = 16 ones, a quarter of the issued K is zero padding and not counted" if attn_f16 else "; v_mfma_f32_16x16x4_f32") + ")"}
        live = pmc_traffic_live(args, dom, B * (2 if args.cond_scale != 1 else 1), sizes[stage]) if world == 1 else None
        res["roofline"].update(live if live is not None else pmc_traffic(dom, B * (2 if args.cond_scale != 1 else 1), sizes[stage]))
        # SURVEY 8(d) algorithmic bytes of ONE image-forward, summed over this U-Net's own launch plan: every entry's bytes divided by the
        # rows it serves (a tensor the engine computes once for both guidance halves still counts once per forward, as in the reference)
        alg_fwd_mb = sum(r["alg_bytes"] / r["rows"] for r in rows if r["rows"]) / 1e6
        survey_mb = {("cascade64_256", 256): 124.97, ("base64", 64): 28.82}.get((args.workload, sizes[stage]))
        nfwd = 2 if args.cond_scale != 1 else 1
        conv_rows = [r for r in rows if r["kernel"] == "conv" and not r.get("hoisted")]
        by_level = {}
        for r in conv_rows:
            lv = by_level.setdefault(r["op"].split("@")[1].split()[0], {"launches": 0, "ms": 0.0, "alg_GB": 0.0})
            lv["launches"] += 1; lv["ms"] += r["ms"]; lv["alg_GB"] += r["alg_bytes"] / 1e9
        for lv in by_level.values():
            lv["alg_GBps"] = lv["alg_GB"] / (lv["ms"] * 1e-3) if lv["ms"] else None
        conv_ms, conv_gb = sum(r["ms"] for r in conv_rows), sum(r["alg_bytes"] for r in conv_rows) / 1e9
        res["unet_eval"] = {"stage": stage, "sum_kernel_ms": total_ms, "launches": sum(1 for r in rows if not r.get("hoisted")),
                            # the conv family on ITS OWN bytes (the whole-forward fraction below is carried by the attention's algorithmic
                            # q / k / v / projection tensors, which the folded kernel never materialises)
                            "conv_only": {"launches": len(conv_rows), "ms": conv_ms, "alg_GB": conv_gb, "alg_GBps": conv_gb / (conv_ms * 1e-3) if conv_ms else None,
                                          "hbm_frac": conv_gb / (conv_ms * 1e-3) / HBM_PEAK_GBS if conv_ms else None, "by_level": by_level},
                            "activation_bytes": "bf16 storage (2 B per activation element) where mi_act.st is set" if any(getattr(p, "out_st", 0) for _, p, _ in unet_prog(im, stage, B, args)) else "fp32 (4 B per activation element)",
                            "alg_bytes_MB_per_image_forward": alg_fwd_mb, "alg_bytes_MB_per_image_forward_SURVEY_8d": survey_mb,
                            "hbm_frac_whole_forward": (alg_fwd_mb * 1e6 * B * nfwd / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if alg_fwd_mb else None,
                            "by_kernel_ms": {k: sum(r["ms"] for r in rows if r["kernel"] == k) for k in sorted({r["kernel"] for r in rows})}}
        with torch.cuda.stream(im._stream):          # the stream sample() captured and replays on
            gs = graph_step_ms(im, stage, B, args.cond_scale, args.timesteps, precision=args.precision)
        torch.cuda.synchronize()
        if gs is not None and alg_fwd_mb:
            epi_mb = 6 * 3 * sizes[stage] ** 2 * 4 / 1e6                 # sampler epilogue bytes per image per step
            res["unet_eval"]["graph_step_ms"] = gs
            res["unet_eval"]["hbm_frac_graph_step"] = (alg_fwd_mb * nfwd + epi_mb) * 1e6 * B / (gs * 1e-3) / 1e9 / HBM_PEAK_GBS
            res["unet_eval"]["note"] = ("sum_kernel_ms adds per-launch HIP-event intervals (program order, event overhead included); "
                                        "graph_step_ms is one replay of the captured denoising step (U-Net x2 + CFG + quantile + posterior)")
        # the north-star quantities as SCALARS inside `roofline` (the driver's record keeps the scalars of this dict, and only the key names
        # of the nested extras): HBM fraction of the SR forward by SURVEY 8(d)'s bytes over one replay of the captured step, the conv family
        # on its own bytes, and the throughput of synchronous calls (the reference's call semantics) next to the pipelined headline
        res["roofline"]["hbm_frac_sr_forward"] = res["unet_eval"].get("hbm_frac_graph_step")
        res["roofline"]["graph_step_ms"] = res["unet_eval"].get("graph_step_ms")
        res["roofline"]["conv_only_hbm_frac"] = res["unet_eval"]["conv_only"]["hbm_frac"]
        res["roofline"]["conv_only_ms"] = conv_ms
        if pipelined:
            res["roofline"]["value_sync"] = res["value_no_pipeline"]
            if dt_one is not None:
                res["roofline"]["value_one_lane"] = res["value_one_lane"]
        if args.breakdown_out:
            with open(args.breakdown_out, "w") as f:
                json.dump(rows, f, indent=1)
    if rank == 0 and not args.no_t5:
        res["t5_encode"] = t5_leg(dev, B)
    if rank == 0 and world == 1 and not args.no_secondary and args.workload == "cascade64_256" and args.precision == "fp32":
        res["secondary"] = secondary_lines(args.timesteps, args.cond_scale)
        res["secondary"]["train_step_sr_unet_B32"] = train_step_leg()
        res["secondary"]["wide_unet_default_B16"] = wide_step_leg()
        wide = res["secondary"]["wide_unet_default_B16"]
        if "roofline" in res and isinstance(wide.get("roofline"), dict) and "frac" in wide["roofline"]:
            res["roofline"]["wide_unet_default_step_ms"] = wide.get("ms_per_denoising_step")
            res["roofline"]["wide_unet_default_mfma_frac"] = wide["roofline"]["frac"]
            res["roofline"]["wide_unet_default_mfma_issued_frac"] = wide["roofline"]["issued_frac"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(res))
    if dist_on:
        dist.barrier()               # rank 0 did the (untimed) per-kernel breakdown: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
