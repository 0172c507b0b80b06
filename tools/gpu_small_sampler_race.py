"""Isolate: mi_sampler_step_small_fwd (one 1024-thread workgroup per image) on fixed inputs while another stream keeps the GPU busy."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from minimagen_amd import _lib as L
from minimagen_amd.helpers import quantile_rank
from minimagen_amd.diffusion_model import GaussianDiffusion

dev = torch.device("cuda:0")
lib = L.lib()
B, n, T = 32, 3 * 64 * 64, 100
g = torch.Generator().manual_seed(0)
pred = torch.randn(2 * B, n, generator=g).to(dev)
x_in = torch.randn(B, n, generator=g).to(dev)
sched = GaussianDiffusion(timesteps=T)
coef = sched.sampler_coef_table().to(dev).contiguous()
t_state = torch.full((1,), 37, dtype=torch.int32, device=dev)
k_lo, k_hi, w = quantile_rank(n, 0.9)
x = torch.empty_like(x_in)
s_q, v_q = torch.zeros(B, device=dev), torch.zeros(B, 2, device=dev)
seed_dev = torch.full((1,), 1234, dtype=torch.int64, device=dev)
cp = L.MiCfgX0Params(B, n, L.ptr(pred), 1, 3.0, L.ptr(x), L.ptr(coef), L.ptr(t_state), 0, 0, 0)
qp = L.MiQuantileParams(B, n, 0, k_lo, k_hi, w, 0, L.ptr(s_q), L.ptr(v_q), 1, 1)
INJECT = os.environ.get("INJECT", "0") == "1"
noise = torch.randn(T, B, n, generator=g).to(dev) if INJECT else None
pp = L.MiPosteriorParams(B, n, T, 0, L.ptr(s_q), L.ptr(x), L.ptr(coef), L.ptr(t_state), L.ptr(noise), 1234, 0, 0, 0 if INJECT else L.ptr(seed_dev))
print("noise:", "injected tensor (no Philox / Box-Muller in the kernel)" if INJECT else "Philox + Box-Muller in the kernel", flush=True)

main = torch.cuda.Stream(device=dev)
SENT = 1.0e6
def run_once():
    with torch.cuda.stream(main):
        x.fill_(SENT)             # a stale read of x inside the kernel would see this ...
        x.copy_(x_in)             # ... a lost / stale write of the kernel's result would leave this
        L.check(lib.mi_sampler_step_small_fwd(C.byref(cp), C.byref(qp), C.byref(pp), main.cuda_stream), "small")
        return x.clone(), s_q.clone(), v_q.clone()
torch.cuda.synchronize()
rx, rs, rv = run_once()
torch.cuda.synchronize()
for _ in range(20):
    ax, as_, av = run_once()
    torch.cuda.synchronize()
    assert torch.equal(ax, rx) and torch.equal(as_, rs) and torch.equal(av, rv)
print("idle: 20 repeats bit-identical", flush=True)

im2, _ = bench.build_imagen("cascade64_256", T, dev)
eng = im2.unets[1].engine(); eng.pack()
ws = eng.workspace(B, 2 * B, 256, 256)
emb, mask = bench.synthetic_text(B)
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    eng.set_text(ws, emb.to(dev), mask.to(dev), torch.cat((torch.ones(B, dtype=torch.bool), torch.zeros(B, dtype=torch.bool))))
    eng.prepare_lowres(ws)
torch.cuda.synchronize()
LOAD = os.environ.get("LOAD", "all")
entries = [(fn, p_, name) for fn, p_, name in ws.prog if p_ is not None and (LOAD == "all" or name == LOAD or (LOAD.startswith("#") and False))]
if LOAD.startswith("conv@"):          # one conv launch shape, e.g. conv@64 = the 64x64 level's convs
    side_px = int(LOAD.split("@")[1])
    entries = [(fn, p_, name) for fn, p_, name in ws.prog if p_ is not None and name == "conv" and p_.H == side_px]
print(f"load: {LOAD} ({len(entries)} program entries per pass)", flush=True)
def load_pass(reps):
    with torch.cuda.stream(side):
        if LOAD == "all":
            for _ in range(reps):
                eng.run(ws)
        else:
            for _ in range(reps * max(1, 34 // max(1, len(entries)))):
                for fn, p_, name in entries:
                    fn(C.byref(p_), side.cuda_stream)
bad_x = bad_s = bad_v = 0
for rep in range(6):
    load_pass(30)
    outs = [run_once() for _ in range(150)]
    torch.cuda.synchronize()
    for ax, as_, av in outs:
        bx = (ax != rx).flatten(1).any(1); bs = as_ != rs; bv = (av != rv).any(1)
        bad_x += int(bx.sum()); bad_s += int(bs.sum()); bad_v += int(bv.sum())
        if bx.any() and bad_x <= 6:
            r = int(torch.nonzero(bx)[0])
            idx = torch.nonzero(ax[r] != rx[r]).flatten()
            vals = ax[r][idx]
            if INJECT and bad_x <= 2:
                tt = int(t_state.item()); cf = coef[tt].double().cpu()
                ca_, cb_, c1_, c2_, sg_ = [float(v) for v in cf[:5]]
                pc, pn, xt = pred[r].double().cpu(), pred[B + r].double().cpu(), x_in[r].double().cpu()
                pr_ = pn + (pc - pn) * 3.0
                x0_ = ca_ * xt - cb_ * pr_
                sv = max(1.0, float(rs[r]))
                x0c = x0_.clamp(-sv, sv) / sv
                zz = noise[T - 1 - tt, r].double().cpu()
                ii = idx.cpu()
                got, exp = ax[r].double().cpu()[ii], rx[r].double().cpu()[ii]
                cand = {"expected (host fp64)": c1_ * x0c[ii] + c2_ * xt[ii] + sg_ * zz[ii],
                        "x0 = 0": c2_ * xt[ii] + sg_ * zz[ii], "no noise": c1_ * x0c[ii] + c2_ * xt[ii],
                        "xt = 0": c1_ * x0c[ii] + sg_ * zz[ii], "unclamped x0 / s": c1_ * (x0_[ii] / sv) + c2_ * xt[ii] + sg_ * zz[ii],
                        "x0 not divided": c1_ * x0_[ii].clamp(-sv, sv) + c2_ * xt[ii] + sg_ * zz[ii], "x0 raw": c1_ * x0_[ii] + c2_ * xt[ii] + sg_ * zz[ii],
                        "pred_c only": c1_ * ((ca_ * xt[ii] - cb_ * pc[ii]).clamp(-sv, sv) / sv) + c2_ * xt[ii] + sg_ * zz[ii],
                        "pred_n only": c1_ * ((ca_ * xt[ii] - cb_ * pn[ii]).clamp(-sv, sv) / sv) + c2_ * xt[ii] + sg_ * zz[ii]}
                for name, v in cand.items():
                    print(f"     candidate {name:24s}: max|got - cand| = {(got - v).abs().max().item():.3e}", flush=True)
                for j in range(16):
                    q_ = ii[j].item()
                    alt = [k for k in range(max(0, q_ - 8192), min(n, q_ + 8192)) if abs(c1_ * x0c[k] + c2_ * xt[k] + sg_ * zz[k] - got[j]) < 1e-5]
                    print(f"     idx {q_}: got {got[j].item():+.6f} expected {exp[j].item():+.6f}  indices whose correct result equals got: {alt[:6]}", flush=True)
            print(f"  row {r}: mismatching indices {idx[0].item()}..{idx[-1].item()} (count {idx.numel()}, contiguous {bool((idx[-1] - idx[0] + 1) == idx.numel())}, byte offset of first % 128 = {(r * n + idx[0].item()) * 4 % 128}); "
                  f"equals x_in there: {bool(torch.equal(vals, x_in[r][idx]))}; |value| max {vals.abs().max().item():.3e} (sentinel-driven if ~1e5+)", flush=True)
            print(f"  row {r}: s {as_[r].item():.6f} vs {rs[r].item():.6f}; v {av[r].tolist()} vs {rv[r].tolist()}; x mismatching elements {(ax[r] != rx[r]).sum().item()} of {n}, max|d| {(ax[r]-rx[r]).abs().max().item():.3e}", flush=True)
print(f"under load: of {6*150*B} image-steps: x wrong {bad_x}, s wrong {bad_s}, order statistics wrong {bad_v}", flush=True)

# in-kernel self-check of the packed product (library built with -DSS_PKCHECK, profiles/r04_pk_f32_hazard_bisect.txt)
if hasattr(lib, "mi_debug_read_ss"):
    import numpy as np
    rec = np.zeros(256 * 8, dtype=np.float32)
    cnt = C.c_uint(0)
    lib.mi_debug_read_ss.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
    lib.mi_debug_read_ss(rec.ctypes.data, C.byref(cnt))
    print(f"in-kernel check: {cnt.value} packed products differ from the scalar product of the same registers", flush=True)
    for r in rec.reshape(256, 8)[:min(cnt.value, 24)]:
        print(f"   image {int(r[0])} work-item {int(r[1])} (lane {int(r[1]) & 63}, wave {int(r[1]) >> 6}) element {int(r[2])}: " + " ".join(f"{v:+.6f}" for v in r[3:]), flush=True)
