"""Find the first launch that produces a non-finite value when sampling with the default Unet() (eager steps, every workspace tensor checked
after every program entry).  usage: python tools/gpu_wide_nan_hunt.py [B] [T]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minimagen_amd import _lib as L, engine as E
from minimagen_amd.Imagen import Imagen
from minimagen_amd.Unet import Unet
from oracle import restated as R
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 25          # (T <= 20 is outside the reference's linear schedule: beta_end >= 1, diffusion_model.py:23)
dev = torch.device("cuda:0")
torch.manual_seed(6)
im = Imagen((Unet(),), text_encoder_name="t5_small", image_sizes=(64,), timesteps=T, cond_drop_prob=0.1).to(dev).eval()
emb, mask = R.synthetic_text(B, length=20, seed=8)
emb, mask = emb.to(dev), mask.to(dev)
state = {"step": 0, "found": False}
def run_step(self, ws, stream=None, t_off=0):
    st = L.current_stream() if stream is None else stream
    entries = self.stage_prog(ws, t_off) + ws.prog
    for i, (fn, p, name) in enumerate(entries):
        rc = fn(C.byref(p), st) if p is not None else fn(None, st)
        if rc != 0:
            L.check(rc, name)
        if not state["found"]:
            torch.cuda.synchronize()
            bad = [(j, tuple(t.shape)) for j, t in enumerate(ws.tensors) if isinstance(t, torch.Tensor) and t.is_floating_point() and not bool(torch.isfinite(t.float()).all())]
            if bad:
                state["found"] = True
                print(f"step {state['step']}: first non-finite after entry {i}/{len(entries)} '{name}': tensors {bad[:4]}", flush=True)
                if p is not None:
                    for f, _ in p._fields_:
                        v = getattr(p, f)
                        if isinstance(v, (int, float)) and v and not f.startswith(("w", "gn_", "bias", "out", "res_w", "res_b", "scale_shift")):
                            print("   ", f, v)
                        elif isinstance(v, L.MiAct) and v.data:
                            print("   ", f, "C", v.C, "nt", v.nt, "scale", v.scale)
                x = ws.x
                print("    |x_t| max", float(x.abs().max()), "finite", bool(torch.isfinite(x).all()))
    state["step"] += 1
E.UnetEngine.run_step = run_step
out = im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=10, _use_graph=False)
torch.cuda.synchronize()
print("finite:", bool(torch.isfinite(out).all()), "steps run", state["step"])
