"""Tail of the SR U-Net (final ResnetBlock.block2 + 1x1 residual, then final_conv): two mi_conv_fwd launches against the fused
mi_conv_tail_fwd.  usage: python tools/bench_tail.py [B H W [strip]]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minimagen_amd import _lib as L          # noqa: E402
from minimagen_amd import packing as P       # noqa: E402

a = sys.argv[1:]
B, H, W = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (64, 256, 256)
strip = int(a[3]) if len(a) > 3 else 0
dev = torch.device("cuda:0")
L.use_library(os.environ.get("MINIMAGEN_HIP_LIB", L.DEFAULT_LIB))
lib = L.lib()
g = torch.Generator().manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g)


def stats(x):          # [B][C][1][2] sum / sum of squares
    return torch.stack((x.sum((2, 3)), (x * x).sum((2, 3))), -1).unsqueeze(2).contiguous()


keep = {}
d = lambda k, t: keep.setdefault(k, t.to(dev).contiguous())
h1, r0, r1 = rn(B, 8, H, W), rn(B, 8, H, W), rn(B, 8, H, W)
w, bias, gamma, beta = rn(8, 8, 3, 3) * 0.2, rn(8), 1 + 0.2 * rn(8), 0.1 * rn(8)
rw, rb, w2, b2 = rn(8, 16, 1, 1) * 0.3, rn(8), rn(3, 8, 3, 3) * 0.2, rn(3)
sst = rn(B, 16) * 0.3
tp = L.MiConvTailParams()
p = tp.conv
p.B, p.H, p.W = B, H, W
p.in0 = L.MiAct(d("h1", h1).data_ptr(), 8, d("s0", stats(h1)).data_ptr(), 1, 1.0, 0)
p.Cout, p.ksize, p.stride, p.up2 = 8, 3, 1, 0
wf, wexp = P.pack_conv_weight_rp(w)
p.w_rp, p.w_rp_exp, p.bias = d("wf", wf).data_ptr(), wexp, d("b", bias).data_ptr()
p.gn_groups, p.gn_gamma, p.gn_beta, p.gn_eps = 8, d("g", gamma).data_ptr(), d("be", beta).data_ptr(), 1e-5
p.scale_shift, p.ss_stride, p.ss_off = d("ss", sst).data_ptr(), 16, 0
p.res0 = L.MiAct(d("r0", r0).data_ptr(), 8, d("rs0", stats(r0)).data_ptr(), 1, 1.0, 0)
IDENT = os.environ.get("TAIL_RES", "id") == "id"   # ResnetBlock(dim, dim): identity residual (default); else the 1x1 conv over 16 channels
if not IDENT:
    p.res1 = L.MiAct(d("r1", r1).data_ptr(), 8, d("rs1", stats(r1)).data_ptr(), 1, 2 ** -0.5, 0)
rwf, rwexp = P.pack_conv_weight_rp(torch.eye(8).reshape(8, 8, 1, 1) if IDENT else rw)
d("rwf", rwf); d("rb", rb)


def set_res(fused_form):
    if IDENT and not fused_form:          # the separate launch adds the fp32 residual itself
        p.res_w, p.res_w_rp, p.res_w_rp_exp, p.res_b = 0, 0, 0, 0
    else:
        p.res_w, p.res_w_rp, p.res_w_rp_exp, p.res_b = 1, keep["rwf"].data_ptr(), rwexp, (0 if IDENT else keep["rb"].data_ptr())
w2f, w2exp = P.pack_conv_weight_rp(w2)
out2 = torch.zeros(B, 3, H, W, device=dev)
tp.w2_rp, tp.w2_rp_exp, tp.Cout2, tp.bias2, tp.out2 = d("w2f", w2f).data_ptr(), w2exp, 3, d("b2", b2).data_ptr(), out2.data_ptr()
th, tw = C.c_int(), C.c_int()
lib.mi_conv_tile_shape(5, C.byref(th), C.byref(tw))
nt = -(-H // th.value) * -(-W // tw.value)
mid, mst, sep = torch.zeros(B, 8, H, W, device=dev), torch.zeros(B, 8, nt, 2, device=dev), torch.zeros(B, 3, H, W, device=dev)
q = L.MiConvParams()
q.B, q.H, q.W = B, H, W
q.in0 = L.MiAct(mid.data_ptr(), 8, mst.data_ptr(), nt, 1.0, 0)
q.Cout, q.ksize, q.stride, q.up2 = 3, 3, 1, 0
q.w_rp, q.w_rp_exp, q.bias, q.out, q.tile_cfg = keep["w2f"].data_ptr(), w2exp, keep["b2"].data_ptr(), sep.data_ptr(), 5
st = L.current_stream()


def separate():
    set_res(False)
    p.out, p.out_stats, p.tile_cfg = mid.data_ptr(), mst.data_ptr(), 5
    L.check(lib.mi_conv_fwd(C.byref(p), st), "block2")
    L.check(lib.mi_conv_fwd(C.byref(q), st), "final conv")


def fused():
    set_res(True)
    p.tile_cfg = 5 | (strip << 12)
    L.check(lib.mi_conv_tail_fwd(C.byref(tp), st), "fused tail")


separate(); fused()
torch.cuda.synchronize()
print(f"max |fused - separate| = {(out2 - sep).abs().max().item():.2e} (|out| max {sep.abs().max().item():.2f})")
for name, fn in (("separate (2 launches)", separate), ("fused (1 launch)", fused)):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"B{B} {H}x{W} strip {strip} {name}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us")
if hasattr(lib, "mi_debug_read_trace_rp"):        # a -DMI_TRACE build: per-phase shader clocks of the first workgroups of the last (fused) launch
    import numpy as np
    NS = lib.mi_debug_trace_rp_slots() if hasattr(lib, "mi_debug_trace_rp_slots") else 8
    buf = np.zeros(1024 * NS, dtype=np.uint64)
    lib.mi_debug_read_trace_rp.argtypes = [C.c_void_p, C.c_size_t]
    lib.mi_debug_read_trace_rp(buf.ctypes.data, buf.nbytes)
    t = buf.reshape(1024, NS).astype(np.int64)
    names = ["stats+geometry+issue loads", "affine prologue", "barrier waits", "wait raw + transform + LDS write", "MFMA loop", "epilogue / mid + second conv", "B-frag issue"]
    for i, n in enumerate(names):
        print(f"      {n:34s} {np.median(t[:, i]):9.0f} {np.percentile(t[:, i], 10):9.0f} {np.percentile(t[:, i], 90):9.0f}")
    print(f"      total                              {np.median(t[:, :7].sum(1)):9.0f}")
    w = buf.reshape(1024, NS)[:, 7]
    w0 = ((w >> np.uint64(32)) & np.uint64(0xffffffff)).astype(np.int64); w1 = (w & np.uint64(0xffffffff)).astype(np.int64)
    ok = w1 > 0
    if ok.any():
        life = (w1[ok] - w0[ok]) / 100.0
        print(f"      workgroup life: median {np.median(life):.1f} us, 10/90 % {np.percentile(life, 10):.1f} / {np.percentile(life, 90):.1f} us")
