"""CrossEmbed micro-benchmark (dev tool): python tools/bench_ce.py [B H W cfg mfma half]  -- SR shape by default"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minimagen_amd import _lib as L
from minimagen_amd import packing as P

a = sys.argv[1:]
B, H, W = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (32, 256, 256)
cfg = int(a[3]) if len(a) > 3 else 8
mfma = (a[4] != "0") if len(a) > 4 else True
half = (a[5] != "0") if len(a) > 5 else False
dev = torch.device("cuda:0")
L.use_library(os.environ.get("MINIMAGEN_HIP_LIB", L.DEFAULT_LIB))
lib = L.lib()
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 3, H, W, generator=g).to(dev)
ws = [torch.randn(co, 3, k, k, generator=g) * 0.1 for k, co in zip((3, 7, 15), (4, 2, 2))]
bs = [torch.randn(co, generator=g).to(dev) for co in (4, 2, 2)]
add = torch.randn(B, 8, H, W, generator=g).to(dev)
p = L.MiCrossEmbedParams()
p.B, p.H, p.W, p.in0, p.C0, p.n_kernels = B, H, W, x.data_ptr(), 3, 3
wp = [w.permute(1, 2, 3, 0).contiguous().to(dev) for w in ws]
tab, exps = P.pack_crossembed_mfma(ws, 0, 3)
tab = tab.to(dev)
for i, (k, co) in enumerate(zip((3, 7, 15), (4, 2, 2))):
    p.ksize[i], p.cout[i], p.w[i], p.bias[i], p.w_mfma_exp[i] = k, co, wp[i].data_ptr(), bs[i].data_ptr(), exps[i]
if mfma:
    p.w_mfma = tab.data_ptr()
th, tw = C.c_int(), C.c_int()
lib.mi_conv_tile_shape(cfg, C.byref(th), C.byref(tw))
nt = -(-H // th.value) * -(-W // tw.value)
out = torch.empty(B, 8, H, W, device=dev)
ost = torch.zeros(B, 8, nt, 2, dtype=torch.float64, device=dev)
p.out, p.out_stats, p.tile_cfg, p.addend = out.data_ptr(), ost.data_ptr(), cfg | (0x400 if half else 0), add.data_ptr()
st = L.current_stream()
for _ in range(3):
    L.check(lib.mi_crossembed_fwd(C.byref(p), st), "crossembed")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    L.check(lib.mi_crossembed_fwd(C.byref(p), st), "crossembed")
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"crossembed B{B} {H}x{W} cfg {cfg} mfma={int(mfma)} half={int(half)}: {us:.1f} us")
if hasattr(lib, "mi_debug_read_trace_ce"):
    import numpy as np
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    lib.mi_debug_read_trace_ce.argtypes = [C.c_void_p, C.c_size_t]
    lib.mi_debug_read_trace_ce(buf.ctypes.data, buf.nbytes)
    t = buf.reshape(1024, 8)
    for i, n in enumerate(["issue loads", "wait + max + split + LDS write", "GEMM", "epilogue"]):
        v = t[:, i].astype(np.int64)
        print(f"      {n:32s} {np.median(v):9.0f} {np.percentile(v, 10):9.0f} {np.percentile(v, 90):9.0f}")
    w = t[:, 7]
    w0 = ((w >> np.uint64(32)) & np.uint64(0xffffffff)).astype(np.int64); w1 = (w & np.uint64(0xffffffff)).astype(np.int64)
    ok = w1 > 0
    base = w0[ok].min()
    print("      wall: starts", np.percentile(w0[ok] - base, [0, 50, 90, 100]) / 100.0, "us, ends", np.percentile(w1[ok] - base, [0, 50, 90, 100]) / 100.0,
          "us; life med %.1f us" % (np.median(w1[ok] - w0[ok]) / 100.0), "-> clock %.2f GHz" % (np.median(t[ok][:, :4].astype(np.int64).sum(1) / np.maximum((w1[ok] - w0[ok]) / 100.0, 1e-3)) / 1e3))
