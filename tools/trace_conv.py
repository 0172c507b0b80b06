"""Development aid: per-phase shader-clock timeline of the matrix-core conv (needs the -DMI_TRACE build of the library,
see tools/gpu_trace_conv.sh).  Launches the 16->16 GroupNorm conv of the SR U-Net's 64x64 level (B=64) and prints, per
workgroup sample, the cycles spent between the MI_STAMP points of conv_mfma.hip."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minimagen_amd import _lib as L
from minimagen_amd import packing as P

L.use_library(os.environ["MINIMAGEN_HIP_LIB"])
lib = L.lib()
dev = torch.device("cuda:0")
B, Cc, H, W = 64, 16, 64, 64
flags = int(os.environ.get("TRACE_FLAGS", "0"), 0)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, Cc, H, W, generator=g).to(dev)
nt_in = 8
stats = torch.zeros(B, Cc, nt_in, 2, device=dev)
stats[:, :, 0, 0] = x.sum((2, 3)); stats[:, :, 0, 1] = (x * x).sum((2, 3))
w = torch.randn(Cc, Cc, 3, 3, generator=g) * 0.2
wf = P.pack_conv_weight_f16frag(w).to(dev)
bias, gamma, beta = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
p = L.MiConvParams()
p.B, p.H, p.W = B, H, W
p.in0 = L.MiAct(x.data_ptr(), Cc, stats.data_ptr(), nt_in, 1.0, 0)
p.Cout, p.ksize, p.stride, p.up2 = Cc, 3, 1, 0
p.w_f16, p.bias = wf.data_ptr(), bias.data_ptr()
p.gn_groups, p.gn_gamma, p.gn_beta, p.gn_eps = 8, gamma.data_ptr(), beta.data_ptr(), 1e-5
res = torch.randn(B, Cc, H, W, generator=g).to(dev)
if os.environ.get("TRACE_RES", "0") == "1":
    p.res0 = L.MiAct(res.data_ptr(), Cc, 0, 0, 1.0, 0)
out = torch.empty(B, Cc, H, W, device=dev)
ost = torch.zeros(B, Cc, 8, 2, device=dev)
p.out, p.out_stats, p.tile_cfg = out.data_ptr(), ost.data_ptr(), 3 | flags
st = L.current_stream()
for _ in range(5):
    L.check(lib.mi_conv_fwd(C.byref(p), st), "conv")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    L.check(lib.mi_conv_fwd(C.byref(p), st), "conv")
e1.record(); torch.cuda.synchronize()
print(f"flags={flags:#x} avg launch {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
buf = np.zeros(1024 * 8, dtype=np.uint64)
lib.mi_debug_read_trace.argtypes = [C.c_void_p, C.c_size_t]
rc = lib.mi_debug_read_trace(buf.ctypes.data, buf.nbytes)
t = buf.reshape(1024, 8).astype(np.int64)
t = t[:B * 8]
d = np.diff(t[:, :7], axis=1)
names = ["geometry + raw loads issued", "GN-stats prologue", "acc init + barrier", "weights + transform + LDS write", "MFMA loop", "epilogue"]
print("phase cycles (median / p10 / p90 over the first 512 workgroups):")
for i, n in enumerate(names):
    print(f"  {n:34s} {np.median(d[:, i]):9.0f} {np.percentile(d[:, i], 10):9.0f} {np.percentile(d[:, i], 90):9.0f}")
tot = t[:, 6] - t[:, 0]
print(f"  workgroup total                    {np.median(tot):9.0f} {np.percentile(tot, 10):9.0f} {np.percentile(tot, 90):9.0f}")
print(f"  first start -> last end            {t[:, 6].max() - t[:, 0].min():9.0f} cycles;  start spread {t[:, 0].max() - t[:, 0].min()}")

# ---- the VALU conv (conv.hip): 8->8 GroupNorm conv of the 256x256 level, B=64 -> accumulated cycles per phase
if hasattr(lib, "mi_debug_read_trace_conv"):
    B, Cc, H, W = 64, 8, 256, 256
    x = torch.randn(B, Cc, H, W, generator=g).to(dev)
    nt_in = 64
    stats = torch.zeros(B, Cc, nt_in, 2, device=dev)
    stats[:, :, 0, 0] = x.sum((2, 3)); stats[:, :, 0, 1] = (x * x).sum((2, 3))
    w = torch.randn(Cc, Cc, 3, 3, generator=g) * 0.2
    wp = P.pack_conv_weight(w, 8).to(dev)
    bias, gamma, beta = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    p = L.MiConvParams()
    p.B, p.H, p.W = B, H, W
    p.in0 = L.MiAct(x.data_ptr(), Cc, stats.data_ptr(), nt_in, 1.0, 0)
    p.Cout, p.ksize, p.stride, p.up2 = Cc, 3, 1, 0
    p.w, p.bias = wp.data_ptr(), bias.data_ptr()
    p.gn_groups, p.gn_gamma, p.gn_beta, p.gn_eps = 8, gamma.data_ptr(), beta.data_ptr(), 1e-5
    out = torch.empty(B, Cc, H, W, device=dev)
    ost = torch.zeros(B, Cc, 64, 2, device=dev)
    p.out, p.out_stats, p.tile_cfg = out.data_ptr(), ost.data_ptr(), 0 | 0x100
    for _ in range(3):
        L.check(lib.mi_conv_fwd(C.byref(p), st), "conv")
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        L.check(lib.mi_conv_fwd(C.byref(p), st), "conv")
    e1.record(); torch.cuda.synchronize()
    print(f"VALU conv 8->8 @256 B64: avg launch {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
    lib.mi_debug_read_trace_conv.argtypes = [C.c_void_p, C.c_size_t]
    lib.mi_debug_read_trace_conv(buf.ctypes.data, buf.nbytes)
    t = buf.reshape(1024, 8).astype(np.int64)
    names = ["stats totals + geometry", "issue loads", "rest of prologue", "barrier waits", "activation + LDS write", "FMA loops", "epilogue", "-"]
    for i, n in enumerate(names[:7]):
        print(f"  {n:34s} {np.median(t[:, i]):9.0f} {np.percentile(t[:, i], 10):9.0f} {np.percentile(t[:, i], 90):9.0f}")
    print(f"  total                              {np.median(t[:, :7].sum(1)):9.0f}")

# ---- the stride-2 Downsample conv (k4 s2 p1): 8->8, input 256x256 -> output 128x128, B=32 (shared between the guidance halves)
if hasattr(lib, "mi_debug_read_trace_conv"):
    B, Cc, H, W = 32, 8, 128, 128
    x = torch.randn(B, Cc, 2 * H, 2 * W, generator=g).to(dev)
    w = torch.randn(Cc, Cc, 4, 4, generator=g) * 0.2
    wp = P.pack_conv_weight(w, 8).to(dev)
    p = L.MiConvParams()
    p.B, p.H, p.W = B, H, W
    p.in0 = L.MiAct(x.data_ptr(), Cc, 0, 0, 1.0, 0)
    p.Cout, p.ksize, p.stride, p.up2 = Cc, 4, 2, 0
    p.w, p.bias = wp.data_ptr(), bias.data_ptr()
    out = torch.empty(B, Cc, H, W, device=dev)
    ost = torch.zeros(B, Cc, 16, 2, device=dev)
    p.out, p.out_stats, p.tile_cfg = out.data_ptr(), ost.data_ptr(), 0 | 0x100
    for _ in range(3):
        L.check(lib.mi_conv_fwd(C.byref(p), st), "conv")
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        L.check(lib.mi_conv_fwd(C.byref(p), st), "conv")
    e1.record(); torch.cuda.synchronize()
    print(f"VALU conv k4s2 8->8 @128 B32: avg launch {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
    lib.mi_debug_read_trace_conv(buf.ctypes.data, buf.nbytes)
    t = buf.reshape(1024, 8).astype(np.int64)[:512]
    for i, n in enumerate(names[:7]):
        print(f"  {n:34s} {np.median(t[:, i]):9.0f} {np.percentile(t[:, i], 10):9.0f} {np.percentile(t[:, i], 90):9.0f}")
    print(f"  total                              {np.median(t[:, :7].sum(1)):9.0f}")
