"""Development aid: time ONE conv launch shape of the SR / base U-Net in isolation on the GPU (HIP events, back to back) and, with the
-DMI_TRACE build of the library (MINIMAGEN_HIP_LIB=.../libminimagen_hip_trace.so), print the per-phase shader-clock breakdown of
conv_rp.hip.   python tools/bench_conv.py B Cin Cout H W gn res(none|id|conv) path(rp6|rp7|rp12: full-width stripes|old|w6|w7|w10: the wide regime) [C1]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minimagen_amd import _lib as L
from minimagen_amd import packing as P

lib = L.lib()
dev = torch.device("cuda:0")


def run(B, C0, Cout, H, W, gn, res, path, C1=0, reps=20, nt_in=None):
    g = torch.Generator().manual_seed(1)
    Cin = C0 + C1
    x0 = torch.randn(B, C0, H, W, generator=g).to(dev)
    x1 = torch.randn(B, C1, H, W, generator=g).to(dev) if C1 else None
    nt_in = nt_in or max(1, (H // 16) * (W // 64))

    def stats(x):
        st = torch.zeros(x.shape[0], x.shape[1], nt_in, 2, dtype=torch.float64, device=dev)
        st[:, :, 0, 0] = x.sum((2, 3)); st[:, :, 0, 1] = (x * x).sum((2, 3))
        return st
    s0, s1 = stats(x0), (stats(x1) if C1 else None)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2
    bias, gamma, beta = torch.zeros(Cout, device=dev), torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    p = L.MiConvParams()
    p.B, p.H, p.W = B, H, W
    p.in0 = L.MiAct(x0.data_ptr(), C0, s0.data_ptr(), nt_in, 1.0, 0)
    if C1:
        p.in1 = L.MiAct(x1.data_ptr(), C1, s1.data_ptr(), nt_in, 0.7071, 0)
    p.Cout, p.ksize, p.stride, p.up2 = Cout, 3, 1, 0
    keep = []
    wide = path.startswith("w")
    if wide:
        path = "rp" + path[1:]
    gemm = path == "rp11"                      # the wide GEMM kernel (conv_wide.hip)
    pack_w = P.pack_conv_weight_ig if gemm else P.pack_conv_weight_rp
    if path.startswith("rp"):
        wf, p.w_rp_exp = pack_w(w)
        wf = wf.to(dev); keep.append(wf)
        p.w_rp = wf.data_ptr()
        cfg = int(path[2:]) | (int(os.environ.get("NTILE", "0")) << 12)
    else:
        ct = lib.mi_conv_cout_tile(Cout)
        wp = P.pack_conv_weight(w, ct).to(dev); keep.append(wp)
        p.w = wp.data_ptr()
        cfg = (0 if (W >= 64 and H * W > 64 * 64) else 2) | 0x100 | (0x800 if H * W <= 64 * 64 else 0)
    p.bias = bias.data_ptr()
    if gn:
        p.gn_groups, p.gn_gamma, p.gn_beta, p.gn_eps = 8, gamma.data_ptr(), beta.data_ptr(), 1e-5
        if not all(s_ is not None for s_ in (s0,)):
            raise SystemExit("GroupNorm needs statistics")
    if res == "id":
        r = torch.randn(B, Cout, H, W, generator=g).to(dev); keep.append(r)
        p.res0 = L.MiAct(r.data_ptr(), Cout, 0, 0, 1.0, 0)
    elif res == "conv":
        r = torch.randn(B, Cin, H, W, generator=g).to(dev); keep.append(r)
        rs = stats(r); keep.append(rs)
        p.res0 = L.MiAct(r.data_ptr(), Cin, rs.data_ptr(), nt_in, 1.0, 0)
        rw = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.3
        p.res_w = 1
        if path.startswith("rp"):
            rwf, p.res_w_rp_exp = pack_w(rw); rwf = rwf.to(dev); keep.append(rwf); p.res_w_rp = rwf.data_ptr()
        else:
            rwp = P.pack_conv_weight(rw, lib.mi_conv_cout_tile(Cout)).reshape(Cin, -1).contiguous().to(dev); keep.append(rwp); p.res_w = rwp.data_ptr()
    if (cfg & 0xff) == 12:                         # full-width stripes (conv_stripe.hip): statistics blocks of mi_conv_stripe_rows rows
        rows = lib.mi_conv_stripe_rows(C.byref(p))
        if not rows:
            raise SystemExit("the stripe kernel does not take this shape")
        nt = H // rows
    else:
        th, tw = C.c_int(), C.c_int()
        lib.mi_conv_tile_shape(cfg & 0xff, C.byref(th), C.byref(tw))
        nt = -(-H // th.value) * -(-W // tw.value)
    out = torch.empty(B, Cout, H, W, device=dev)
    ost = torch.zeros(B, Cout, nt, 2, dtype=torch.float64, device=dev)
    p.out, p.out_stats, p.tile_cfg = out.data_ptr(), ost.data_ptr(), cfg
    st = L.current_stream()
    if wide:
        coef = torch.zeros(B, Cin, 4, device=dev); exps = torch.zeros(B, 2, dtype=torch.int32, device=dev); keep += [coef, exps]
        p.gn_coef, p.gn_exps = coef.data_ptr(), exps.data_ptr()
        L.check(lib.mi_gn_coef_fwd(C.byref(p), st), "gn_coef")
        if gemm:
            nbytes = lib.mi_conv_prep_bytes(B, Cin, Cin if res == "conv" else 0, H, W)
            prep = torch.empty(nbytes // 4, device=dev); keep.append(prep)
            p.act_prep, p.act_prep_bytes = prep.data_ptr(), nbytes
            L.check(lib.mi_conv_prep_fwd(C.byref(p), st), "conv_prep")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                L.check(lib.mi_conv_prep_fwd(C.byref(p), st), "conv_prep")
            e1.record(); torch.cuda.synchronize()
            print(f"      conv_prep: {e0.elapsed_time(e1) / reps * 1e3:7.1f} us ({nbytes / 1e6:.0f} MB written)")
    for _ in range(3):
        L.check(lib.mi_conv_fwd(C.byref(p), st), "conv")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.mi_conv_fwd(C.byref(p), st), "conv")
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    mb = (B * (Cin + Cout + (Cout if res == "id" else (Cin if res == "conv" else 0))) * H * W * 4) / 1e6
    tf = 2.0 * (Cin * 9 + (Cin if res == "conv" else 0)) * Cout * H * W * B / us * 1e-6
    print(f"{path:5s}{' wide' if wide else ''} B{B} {Cin}->{Cout} @{H}x{W} gn={int(gn)} res={res}: {us:7.1f} us  ({mb:.0f} MB -> {mb / us * 1e-3 * 1e3:.2f} GB/ms; {tf:.0f} TFLOP/s algorithmic, x3 split terms issued)")
    if path.startswith("rp") and hasattr(lib, "mi_debug_read_trace_rp"):
        NS = lib.mi_debug_trace_rp_slots() if hasattr(lib, "mi_debug_trace_rp_slots") else 8
        buf = np.zeros(1024 * NS, dtype=np.uint64)
        lib.mi_debug_read_trace_rp.argtypes = [C.c_void_p, C.c_size_t]
        lib.mi_debug_read_trace_rp(buf.ctypes.data, buf.nbytes)
        t = buf.reshape(1024, NS).astype(np.int64)
        t = t[t[:, :7].sum(1) > 0]                 # (launches with fewer than 1024 workgroups along x fill only the first rows)
        names = ["stats+geometry+issue loads", "affine prologue", "barrier waits", "wait raw + transform + LDS write", "MFMA loop", "epilogue", "B-frag issue"]
        for i, n in enumerate(names):
            print(f"      {n:34s} {np.median(t[:, i]):9.0f} {np.percentile(t[:, i], 10):9.0f} {np.percentile(t[:, i], 90):9.0f}")
        print(f"      total                              {np.median(t[:, :7].sum(1)):9.0f}")
        if NS > 8 and t[:, 8:14].any():
            for i, n in zip(range(8, 14), ["  fine: residual + prefetch loads issued", "  fine: MFMA loop proper", "  fine: epilogue arithmetic (+ residual wait)", "  fine: stores issued",
                                           "  fine: statistics shuffles", "  fine: barrier + statistics store"]):
                print(f"      {n:46s} {np.median(t[:, i]):9.0f} {np.percentile(t[:, i], 10):9.0f} {np.percentile(t[:, i], 90):9.0f}")
        w = buf.reshape(1024, NS)[:len(t), 7]
        w0 = ((w >> np.uint64(32)) & np.uint64(0xffffffff)).astype(np.int64); w1 = (w & np.uint64(0xffffffff)).astype(np.int64)
        ok = w1 > 0
        if ok.any():
            w0, w1 = w0[ok], w1[ok]; base = w0.min()
            life = (w1 - w0) / 100.0
            idx = np.nonzero(ok)[0]
            for x in range(8):
                m = (idx % 8) == x
                print(f"        xcd {x}: end med {np.median(w1[m] - base) / 100.0:6.1f} min {np.min(w1[m] - base) / 100.0:6.1f} max {np.max(w1[m] - base) / 100.0:6.1f} us")
            q = (idx // 8) % 4
            print("        by (blockIdx/8)%4:", [float(np.median(w1[q == k] - base)) / 100.0 for k in range(4)])
            print("        by blockIdx range quartile:", [float(np.median(w1[(idx * 4 // 1024) == k] - base)) / 100.0 for k in range(4)])
            print(f"      wall: starts {np.percentile(w0 - base, [0, 50, 90, 100]) / 100.0} us, ends {np.percentile(w1 - base, [0, 50, 90, 100]) / 100.0} us, life med {np.median(life):.1f} us"
                  f" -> clock {np.median(t[ok][:, :7].sum(1) / np.maximum(life, 1e-3)) / 1e3:.2f} GHz")
    return us


if __name__ == "__main__":
    a = sys.argv[1:]
    if a:
        run(int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), a[5] == "1", a[6], a[7], int(a[8]) if len(a) > 8 else 0)
    else:
        for nt in (1, 2, 4, 8):
            os.environ["NTILE"] = str(nt)
            print("NTILE", nt)
            for path in ("rp12", "rp6"):
                run(64, 8, 8, 256, 256, True, "id", path)
            run(64, 8, 3, 256, 256, False, "none", "rp6")
            for path in ("rp6", "rp7"):
                run(64, 8, 8, 128, 128, True, "id", path)
            for path in ("rp6", "rp7"):
                run(64, 16, 16, 64, 64, True, "id", path)
