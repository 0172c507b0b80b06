"""Times the resident conv chain (mi_resident_convs_fwd) on the three segments of the super-resolution U-Net's 64^2 level at the bench
batch (64 rows) next to the same layers as separate mi_conv_fwd launches of the row-paired kernel.  GPU only."""
import ctypes as C
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minimagen_amd import _lib as L          # noqa: E402
from tests.test_resident import Chain, sr_level_chain      # noqa: E402


def time_chain(ch, reps=50):
    lib = L.lib()
    st = L.current_stream()
    for _ in range(5):
        L.check(lib.mi_resident_convs_fwd(C.byref(ch.p), st), "resident")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.mi_resident_convs_fwd(C.byref(ch.p), st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    L.use_library(L.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    res = {}
    for variant in ("down", "mid", "up"):
        ch = Chain(dev, B, 64, 64, seed=5)
        sr_level_chain(ch, B, 64, 64, variant)
        ch.run()
        us = time_chain(ch)
        err = int(ch.sync[8:12].cpu().view(torch.int32).item())
        res[variant] = dict(layers=ch.n, us=round(us, 2), us_per_layer=round(us / ch.n, 2), err=err)
        print(variant, res[variant], flush=True)
        if hasattr(L.lib(), "mi_debug_read_trace_rs"):
            import numpy as np
            buf = np.zeros(256 * 8 * 16, dtype=np.uint64)
            L.lib().mi_debug_read_trace_rs.argtypes = [C.c_void_p, C.c_size_t]
            L.lib().mi_debug_read_trace_rs(buf.ctypes.data, buf.nbytes)
            t = buf.reshape(256, 8, 16).astype(np.int64)[:min(256, B * ch.S), :ch.n, :] * 0.01
            t0 = t[:, 0, 0].min()
            print(f"  kernel span (first stamp of any workgroup -> last stamp): {t[:, ch.n - 1, 7].max() - t0:.2f} us; first-stamp skew {t[:, 0, 0].max() - t0:.2f} us")
            seg = [("early loads", 0, 1), ("poll partials", 1, 2), ("affine", 2, 3), ("stage own rows", 3, 4), ("halo + barrier", 4, 5), ("mma (+ later passes)", 5, 6),
                   ("y + residual", 6, 8), ("shuffles + barrier", 8, 9), ("w commit + stats publish", 9, 10), ("halo publish + out stores", 10, 7)]
            if hasattr(L.lib(), "mi_debug_read_trace_rs_w") and variant == "down":
                bw = np.zeros(64 * 8 * 16 * 4, dtype=np.uint64)
                L.lib().mi_debug_read_trace_rs_w.argtypes = [C.c_void_p, C.c_size_t]
                L.lib().mi_debug_read_trace_rs_w(bw.ctypes.data, bw.nbytes)
                w = bw.reshape(64, 8, 16, 4).astype(np.int64) * 0.01
                for wg in (0, 1, 5):
                    base = w[wg, 2, 0, 0]
                    print(f"  workgroup {wg}, layer 2, per wave (mma done, y + residual done, before the reduction barrier, after it) relative to wave 0's mma-done:")
                    for wv in range(8):
                        print("     wave", wv, " ".join(f"{w[wg, 2, wv, k] - base:6.2f}" for k in range(4)))
            for li in range(ch.n):
                print(f"  layer {li}: start {t[:, li, 0].mean() - t0:7.2f}; " + "; ".join(f"{n} {(t[:, li, b_] - t[:, li, a_]).mean():.2f}" for n, a_, b_ in seg) + f"; total {(t[:, li, 7] - t[:, li, 0]).mean():.2f}")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
