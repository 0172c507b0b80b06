"""Development aid (round 6): per-phase shader-clock stamps of the stripe conv kernel (csrc/conv_stripe.hip built with -DST_TRACE ->
minimagen_amd/libminimagen_hip_tr.so): medians over the workgroups of one isolated launch, relative to the workgroup's own start."""
import contextlib
import ctypes as C
import io
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minimagen_amd import _lib as L
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L.use_library(os.path.join(here, "minimagen_amd", "libminimagen_hip_tr.so"))
import tools.bench_conv as BC

NAMES = ["C start", "C totals done", "C barrier 1", "C moments + affine", "C B frags staged", "C barrier 3", "C barrier 4 (first 4 rows in the ring)", "C first step multiplied", "C loop done",
         "C published", "L start", "L first rows requested", "L barrier 3", None, "L barrier 4", "L done",
         "C step 2: start", "C step 2: accumulators done", "C step 2: epilogue issued", "C step 2: past the barrier", "L slot 2: start", "L slot 2: transform written", "L slot 2: loads issued", "L slot 2: past the barrier"]
SHAPES = [(64, 8, 8, 256, 256, True, "none", 0, 1), (64, 8, 8, 256, 256, True, "id", 0, 1), (64, 8, 8, 128, 128, True, "id", 0, 1), (64, 16, 16, 64, 64, True, "none", 0, 1),
          (64, 16, 16, 64, 64, True, "none", 0, 2), (64, 8, 8, 64, 64, True, "id", 0, 1), (64, 8, 8, 32, 32, True, "id", 0, 1)]
lib = L.lib()
lib.mi_debug_read_trace_st.argtypes = [C.c_void_p, C.c_size_t]
for B, C0, Cout, H, W, gn, res, C1, nblk in SHAPES:
    os.environ["NTILE"] = str(nblk)
    with contextlib.redirect_stdout(io.StringIO()):
        us = BC.run(B, C0, Cout, H, W, gn, res, "rp12", C1)
    buf = np.zeros(1024 * 32, dtype=np.uint64)
    lib.mi_debug_read_trace_st(buf.ctypes.data, buf.nbytes)
    t = buf.reshape(1024, 32).astype(np.int64)
    nwg = min(1024, B * 8 // nblk)
    t = t[:nwg]
    t0 = np.minimum(t[:, 0], t[:, 10])
    rel = (t - t0[:, None]) / 100.0            # s_memtime ticks / 100 (calibrated against launch times: one unit ~ 0.115 us)
    print(f"B{B} {C0 + C1}->{Cout} @{W} gn={int(gn)} res={res} nblk={nblk}: {us:.1f} us per launch, {nwg} workgroups traced; workgroup starts spread {np.ptp(t0) / 100.0:.1f} us")
    for k, n in enumerate(NAMES):
        if n is None:
            continue
        print(f"   {n:28s} median {np.median(rel[:, k]):7.2f}  p10 {np.percentile(rel[:, k], 10):7.2f}  p90 {np.percentile(rel[:, k], 90):7.2f} us")
