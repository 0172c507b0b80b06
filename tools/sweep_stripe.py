"""Development aid (round 6): isolated launch times of the stripe conv kernel (tile_cfg 12) over library builds (loader-group depths are
compile-time: csrc/conv_stripe.hip ST_LD_*) x stripe lengths (statistics blocks per workgroup) next to the tile kernel, one process.
  python tools/sweep_stripe.py [lib suffixes, default: '' vA vB vC]"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minimagen_amd import _lib as L
import tools.bench_conv as BC

SHAPES = [  # B, Cin, Cout, H, W, gn, res, C1
    (64, 8, 8, 256, 256, True, "id", 0), (64, 8, 8, 256, 256, True, "none", 0), (64, 8, 3, 256, 256, False, "none", 0),
    (64, 8, 8, 128, 128, True, "id", 0), (64, 8, 8, 128, 128, True, "none", 0), (32, 8, 8, 128, 128, True, "none", 0), (64, 8, 8, 128, 128, True, "none", 8),
    (64, 16, 16, 64, 64, True, "id", 0), (64, 16, 16, 64, 64, True, "none", 0), (64, 16, 16, 64, 64, True, "none", 16),
    (64, 8, 8, 64, 64, True, "id", 0), (32, 8, 8, 64, 64, True, "none", 0), (64, 8, 8, 64, 64, True, "none", 8), (64, 8, 3, 64, 64, False, "none", 0),
    (64, 16, 16, 32, 32, True, "id", 0), (64, 8, 8, 32, 32, True, "id", 0), (64, 16, 16, 32, 32, True, "none", 8), (64, 8, 16, 32, 32, False, "none", 0),
]
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timed(*a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return BC.run(*a, **kw)


def main():
    sufs = [a if a != "base" else "" for a in sys.argv[1:]] or [""]
    shapes = [sh for sh in SHAPES if sh[4] >= int(os.environ.get("SWEEP_MIN_W", "0"))]
    nblks = [int(v) for v in os.environ.get("SWEEP_NBLK", "1,2,4").split(",")]
    rows = {}
    for suf in sufs:
        BC.lib = L.use_library(os.path.join(here, "minimagen_amd", f"libminimagen_hip{'_' + suf if suf else ''}.so"))
        for sh in shapes:
            B, C0, Cout, H, W, gn, res, C1 = sh
            if suf == sufs[0]:
                os.environ["NTILE"] = "0"
                rows.setdefault(sh, {})["tile"] = timed(B, C0, Cout, H, W, gn, res, "rp6" if W >= 64 else "rp7", C1)
            nt = H // (8 if W >= 64 else 4)
            for nblk in nblks:
                if nt % nblk:
                    continue
                os.environ["NTILE"] = str(nblk)
                rows[sh][f"{suf or 'base'}/{nblk}"] = timed(B, C0, Cout, H, W, gn, res, "rp12", C1)
    os.environ["NTILE"] = "0"
    for sh, r in rows.items():
        B, C0, Cout, H, W, gn, res, C1 = sh
        best = min((v, k) for k, v in r.items() if k != "tile")
        print(f"B{B} {C0 + C1}->{Cout} @{W} gn={int(gn)} res={res}: tile {r['tile']:6.1f} | best {best[1]} {best[0]:6.1f} | " + "  ".join(f"{k} {v:5.1f}" for k, v in r.items() if k != "tile"))


if __name__ == "__main__":
    main()
