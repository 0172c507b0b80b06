"""Per-kernel register / spill / LDS table of one HIP source, from the compiler's resource remarks.
usage: python tools/kernel_resources.py minimagen_amd/csrc/conv_rp.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Iinclude", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.split("\n"):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for k, pat in (("vgpr", r" VGPRs: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                   ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[k] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = re.sub(r"\(mi_conv_params.*$", "", re.sub(r"^void ", "", n)).replace("(int)", "").replace("(bool)", "")
    print("%4d vgpr %4d vspill %3d sspill %6d lds occ %d  %s" % (r.get("vgpr", -1), r.get("vspill", 0), r.get("sspill", 0), r.get("lds", 0), r.get("occ", 0), n))
