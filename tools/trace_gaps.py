"""Per-call overhead from a rocprofv3 kernel trace of SYNCHRONOUS sample() calls: the GPU time between consecutive kernels (idle gaps),
split into the gaps inside the denoising loops (graph replay boundaries) and everything else, plus the kernels that are not part of a
denoising step (text conditioning, folds, step tables, resize, finalize, fingerprints).  usage: python tools/trace_gaps.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::|void |<.*|\(.*", "", r["Kernel_Name"])))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, gaps, end = 0, [], rows[0][0]
for s, e, n in rows:
    if s > end:
        gaps.append((s - end, n))
    busy += max(0, e - max(s, end))
    end = max(end, e)
print(f"window {(t1 - t0) / 1e6:.2f} ms, kernels {len(rows)}, busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms")
hist = collections.Counter()
for g, n in gaps:
    hist["<5us" if g < 5e3 else "<20us" if g < 2e4 else "<100us" if g < 1e5 else "<1ms" if g < 1e6 else ">=1ms"] += g
print("idle time by gap size (ms):", {k: round(v / 1e6, 3) for k, v in hist.items()})
big = sorted(gaps, reverse=True)[:15]
print("largest gaps (us, kernel that follows):", [(round(g / 1e3, 1), n) for g, n in big])
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in rows:
    tot[n] += e - s; cnt[n] += 1
print("kernel totals (ms, launches):")
for n, v in tot.most_common(40):
    print(f"  {v / 1e6:9.3f} {cnt[n]:7d}  {n}")
