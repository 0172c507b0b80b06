"""GPU busy fraction of a rocprofv3 kernel trace: union of the kernel intervals over the traced window, how many kernels run side by
side, and the time with exactly one / two or more kernels resident.  usage: python tools/trace_busy.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5          # drop the first part of the trace (model upload, warm-up, graph capture)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
lo = t0 + (t1 - t0) * skip
iv = [(s, e, n) for s, e, n in iv if s >= lo]
ev = sorted([(s, 1) for s, e, n in iv] + [(e, -1) for s, e, n in iv])
depth, last, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth, last = depth + d, t
tot = sum(hist.values())
print(f"window {tot / 1e6:.2f} ms, {len(iv)} kernels; sum of kernel durations {sum(e - s for s, e, n in iv) / 1e6:.2f} ms")
for k in sorted(hist):
    print(f"  {k} kernels resident: {hist[k] / 1e6:8.2f} ms  ({100.0 * hist[k] / tot:5.1f} %)")
gaps = sorted((b[0] - a for a, b in zip([e for _, e, _ in iv], iv[1:])), reverse=True)
by = {}
for s, e, n in iv:
    k = n.split("<")[0].split("(")[0][-40:]
    by[k] = by.get(k, 0) + (e - s)
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]:
    print(f"  {v / 1e6:8.2f} ms  {k}")
