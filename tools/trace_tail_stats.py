"""Aggregate the LAST `seconds` of a rocprofv3 kernel trace by kernel name (steady state of a run whose start is polluted by library
auto-tuning).  usage: trace_tail_stats.py <kernel_trace.csv> <seconds> [top]"""
import csv, sys
from collections import defaultdict
path, secs = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = list(csv.DictReader(open(path)))
end = max(int(r["End_Timestamp"]) for r in rows)
t0 = end - int(secs * 1e9)
agg = defaultdict(lambda: [0, 0])
tot = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= t0:
        a = agg[r["Kernel_Name"]]
        a[0] += 1; a[1] += e - s; tot += e - s
print(f"last {secs} s: {sum(a[0] for a in agg.values())} launches, {tot / 1e6:.1f} ms of kernel time")
for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{ns / 1e6:9.2f} ms {100 * ns / tot:5.1f}% {n:6d} x {ns / n / 1e3:9.1f} us  {name[:110]}")
