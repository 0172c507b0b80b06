#!/bin/bash
# per-launch kernel trace of sampling with the default Unet(): time per (kernel, grid) -> gpurun_out/wide_by_grid.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/wide_tr
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/wide_tr -o wide -- python $R/tools/gpu_wide_sample.py ${B:-16} 25 > $R/gpurun_out/wide_trace.log 2>&1
python - <<PY > $R/gpurun_out/wide_by_grid.txt
import csv, glob, re, collections
f = glob.glob("/tmp/wide_tr/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]); n = re.sub(r"\(mi_.*|\(float.*", "", n)
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]); w = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    d[(n, g, w)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in d.values())
print("kernel | grid threads | wg | launches | avg us | share")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(f"{k[0][:64]:64s} {k[1]:9d} {k[2]:5d} {len(v):6d} {sum(v) / len(v) / 1e3:9.1f} {100 * sum(v) / tot:5.1f} %")
PY
cat $R/gpurun_out/wide_by_grid.txt
