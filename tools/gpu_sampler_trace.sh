# per-kernel durations of the sampler tail inside the captured step (rocprofv3 kernel trace of a short cascade run), per library build
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for lib in $LIBS; do
  rm -rf $OUT/prof_s
  MINIMAGEN_HIP_LIB=$ROOTDIR/minimagen_amd/$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_s -o s -- python $ROOTDIR/bench.py --steps 1 --warmup 0 --timesteps 25 --no-cpu-baseline --no-secondary --no-breakdown > $OUT/prof_s.log 2>&1
  echo "== $lib"
  python - <<PY
import csv, collections, statistics as st, re
d = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/prof_s/s_kernel_trace.csv")):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(mi_.*|\(float.*|\(int\*.*", "", n).replace("void ", "")
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    d[(n, g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (n, g), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if any(k in n for k in ("cfg_x0", "posterior", "quantile", "step_advance", "attn_fold")) and len(v) >= 25:
        print(f"  {n:36s} grid {g:8d} x{len(v):3d}  avg {st.mean(v) / 1e3:6.2f} us  min {min(v) / 1e3:6.2f}")
PY
done
