#!/bin/bash
# one rocprofv3 PMC pass (LDS bank conflicts, MFMA busy) over a short synchronous cascade run -> gpurun_out/$1/lds_by_kernel.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out/${1:-lds}; mkdir -p $OUT
CMD="python $ROOTDIR/bench.py --steps 1 --warmup 0 --timesteps 25 --no-cpu-baseline --no-secondary --no-breakdown --no-t5 --no-pipeline"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/prof_lds -o cascade -- $CMD > $OUT/rocprof.log 2>&1
python - <<PY > $OUT/lds_by_kernel.txt
import csv, collections, re, glob
f = glob.glob("/tmp/prof_lds/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]); n = re.sub(r"\(mi_.*|\(float.*|\(int\*.*", "", n)
    d[(n, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in d.items():
    m = {a: sum(v) / len(v) for a, v in c.items()}
    rows.append((len(c["SQ_LDS_IDX_ACTIVE"]) * m.get("SQ_LDS_IDX_ACTIVE", 0), k, m))
print("kernel | grid | launches | LDS bank-conflict cycles / LDS active cycles")
for _, k, m in sorted(rows, reverse=True)[:24]:
    print(f"{k[0][:78]:78s} {k[1]:8d} {len(d[k]['SQ_LDS_IDX_ACTIVE']):5d}  {100 * m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1):5.1f} %")
PY
cat $OUT/lds_by_kernel.txt
