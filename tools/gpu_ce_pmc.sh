cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out; mkdir -p $OUT
python tools/bench_ce.py 32 256 256 8 1 0; python tools/bench_ce.py 32 256 256 8 1 1; python tools/bench_ce.py 32 256 256 0 0 0; python tools/bench_ce.py 32 64 64 9 1 0
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/prof_ce -o ce -- python $ROOTDIR/tools/bench_ce.py 32 256 256 8 1 0 > $OUT/prof_ce.log 2>&1
python - <<PY
import csv, collections, statistics as st
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$OUT/prof_ce/ce_counter_collection.csv")):
    if "crossembed" in r["Kernel_Name"]:
        d[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in d.items():
    m = {n: st.mean(v) for n, v in c.items()}
    print(k, {n: round(v) for n, v in m.items()})
    print("  lds bank conflict / idx active = %.2f" % (m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1)), " mfma busy / (gui/8*1024) = %.2f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)))
PY
