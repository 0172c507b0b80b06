cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out; mkdir -p $OUT
python tools/wide_unet_forward.py 2 2>&1 | tail -1; python tools/wide_unet_forward.py 16 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/prof_wide -o wide -- python $ROOTDIR/tools/wide_unet_forward.py 16 > $OUT/prof_wide.log 2>&1
python - <<PY
import csv, collections, statistics as st, re
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$OUT/prof_wide/wide_counter_collection.csv")):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(mi_.*|\(float.*|\(int\*.*", "", n).replace("void ", "")
    key = (n, int(r["Grid_Size"]), int(r["Workgroup_Size"]))
    d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    d[key]["dur"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for k, c in d.items():
    m = {n: st.mean(v) for n, v in c.items()}
    rows.append(dict(kernel=k[0], grid=k[1], wg=k[2], launches=len(c["dur"]), avg_us=round(m["dur"] / 1e3, 1), total_ms=round(sum(c["dur"]) / 1e6, 3),
                     mfma_pipe_util_pct=round(100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(m.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024, 1), 1),
                     lds_bank_conflict_pct=round(100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1), 1)))
rows.sort(key=lambda r: -r["total_ms"])
with open("$OUT/wide_pmc.csv", "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
tot = sum(r["total_ms"] for r in rows)
for r in rows[:16]: print(r)
print("total kernel ms", round(tot, 2))
PY
