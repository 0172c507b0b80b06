"""Turn the rocprofv3 outputs that tools/gpu_pmc.sh leaves in gpurun_out/ (scratch) into the tracked summaries
under profiles/:  python tools/summarize_profiles.py <tag>   (e.g. r01)"""
import collections
import csv
import os
import re
import shutil
import statistics as st
import sys

G = "gpurun_out"
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
os.makedirs("profiles", exist_ok=True)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(mi_.*|\(float.*|\(int\*.*", "", n).replace("void ", "")


def load(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return d
    for r in csv.DictReader(open(path)):
        key = (short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r["Workgroup_Size"]))
        d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d[key]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        d[key]["vgpr"], d[key]["agpr"], d[key]["lds"] = [int(r["VGPR_Count"])], [int(r["Accum_VGPR_Count"])], [int(r["LDS_Block_Size"])]
    return d


shutil.copy(f"{G}/prof_trace/cascade_kernel_stats.csv", f"profiles/{tag}_cascade_T25_kernel_stats.csv")
a, b = load(f"{G}/prof_sq/cascade_counter_collection.csv"), load(f"{G}/prof_sq2/cascade_counter_collection.csv")
f, w = load(f"{G}/prof_fetch/cascade_counter_collection.csv"), load(f"{G}/prof_write/cascade_counter_collection.csv")
rows = []
for key, c in a.items():
    m = {k: st.mean(v) for k, v in c.items()}
    n = len(c["SQ_WAVES"])
    m2 = {k: st.mean(v) for k, v in b.get(key, {}).items()}
    wv, wc = m["SQ_WAVES"], max(m["SQ_WAVE_CYCLES"], 1)
    rows.append(dict(
        kernel=key[0], grid=key[1], wg=key[2], launches=n, avg_us=round(m["dur_ns"] / 1e3, 1), total_ms=round(m["dur_ns"] * n / 1e6, 2),
        vgpr=m["vgpr"], agpr=m["agpr"], lds_bytes=m["lds"], waves=int(wv), valu_per_wave=round(m["SQ_INSTS_VALU"] / wv),
        salu_per_wave=round(m["SQ_INSTS_SALU"] / wv), lds_per_wave=round(m["SQ_INSTS_LDS"] / wv), smem_per_wave=round(m["SQ_INSTS_SMEM"] / wv),
        active_valu_pct=round(100 * m["SQ_ACTIVE_INST_VALU"] / wc, 1), wait_inst_any_pct=round(100 * m["SQ_WAIT_INST_ANY"] / wc, 1),
        wait_any_pct=round(100 * m2.get("SQ_WAIT_ANY", 0) / wc, 1),
        lds_bank_conflict_pct=round(100 * m2.get("SQ_LDS_BANK_CONFLICT", 0) / max(m2.get("SQ_LDS_IDX_ACTIVE", 1), 1), 1),
        # SQ_VALU_MFMA_BUSY_CYCLES sums SIMD-cycles (32 per v_mfma_f32_16x16x4_f32); GRBM_GUI_ACTIVE sums the 8 XCDs' clocks
        mfma_pipe_util_pct=round(100 * m2.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(m2.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024, 1), 1),
        # gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM section); KiB -> MB
        fetch_MB_x2=round(2 * st.mean(f[key]["FETCH_SIZE"]) / 1024, 1) if key in f and "FETCH_SIZE" in f[key] else None,
        write_MB=round(st.mean(w[key]["WRITE_SIZE"]) / 1024, 1) if key in w and "WRITE_SIZE" in w[key] else None))
rows.sort(key=lambda r: -r["total_ms"])
with open(f"profiles/{tag}_cascade_T25_pmc_by_launch_shape.csv", "w", newline="") as fh:
    wr = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    wr.writeheader()
    wr.writerows(rows)
print("wrote", len(rows), "rows")

# per launch shape from the plain --kernel-trace run (no counters: undisturbed durations)
tr = collections.defaultdict(list)
for r in csv.DictReader(open(f"{G}/prof_trace/cascade_kernel_trace.csv")):
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    tr[(short(r["Kernel_Name"]), grid, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
trows = sorted(({"kernel": k[0], "grid": k[1], "wg": k[2], "launches": len(v), "avg_us": round(st.mean(v) / 1e3, 2), "min_us": round(min(v) / 1e3, 2),
                 "total_ms": round(sum(v) / 1e6, 3)} for k, v in tr.items()), key=lambda r: -r["total_ms"])
with open(f"profiles/{tag}_cascade_T25_kernel_trace_by_launch_shape.csv", "w", newline="") as fh:
    wr = csv.DictWriter(fh, fieldnames=list(trows[0].keys()))
    wr.writeheader()
    wr.writerows(trows)
# the bench lines, breakdowns and the GPU test log of the same gpurun call
for src, dst in (("bench_cascade.log", f"{tag}_bench_cascade64_256_n1.json"), ("bench_half.log", f"{tag}_bench_cascade64_256_half_n1.json"),
                 ("bench_base.log", f"{tag}_bench_base64_n1.json")):
    if os.path.exists(f"{G}/{src}"):
        lines = [l for l in open(f"{G}/{src}") if l.startswith("{")]
        if lines:
            open(f"profiles/{dst}", "w").write(lines[-1])
for src, dst in (("bd_cascade.json", f"{tag}_bench_breakdown_stage1_256.json"), ("bd_base.json", f"{tag}_bench_breakdown_stage0_64.json"),
                 ("pytest_gpu.log", f"{tag}_pytest_gpu.log")):
    if os.path.exists(f"{G}/{src}"):
        shutil.copy(f"{G}/{src}", f"profiles/{dst}")
