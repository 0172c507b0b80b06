#!/bin/bash
# PMC counters of ONE conv launch shape (tools/bench_conv.py), old VALU kernel vs the row-paired kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export NTILE=${NTILE:-8}
for PATHK in old rp6; do
  CMD="python $ROOTDIR/tools/bench_conv.py 64 8 8 256 256 1 id $PATHK"
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/pmc_${PATHK}_sq -o c -- $CMD > $OUT/pmc_${PATHK}_sq.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR -d $OUT/pmc_${PATHK}_sq2 -o c -- $CMD > $OUT/pmc_${PATHK}_sq2.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc_${PATHK}_f -o c -- $CMD > $OUT/pmc_${PATHK}_f.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_${PATHK}_w -o c -- $CMD > $OUT/pmc_${PATHK}_w.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/pmc_${PATHK}_t -o c -- $CMD > $OUT/pmc_${PATHK}_t.log 2>&1
done
python - <<PY
import csv, glob, collections
for pk in ("old", "rp6"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$OUT/pmc_%s_*/**/*counter_collection.csv" % pk, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            if "conv" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in agg.items():
        n = 23.0
        print(pk, k)
        print("   ", {c: round(v / n / 1e6, 3) for c, v in sorted(d.items())}, "(millions per launch)")
PY
