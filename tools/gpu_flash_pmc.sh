#!/bin/bash
# PMC passes over tools/bench_flash.py (the 4096-token multi-query self-attention): where the cycles of flash_attn_mq_kernel go
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/flash_pmc; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for q in ${QTS:-1 2 4}; do
  MI_FLASH_MQ_QT=$q python $R/tools/bench_flash.py 32 4096 5
  i=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
    i=$((i+1)); rm -rf /tmp/fp
    MI_FLASH_MQ_QT=$q timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/fp -o f -- python $R/tools/bench_flash.py 32 4096 2 > $OUT/rocprof_${q}_${i}.log 2>&1
    python - <<PY
import csv, glob, collections
fs = glob.glob("/tmp/fp/**/*counter_collection.csv", recursive=True)
d = collections.defaultdict(list)
for f in fs:
    for r in csv.DictReader(open(f)):
        if "flash_attn_mq" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("QT=$q", {k: round(sum(v) / len(v)) for k, v in d.items()} if d else open("$OUT/rocprof_${q}_${i}.log").read()[-600:])
PY
  done
done
