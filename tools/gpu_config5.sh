#!/bin/bash
# BASELINE config 5's per-GPU shape (cascade 64 -> 256 -> 1024, B = 8, reduced precision): bench line with the per-launch breakdown of the 1024^2
# stage, rocprofv3 kernel trace + PMC passes of one sample() call (T = 25 per stage).  Output under gpurun_out/; copy summaries into profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out; mkdir -p $OUT
timeout 600 python bench.py --workload cascade64_256_1024 --batch 8 --precision half --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-t5 --breakdown-out $OUT/bd_config5_stage2.json > $OUT/bench_config5.log 2>&1; tail -1 $OUT/bench_config5.log | cut -c1-300
CMD="python $ROOTDIR/bench.py --workload cascade64_256_1024 --batch 8 --precision half --steps 1 --warmup 0 --timesteps 25 --no-cpu-baseline --no-secondary --no-breakdown --no-t5 --no-pipeline"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof5_trace -o c5 -- $CMD > $OUT/rocprof5_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/prof5_sq -o c5 -- $CMD > $OUT/rocprof5_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/prof5_fetch -o c5 -- $CMD > $OUT/rocprof5_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/prof5_write -o c5 -- $CMD > $OUT/rocprof5_write.log 2>&1
ls $OUT/prof5_*/ 2>/dev/null | head
