#!/bin/bash
# rocprofv3 passes on the GPU box: kernel trace + stats, then PMC counters in their own runs (SQ set, FETCH_SIZE, WRITE_SIZE)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out; mkdir -p $OUT
export MINIMAGEN_ATTN_VARIANT=${MINIMAGEN_ATTN_VARIANT:-1} MINIMAGEN_CONV_SPLIT16=${MINIMAGEN_CONV_SPLIT16:-1}
CMD="python $ROOTDIR/bench.py --steps 1 --warmup 0 --timesteps 25 --no-cpu-baseline --no-secondary --no-breakdown"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o cascade -- $CMD > $OUT/rocprof_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/prof_sq -o cascade -- $CMD > $OUT/rocprof_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/prof_sq2 -o cascade -- $CMD > $OUT/rocprof_sq2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/prof_fetch -o cascade -- $CMD > $OUT/rocprof_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/prof_write -o cascade -- $CMD > $OUT/rocprof_write.log 2>&1
cd $OUT; find . -name "*.csv" | head -30; du -sh prof_*
tail -3 rocprof_sq.log; tail -3 rocprof_sq2.log
