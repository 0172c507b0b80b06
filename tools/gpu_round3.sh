#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
for AV in 0 1; do for SP in 0 1; do
  export MINIMAGEN_ATTN_VARIANT=$AV MINIMAGEN_CONV_SPLIT16=$SP
  timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --breakdown-out $OUT/bd_cascade_a${AV}_s${SP}.json > $OUT/bench_cascade_a${AV}_s${SP}.log 2>&1
  echo "== attn_variant=$AV split16=$SP"; tail -1 $OUT/bench_cascade_a${AV}_s${SP}.log | cut -c1-200
done; done
export MINIMAGEN_ATTN_VARIANT=1 MINIMAGEN_CONV_SPLIT16=1
CMD="python $ROOTDIR/bench.py --steps 1 --warmup 0 --timesteps 20 --no-cpu-baseline --no-breakdown"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/prof_sq -o cascade -- $CMD > $OUT/rocprof_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/prof_sq2 -o cascade -- $CMD > $OUT/rocprof_sq2.log 2>&1
