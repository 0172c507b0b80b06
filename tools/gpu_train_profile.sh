#!/bin/bash
# rocprofv3 kernel trace of the training step (tools/gpu_train_step.py); the steady-state tail (the last 10 optimiser steps of the SR U-Net)
# aggregated by kernel -> gpurun_out/train_tail_hip<0|1>.txt (the trace itself is dropped: library auto-tuning pollutes its head)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in ${MODES:-1 0}; do
  rm -rf /tmp/train_prof
  PROFILE_ONLY=$mode B1=${B1:-32} B2=${B2:-32} timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/train_prof -o train -- python $R/tools/gpu_train_step.py > $R/gpurun_out/train_prof_$mode.log 2>&1
  f=$(find /tmp/train_prof -name "*kernel_trace.csv" | head -1)
  echo "== MINIMAGEN_TRAIN_HIP=$mode"
  grep "unet2" $R/gpurun_out/train_prof_$mode.log | tail -3
  if [ -n "$f" ]; then python $R/tools/trace_tail_stats.py "$f" ${TAIL:-0.3} 32 | tee $R/gpurun_out/train_tail_hip$mode.txt; fi
done
