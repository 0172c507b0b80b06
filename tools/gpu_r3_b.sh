#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out; mkdir -p $OUT
timeout 300 python tools/gpu_async_diag.py > $OUT/async_diag.log 2>&1; echo "diag rc=$?"; grep -v amdgpu.ids $OUT/async_diag.log | tail -12
MINIMAGEN_STAGE_PRIORITY=0 timeout 300 python tools/gpu_async_diag.py > $OUT/async_diag_noprio.log 2>&1; echo "diag(noprio) rc=$?"; grep -v amdgpu.ids $OUT/async_diag_noprio.log | tail -8
B=8 T=25 timeout 300 python tools/gpu_async_diag.py > $OUT/async_diag_b8.log 2>&1; echo "diag(B8 T25) rc=$?"; grep -v amdgpu.ids $OUT/async_diag_b8.log | tail -8
timeout 300 python bench.py --workload base64 --precision half --no-cpu-baseline --no-secondary --no-t5 --no-pipeline --breakdown-out $OUT/bd_base_half.json > $OUT/bench_base_half.log 2>&1; tail -1 $OUT/bench_base_half.log | cut -c1-600
timeout 300 python bench.py --workload base64 --no-cpu-baseline --no-secondary --no-t5 --no-pipeline --breakdown-out $OUT/bd_base_fp32.json > $OUT/bench_base_fp32.log 2>&1; tail -1 $OUT/bench_base_fp32.log | cut -c1-600
timeout 400 python tools/gpu_dual_lane.py > $OUT/dual_lane.log 2>&1; echo "dual rc=$?"; grep -v amdgpu.ids $OUT/dual_lane.log | tail -8
