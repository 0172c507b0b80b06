#!/bin/bash
# round 3, first GPU call: new parity tests, the bench line with the pipelined-mode check, config-3 timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
timeout 600 python tools/gpu_cfg3.py > $OUT/cfg3.log 2>&1; echo "cfg3 rc=$?"; tail -6 $OUT/cfg3.log
timeout 900 python bench.py --breakdown-out $OUT/bd_cascade.json > $OUT/bench_cascade.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench_cascade.log | cut -c1-3000
