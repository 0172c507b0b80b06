#!/bin/bash
# Runs on the GPU box (via gpurun): GPU test-suite, smoke, benches, rocprofv3 kernel trace.  Logs -> gpurun_out/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx" | head -6 > $OUT/rocminfo.txt
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
timeout 900 python bench.py --steps 2 --warmup 1 --breakdown-out $OUT/breakdown_cascade.json > $OUT/bench_cascade.log 2>&1; echo "bench rc=$?"
timeout 300 python bench.py --workload base64 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --breakdown-out $OUT/breakdown_base.json > $OUT/bench_base.log 2>&1
ROOTDIR=$(pwd)
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $ROOTDIR/$OUT/prof -o cascade -- python $ROOTDIR/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-breakdown > $ROOTDIR/$OUT/rocprof.log 2>&1 )
find $OUT/prof -name "*stats*" | head
echo "---- pytest"; tail -25 $OUT/pytest_gpu.log
echo "---- smoke"; tail -5 $OUT/smoke.log
echo "---- bench cascade"; tail -3 $OUT/bench_cascade.log
echo "---- bench base"; tail -3 $OUT/bench_base.log
